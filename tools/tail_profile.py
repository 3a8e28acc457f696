"""How long each world's wave lives inside one k_control_step launch (profiling build, GPU box):
the launch ends with its slowest world, so mean / max of these lifetimes is the share of the launch
during which the average SIMD slot is still occupied.

  MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so python tools/tail_profile.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots  # noqa: E402
from mjlab_amd.rollout import VELOCITY_TASK_EVENTS, PhysicsRollout, g1_action_scale  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402

model = robots.load_model("g1_velocity_flat")
sim = Simulation(4096, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=42, substeps_per_call=4, control_kernel=True, **VELOCITY_TASK_EVENTS["g1"])
for _ in range(40):
  roll.step(roll.random_action())
sim.data.profile[:] = 0
n = 20
t = []
for _ in range(n):
  a = roll.random_action()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  roll.step(a)
  e1.record()
  torch.cuda.synchronize()
  t.append(e0.elapsed_time(e1))
life = sim.data.profile.cpu().numpy().astype(np.float64)[:, 63] / n
print(f"k_control_step (profiling build): {np.mean(t):.3f} ms per control step incl. torch.rand")
print(f"wave lifetime per launch, cycles of clock64: mean {life.mean():.0f}  p50 {np.percentile(life, 50):.0f}  p90 {np.percentile(life, 90):.0f}  "
      f"p99 {np.percentile(life, 99):.0f}  max {life.max():.0f}")
print(f"mean / max = {life.mean() / life.max():.3f}   mean / p99 = {life.mean() / np.percentile(life, 99):.3f}")
nefc = sim.data.nefc.cpu().numpy().ravel()
heavy = life > np.percentile(life, 98)
print(f"slowest 2 % of worlds: nefc {nefc[heavy].mean():.1f} (all worlds {nefc.mean():.1f}); lifetime {life[heavy].mean():.0f}")

# the same with the worlds dealt out over the SIMDs by expected cost (PhysicsRollout.balance_worlds)
for stride in (1024, 512, 256):
  sim.data.profile[:] = 0
  t = []
  roll.world_order = None
  for k in range(n):
    if k % 4 == 0:
      roll.balance_worlds(stride) if stride == 1024 else None
      if stride != 1024:  # other guesses of which workgroups share a SIMD: ranks dealt with this stride
        d = sim.data
        cost = d.nefc.view(-1).float() * (d.solver_niter.view(-1).float() + 2.0)
        idx = torch.argsort(cost, descending=True).to(torch.int32)
        m = 4096 // stride
        order = torch.empty(4096, dtype=torch.int32, device="cuda")
        kk = torch.arange(stride, device="cuda")
        for tt in range(m):
          ranks = tt * stride + (kk if tt % 2 == 0 else stride - 1 - kk)
          order[kk + stride * tt] = idx[ranks]
        roll.world_order = order
    a = roll.random_action()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    roll.step(a)
    e1.record()
    torch.cuda.synchronize()
    t.append(e0.elapsed_time(e1))
  life = sim.data.profile.cpu().numpy().astype(np.float64)[:, 63] / n
  print(f"balanced, stride {stride}: {np.mean(t):.3f} ms per control step; lifetime mean {life.mean():.0f} max {life.max():.0f}  mean / max = {life.mean() / life.max():.3f}")
