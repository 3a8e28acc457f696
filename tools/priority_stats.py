"""Which worlds a k_control_step launch waits for (profiling build, GPU box): per-launch wave lifetimes against the row
count / Newton iterations of the world, by row-count bucket.  Motivates common.h::wave_priority.

  MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so python tools/priority_stats.py [scene]
"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots  # noqa: E402
from mjlab_amd.rollout import VELOCITY_TASK_EVENTS, PhysicsRollout, g1_action_scale, go1_action_scale  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "g1_velocity_flat"
robot = "go1" if scene.startswith("go1") else "g1"
model = robots.load_model(scene)
sim = Simulation(4096, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
scale = go1_action_scale(model) if robot == "go1" else g1_action_scale(model)
roll = PhysicsRollout(sim, action_scale=scale, seed=42, substeps_per_call=4, control_kernel=True, **VELOCITY_TASK_EVENTS[robot])
for _ in range(60):
  roll.step(roll.random_action())
life, nefc, niter, ms = [], [], [], []
for _ in range(30):
  sim.data.profile[:] = 0
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a = roll.random_action()
  e0.record()
  roll.step(a)
  e1.record()
  torch.cuda.synchronize()
  ms.append(e0.elapsed_time(e1))
  life.append(sim.data.profile.cpu().numpy().astype(np.float64)[:, 63])
  nefc.append(sim.data.nefc.cpu().numpy().ravel())
  niter.append(sim.data.solver_niter.cpu().numpy().ravel())
life, nefc, niter = np.concatenate(life), np.concatenate(nefc), np.concatenate(niter)
print(f"{scene}: {np.mean(ms):.3f} ms per control step (profiling build); lifetime mean {life.mean():.0f}, mean of the per-launch max {np.mean([x.max() for x in np.split(life, 30)]):.0f}")
print("rows quantiles 50 / 90 / 98 / 99.5 / max:", [int(np.percentile(nefc, q)) for q in (50, 90, 98, 99.5)], int(nefc.max()))
for name, x in (("rows", nefc), ("iterations", niter), ("rows x (iterations + 2)", nefc * (niter + 2.0))):
  print(f"correlation of the wave lifetime with {name}: {np.corrcoef(life, x)[0, 1]:.3f}")
edges = [0, 16, 32, 48, 64, 80, 10**6]
for lo, hi in zip(edges, edges[1:]):
  sel = (nefc > lo) & (nefc <= hi)
  if sel.any():
    print(f"rows in ({lo}, {hi if hi < 10**6 else 'inf'}]: {100 * sel.mean():6.2f} % of the worlds, lifetime {life[sel].mean() / life.mean():.2f} x the mean, iterations {niter[sel].mean():.2f}")
top = life >= np.percentile(life, 99)
print(f"slowest 1 % of the waves: rows {nefc[top].mean():.1f}, iterations {niter[top].mean():.2f}; share with rows > 64: {100 * (nefc[top] > 64).mean():.0f} %, > 32: {100 * (nefc[top] > 32).mean():.0f} %")
