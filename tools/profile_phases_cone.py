"""Per-phase shader-clock breakdown of the elliptic-cone solve kernel k_solve_cone (GPU box, profiling library):
  MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so python tools/profile_phases_cone.py       (NWORLD=1024: one wave per SIMD)"""
import copy
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import mjcf, robots  # noqa: E402
from mjlab_amd.rollout import PhysicsRollout, g1_action_scale  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402

NAMES = {0: "M load + factor + qacc_smooth", 1: "rows' D / roles, warm start", 2: "first constraint update (cost, J^T f, H)", 4: "H factor + solve (per iteration)",
         5: "LS prep (Mv, Jv, quad)", 6: "line search", 3: "constraint update after the step", 7: "convergence test"}
model = copy.deepcopy(robots.load_model(os.environ.get("SCENE", "g1_velocity_flat")))
model.opt.cone = mjcf.CONE_ELLIPTIC
NW = int(os.environ.get("NWORLD", "4096"))
sim = Simulation(NW, SimulationCfg(njmax=300, use_graph=False, fuse="stage"), model, "cuda:0")
roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=42)
for _ in range(40):
  roll.step(roll.random_action())
sim.data.profile[:] = 0
nstep = 20
for _ in range(nstep):
  sim.step()
torch.cuda.synchronize()
p = sim.data.profile.cpu().numpy().astype(np.float64)[:, 48:64] / nstep
tot = p[:, list(NAMES)].sum(axis=1)
print(f"{NW} worlds: mean cycles per world-step in k_solve_cone: {tot.mean():.0f}  (p50 {np.percentile(tot, 50):.0f}, p90 {np.percentile(tot, 90):.0f}, max {tot.max():.0f}); "
      f"iterations {p[:, 8].mean():.2f}; nefc mean {sim.data.nefc.float().mean().item():.1f}")
for i, n in NAMES.items():
  print(f"  {n:44s} {p[:, i].mean():10.0f} cycles  {100 * p[:, i].mean() / tot.mean():5.1f}%")
