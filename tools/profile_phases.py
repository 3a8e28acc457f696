"""Per-phase shader-clock breakdown of the solve kernel (GPU box).

Build first (in the dev container):  python -m mjlab_amd.native --out gpurun_prof/libmjlab_amd_prof.so -DMJLAB_PROFILE
Run: MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so python tools/profile_phases.py
"""
import sys
from pathlib import Path

import os

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots  # noqa: E402
from mjlab_amd.rollout import PhysicsRollout, g1_action_scale  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402

NAMES = ["M load+factor+qLD", "qacc_smooth solve", "warmstart", "init hessian pass", "H factor+solve", "LS prep (Mv, Jv)",
         "LS evals", "post-LS update + J^T f", "solve tail", "integrate", "ls evals (count)", "line searches (count)", "chol_factor (all sites)", "chol_solve (all sites)", "factor calls (count)", "solve calls (count)"]

model = robots.load_model(os.environ.get("SCENE", "g1_velocity_flat"))
NW = int(os.environ.get("NWORLD", "4096"))  # 256 = one wave per CU: pure single-wave latency
sim = Simulation(NW, SimulationCfg(njmax=300, use_graph=False, fuse=os.environ.get("FUSE", "stage")), model, "cuda:0")
roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=42)
for _ in range(40):
  roll.step(roll.random_action())
sim.data.profile[:] = 0
nstep = 20
for _ in range(nstep):
  sim.step()
torch.cuda.synchronize()
pall = sim.data.profile.cpu().numpy().astype(np.float64) / nstep
p = pall[:, :16]
tot = p[:, [0, 2, 3, 5, 6, 7, 8, 9, 12, 13]].sum(axis=1)
print(f"mean cycles per world-step in k_solve_integrate: {tot.mean():.0f}  (p50 {np.percentile(tot,50):.0f}, p90 {np.percentile(tot,90):.0f}, max {tot.max():.0f})")
for i, n in enumerate(NAMES):
  if i in (1, 4):
    continue  # (re-used as counters: see below)
  if i < 10 or i in (12, 13):
    print(f"  {n:28s} {p[:, i].mean():10.0f} cycles  {100*p[:, i].mean()/tot.mean():5.1f}%")
  else:
    print(f"  {n:28s} {p[:, i].mean():10.2f}")
print("nefc mean", sim.data.nefc.float().mean().item(), "niter mean", sim.data.solver_niter.float().mean().item())
print(f"low-rank corrections per world-step (MJLAB_SMW): {p[:, 1].mean():.2f} iterations ran on a corrected factor ({p[:, 4].mean():.2f} set-ups), "
      f"{p[:, 14].mean():.2f} factorizations counted at the M / Newton sites, {p[:, 11].mean():.2f} line searches")
# the kernel ends with its slowest wave: same breakdown for the slowest 2% of worlds
slow = np.argsort(tot)[-max(1, len(tot) // 50):]
print(f"slowest 2% of worlds: mean cycles {tot[slow].mean():.0f}; nefc {sim.data.nefc.cpu().numpy().ravel()[slow].mean():.1f}; niter {sim.data.solver_niter.cpu().numpy().ravel()[slow].mean():.2f}")
for i, n in enumerate(NAMES):
  if i < 10 or i in (12, 13):
    print(f"  {n:28s} {p[slow, i].mean():10.0f} cycles  {100*p[slow, i].mean()/tot[slow].mean():5.1f}%")
  else:
    print(f"  {n:28s} {p[slow, i].mean():10.2f}")

PNAMES = ["kinematics levels", "ixform/geoms/sites", "write kinematics", "subtree_com+cinert+cdof", "write com/cinert/cdof", "crb sums + crb*cdof", "M assembly", "write qM"]
pp = pall[:, 16:24]
tot = pp.sum(axis=1)
print(f"k_position mean cycles per world-step: {tot.mean():.0f}")
for i, n in enumerate(PNAMES):
  print(f"  {n:28s} {pp[:, i].mean():10.0f} cycles  {100*pp[:, i].mean()/tot.mean():5.1f}%")

for title, base, names in (
  ("k_velocity", 24, ["prologue loads", "dof chain sums (cdof_dot)", "body chain sums (cvel, cfrc)", "writes + subtree sums", "actuation + bias + stores"]),
  ("k_collision", 32, ["staging (poses, constants)", "static pair sweeps", "terrain narrow phase + ncon", "terrain grid walk"]),
  ("k_constraint", 40, ["limits", "contacts phase A (per contact)", "contacts phase B (rows)", "nefc + sensors"]),
):
  pp = pall[:, base : base + len(names)]
  tot = pp.sum(axis=1)
  print(f"{title} mean cycles per world-step: {tot.mean():.0f}")
  for i, n in enumerate(names):
    print(f"  {n:36s} {pp[:, i].mean():10.0f} cycles  {100*pp[:, i].mean()/tot.mean():5.1f}%")
