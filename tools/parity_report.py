"""GPU-vs-oracle parity over the state distribution of a rollout (GPU box; oracle on the host cores).

For each scene: N worlds are rolled out on the GPU with random actions, the resulting states
(qpos, qvel, ctrl, qacc_warmstart) are handed to the fp64 CPU oracle, both sides run forward()
and one step(), and per-field relative errors (max |gpu - oracle| / max |oracle| per world) are
summarised.  Output is committed as profiles/<tag>/parity_report.txt.
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots  # noqa: E402
from mjlab_amd.rollout import PhysicsRollout, g1_action_scale, go1_action_scale  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402
from oracle.oracle import OracleSim  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
FWD = ["xpos", "xquat", "xipos", "subtree_com", "cvel", "geom_xpos", "qM", "qfrc_bias", "actuator_force", "qfrc_smooth",
       "qacc_smooth", "efc_D*", "efc_aref*", "efc_J*", "qacc", "qfrc_constraint"]
STEP = ["qpos", "qvel"]
cores = os.cpu_count() or 8


def per_world_rel(a, b):
  a = a.reshape(a.shape[0], -1).astype(np.float64)
  b = b.reshape(b.shape[0], -1).astype(np.float64)
  return np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-6)


SCENES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["g1_velocity_flat", "g1_tracking_flat", "go1_velocity_flat", "g1_velocity_rough", "go1_velocity_rough"]
for scene in SCENES:
  model = robots.load_model(scene)
  njmax = 300 if "velocity" in scene else 250
  sim = Simulation(N, SimulationCfg(njmax=njmax, use_graph=False), model, "cuda:0")
  scale = g1_action_scale(model) if scene.startswith("g1") else go1_action_scale(model)
  roll = PhysicsRollout(sim, action_scale=scale, seed=123, min_height=0.3 if scene.startswith("g1") else 0.15)
  for _ in range(25):
    roll.step(roll.random_action())
  sim.data.ctrl[:] = roll.default_joint + roll.random_action() * roll.action_scale
  ora = OracleSim(model, N, njmax=njmax, precision="f64")
  for f in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(ora, f)[:] = getattr(sim.data, f).cpu().numpy().astype(np.float64)
  ws = sim.data.qacc_warmstart.clone()
  sim.forward()
  ora.forward(nthread=cores)
  torch.cuda.synchronize()
  nefc_g, nefc_o = sim.data.nefc.cpu().numpy().ravel(), ora.nefc.ravel()
  ncon_g, ncon_o = sim.data.ncon.cpu().numpy().ravel(), ora.ncon.ravel()
  print(f"== {scene}: {N} worlds after 25 control steps of random actions; nefc mean {nefc_o.mean():.1f} max {nefc_o.max()}, "
        f"ncon mean {ncon_o.mean():.1f}")
  same = (nefc_g == nefc_o) & (ncon_g == ncon_o)
  print(f"   identical contact / row counts: {same.sum()} of {N} worlds"
        f" (fp32 vs fp64 can flip a contact that sits exactly on its margin)")
  print(f"   sensordata identical: {(sim.data.sensordata.cpu().numpy() == ora.sensordata.astype(np.float32)).all(axis=1).sum()} of {N}")
  print(f"   {'field':18s} {'median':>10s} {'p99':>10s} {'max':>10s}   (relative error per world, worlds with identical counts)")
  nv = model.nv
  for f in FWD:
    name = f.rstrip("*")
    g = getattr(sim.data, name).cpu().numpy()
    o = getattr(ora, name)
    if f.endswith("*"):  # row arrays: only the first nefc rows are defined
      w = name == "efc_J" and nv or 1
      rows = np.arange(njmax)[None, :] < nefc_o[:, None]
      mask = np.repeat(rows, w, axis=1) if w > 1 else rows
      g = np.where(mask, g.reshape(N, -1), 0)
      o = np.where(mask, o.reshape(N, -1), 0)
    e = per_world_rel(g, o)[same]
    print(f"   {name:18s} {np.median(e):10.2e} {np.percentile(e, 99):10.2e} {e.max():10.2e}")
  niter_g, niter_o = sim.data.solver_niter.cpu().numpy().ravel(), ora.solver_niter.ravel()
  print(f"   Newton iterations: gpu mean {niter_g.mean():.2f} (max {niter_g.max()}), oracle mean {niter_o.mean():.2f} (max {niter_o.max()})")
  # one step from the same state and warm start
  sim.data.qacc_warmstart[:] = ws
  ora.qacc_warmstart[:] = ws.cpu().numpy().astype(np.float64)
  sim.step()
  ora.step(1, nthread=cores)
  torch.cuda.synchronize()
  for f in STEP:
    e = per_world_rel(getattr(sim.data, f).cpu().numpy(), getattr(ora, f))[same]
    print(f"   step {f:13s} {np.median(e):10.2e} {np.percentile(e, 99):10.2e} {e.max():10.2e}")
  del sim, roll, ora
  torch.cuda.empty_cache()
