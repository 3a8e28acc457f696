"""Field-by-field comparison of the HIP path against the CPU oracle (debug aid, GPU box)."""

import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from mjlab_amd import robots  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402
from oracle.oracle import OracleSim  # noqa: E402

FIELDS = [
  "xpos", "xquat", "xmat", "xipos", "ximat", "xanchor", "xaxis", "geom_xpos", "geom_xmat", "site_xpos", "subtree_com",
  "cinert", "cdof", "qM", "cvel", "cdof_dot", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "actuator_force",
  "qfrc_smooth", "qacc_smooth", "qfrc_constraint", "qacc", "qacc_warmstart", "sensordata", "qpos", "qvel", "time",
]  # fmt: skip


def relerr(a, b):
  a = np.asarray(a, dtype=np.float64).reshape(-1)
  b = np.asarray(b, dtype=np.float64).reshape(-1)
  scale = max(1e-6, np.abs(b).max()) if b.size else 1.0
  return (np.abs(a - b).max() / scale) if b.size else 0.0


def seed_state(model, nworld, rng, key=0, noise=0.1):
  qpos = np.tile(model.key_qpos[key] if model.nkey else model.qpos0, (nworld, 1))
  qvel = rng.normal(0, noise * 3, size=(nworld, model.nv))
  ctrl = np.zeros((nworld, model.nu))
  for j in range(model.njnt):
    qa = model.jnt_qposadr[j]
    if model.jnt_type[j] == 0:
      qpos[:, qa : qa + 2] += rng.uniform(-0.5, 0.5, size=(nworld, 2))
      q = qpos[:, qa + 3 : qa + 7] + rng.normal(0, noise, size=(nworld, 4))
      qpos[:, qa + 3 : qa + 7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    else:
      qpos[:, qa] += rng.normal(0, noise, size=nworld)
  if model.nu:
    jn = model.actuator_trnid[:, 0]
    ctrl = qpos[:, model.jnt_qposadr[jn]] + rng.normal(0, 0.2, size=(nworld, model.nu))
  return qpos, qvel, ctrl


def compare(name, model, nworld=8, nsteps=3, seed=0, graph=False):
  rng = np.random.default_rng(seed)
  qpos, qvel, ctrl = seed_state(model, nworld, rng)
  sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=graph), model, "cuda:0")
  ora = OracleSim(model, nworld, njmax=300, precision="f64")
  for s, to in ((sim.data, lambda x: torch.from_numpy(x.astype(np.float32)).cuda()), (ora, lambda x: x)):
    s.qpos[:] = to(qpos)
    s.qvel[:] = to(qvel)
    s.ctrl[:] = to(ctrl)
  print(f"== {name}: nworld={nworld} nv={model.nv} lds={sim.lds_bytes()}")
  sim.forward()
  ora.forward()
  torch.cuda.synchronize()
  worst = 0.0
  print(" forward: ncon gpu", sim.data.ncon.cpu().numpy().ravel()[:8], "oracle", ora.ncon.ravel()[:8])
  print("          nefc gpu", sim.data.nefc.cpu().numpy().ravel()[:8], "oracle", ora.nefc.ravel()[:8])
  print("          niter gpu", sim.data.solver_niter.cpu().numpy().ravel()[:8], "oracle", ora.solver_niter.ravel()[:8])
  for f in FIELDS:
    e = relerr(getattr(sim.data, f).cpu().numpy(), getattr(ora, f))
    worst = max(worst, e)
    flag = "  <<<<" if e > 1e-3 else ""
    print(f"   {f:18s} {e:.3e}{flag}")
  for k in range(nsteps):
    sim.step()
    ora.step()
  torch.cuda.synchronize()
  print(f" after {nsteps} steps:")
  for f in ("qpos", "qvel", "qacc", "sensordata", "time"):
    e = relerr(getattr(sim.data, f).cpu().numpy(), getattr(ora, f))
    print(f"   {f:18s} {e:.3e}" + ("  <<<<" if e > 1e-3 else ""))
  print("   niter gpu", sim.data.solver_niter.cpu().numpy().ravel()[:8], "oracle", ora.solver_niter.ravel()[:8])
  return sim


def main():
  print(torch.cuda.get_device_name(0))
  compare("pendulum", robots.pendulum_model(), nworld=2)
  compare("box", robots.box_model(), nworld=4)
  compare("mixed", robots.mixed_model(), nworld=8)
  compare("go1", robots.load_model("go1_velocity_flat"), nworld=8)
  sim = compare("g1", robots.load_model("g1_velocity_flat"), nworld=8)
  compare("g1-graph", robots.load_model("g1_velocity_flat"), nworld=8, graph=True)
  # quick timing, 4096 worlds
  model = robots.load_model("g1_velocity_flat")
  for graph in (False, True):
    sim = Simulation(4096, SimulationCfg(njmax=300, use_graph=graph), model, "cuda:0")
    rng = np.random.default_rng(1)
    qpos, qvel, ctrl = seed_state(model, 4096, rng, noise=0.02)
    sim.data.qpos[:] = torch.from_numpy(qpos.astype(np.float32)).cuda()
    sim.data.ctrl[:] = torch.from_numpy(ctrl.astype(np.float32)).cuda()
    for _ in range(20):
      sim.step()
    torch.cuda.synchronize()
    t = time.time()
    n = 100
    for _ in range(n):
      sim.step()
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    print(f"G1 x4096 graph={graph}: {dt*1e3:.3f} ms/step -> {4096/dt/1e6:.3f} M world-steps/s; nefc mean {sim.data.nefc.float().mean().item():.1f} niter mean {sim.data.solver_niter.float().mean().item():.2f}")
    print("   nan?", torch.isnan(sim.data.qpos).any().item(), "z mean", sim.data.qpos[:, 2].mean().item())


if __name__ == "__main__":
  main()
