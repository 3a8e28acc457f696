"""Diagnostic (GPU box): smoke()'s state, forward() + step() with the grid line search on both sides; per-world errors and iterations."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402
from oracle.oracle import OracleSim  # noqa: E402

model = robots.load_model("g1_velocity_flat")
nworld = 8
rng = np.random.default_rng(0)
qpos = np.tile(model.key_qpos[0], (nworld, 1))
qpos[:, 7:] += rng.normal(0, 0.05, size=(nworld, model.nq - 7))
qpos[:, 2] -= 0.02
qvel = rng.normal(0, 0.1, size=(nworld, model.nv))
ctrl = qpos[:, 7:] + rng.normal(0, 0.1, size=(nworld, model.nu))
for par in (True, False):
  sim = Simulation(nworld, SimulationCfg(njmax=300, ls_parallel=par, use_graph=False), model, "cuda:0")
  sim.ls_parallel = par
  ora = OracleSim(model, nworld, njmax=300, precision="f64", ls_parallel=par)
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    getattr(ora, f)[:] = v.astype(np.float32)
  sim.forward(); ora.forward()
  torch.cuda.synchronize()
  pw = lambda a, b: np.abs(a.astype(np.float64) - b).max(axis=1) / np.abs(b).max(axis=1)
  print("ls_parallel", par, "forward qacc err per world", np.array2string(pw(sim.data.qacc.cpu().numpy(), ora.qacc), precision=2), "niter gpu", sim.data.solver_niter.cpu().numpy().ravel(), "oracle", ora.solver_niter.ravel())
  sim.step(); ora.step()
  torch.cuda.synchronize()
  for f in ("qacc", "qvel", "qpos"):
    print("   step", f, "err per world", np.array2string(pw(getattr(sim.data, f).cpu().numpy(), getattr(ora, f)), precision=2))
  print("   step niter gpu", sim.data.solver_niter.cpu().numpy().ravel(), "oracle", ora.solver_niter.ravel())
