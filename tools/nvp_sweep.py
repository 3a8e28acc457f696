"""Cost of one physics step vs the padded dof count the solve / substep kernels are instantiated for (GPU box):
chain models of tests/test_gpu_parity.py::_snake_model (free base + n hinge links on the plane), 4096 worlds,
Simulation.step() with the default launch structure (one kernel per substep)."""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
spec = importlib.util.spec_from_file_location("tp", ROOT / "tests" / "test_gpu_parity.py")
tp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tp)
from mjlab_amd.csrc_sizes import solve_nvp  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402

NW = 4096
for nlink in (2, 13, 25, 29, 33, 41, 57):
  model = tp._snake_model(nlink)
  sim = Simulation(NW, SimulationCfg(njmax=300), model, "cuda:0")
  rng = np.random.default_rng(0)
  q = np.tile(model.qpos0, (NW, 1))
  q[:, 2] = 0.026
  q[:, 7:] = rng.uniform(-1, 1, (NW, nlink)) * np.where(np.arange(nlink) % 2, 0.01, 0.3)
  sim.data.qpos[:] = torch.from_numpy(q.astype(np.float32)).cuda()
  for _ in range(20):
    sim.step()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50):
    sim.step()
  e1.record()
  torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 50 * 1e3
  print(f"nv {model.nv:2d} -> NVP {solve_nvp(model.nv):2d}: {us:7.1f} us per step of {NW} worlds; nefc mean {float(sim.data.nefc.float().mean()):5.1f}, "
        f"Newton iterations mean {float(sim.data.solver_niter.float().mean()):.2f}")
