"""Time of one physics step of 4096 G1 worlds through the STAGE kernels: pyramid against elliptic cones, both line searches
(csrc/stage_cone.h is a correctness feature off the measured path; this records what it costs).  Run on the GPU box:
  python tools/elliptic_timing.py > gpurun_out/elliptic_timing.txt"""

import copy
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

from make_golden import golden_inputs  # noqa: E402

from mjlab_amd import mjcf, robots  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402


def run(cone: int, lsp: bool, fuse: str, nworld: int = 4096, steps: int = 60, iterations: int | None = None, ls_iterations: int | None = None) -> tuple[float, float, float]:
  model = copy.deepcopy(robots.load_model("g1_velocity_flat"))
  model.opt.cone = cone
  if iterations is not None:
    model.opt.iterations = iterations
  if ls_iterations is not None:
    model.opt.ls_iterations = ls_iterations
  sim = Simulation(nworld, SimulationCfg(njmax=300, ls_parallel=lsp, fuse=fuse), model, "cuda:0")
  qpos, qvel, ctrl = golden_inputs(model, nworld, 5)
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
  for _ in range(20):
    sim.step()
  torch.cuda.synchronize()
  t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0.record()
  for _ in range(steps):
    sim.step()
  t1.record()
  torch.cuda.synchronize()
  return t0.elapsed_time(t1) / steps, float(sim.data.solver_niter.float().mean()), float(sim.data.nefc.float().mean())


if __name__ == "__main__":
  if len(sys.argv) > 1 and sys.argv[1] == "sweep":  # where the cone solve's time goes: iterations x line-search evaluations
    for it in (0, 1, 2, 10):
      for ls in (1, 20):
        for lsp in (True, False):
          print(f"elliptic iterations<={it:2d} ls_iterations={ls:2d} ls_parallel={lsp!s:5s}:", "%7.3f ms (%.2f iterations, %.1f rows)" % run(mjcf.CONE_ELLIPTIC, lsp, "stage", iterations=it, ls_iterations=ls))
    sys.exit(0)
  if len(sys.argv) > 1:  # one configuration only (under rocprofv3): elliptic | pyramid
    print(run(mjcf.CONE_ELLIPTIC if sys.argv[1] == "elliptic" else mjcf.CONE_PYRAMIDAL, True, "stage"))
    sys.exit(0)
  print("G1 velocity-flat, 4096 worlds, ms per physics step (mean Newton iterations, mean rows) -- falling robots, golden_inputs seed 5")
  for name, cone, fuse in (("pyramid, fused step kernel", mjcf.CONE_PYRAMIDAL, "step"), ("pyramid, stage kernels", mjcf.CONE_PYRAMIDAL, "stage"), ("elliptic, stage kernels", mjcf.CONE_ELLIPTIC, "stage"), ("elliptic, fused step kernel", mjcf.CONE_ELLIPTIC, "step")):
    for lsp in (True, False):
      ms, it, rows = run(cone, lsp, fuse)
      print(f"{name:28s} ls_parallel={lsp!s:5s}: {ms:7.3f} ms  ({it:.2f} iterations, {rows:.1f} rows)")
