"""Cycles per phase of k_solve_cone (csrc/stage_cone.h) from a -DMJLAB_PROFILE build: G1 velocity-flat, falling robots.
  python tools/cone_phases.py build     (here: mjlab_amd/csrc/build_prof/libmjlab_amd_cone_prof.so, size 36 only)
  MJLAB_AMD_LIB=mjlab_amd/csrc/build_prof/libmjlab_amd_cone_prof.so python tools/cone_phases.py   (GPU box)"""

import copy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
OUT = ROOT / "mjlab_amd" / "csrc" / "build_prof" / "libmjlab_amd_cone_prof.so"

if len(sys.argv) > 1 and sys.argv[1] == "build":
  from mjlab_amd import native

  OUT.parent.mkdir(exist_ok=True)
  native.build(force=True, out=OUT, defines=("MJLAB_PROFILE",), only_size=36)
  print(OUT)
  sys.exit(0)

import numpy as np
import torch
from make_golden import golden_inputs

from mjlab_amd import mjcf, robots
from mjlab_amd.sim import Simulation, SimulationCfg

NAMES = ["M + factor + qacc_smooth", "rows + warm start", "(first update: bookkeeping)", "(update: bookkeeping)", "factor + solve", "M v, J v, Gauss", "line search", "advance + termination tests", "iterations"]
for nworld in (1024, 4096):
  for lsp in (True, False):
    model = copy.deepcopy(robots.load_model("g1_velocity_flat"))
    model.opt.cone = mjcf.CONE_ELLIPTIC
    sim = Simulation(nworld, SimulationCfg(njmax=300, ls_parallel=lsp, use_graph=False), model, "cuda:0")
    qpos, qvel, ctrl = golden_inputs(model, nworld, 5)
    for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
      getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    for _ in range(20):
      sim.step()
    sim.data.profile.zero_()
    n = 40
    for _ in range(n):
      sim.step()
    torch.cuda.synchronize()
    p = sim.data.profile.cpu().numpy().reshape(nworld, 64)[:, 48:64] / n
    print(f"\n{nworld} worlds, ls_parallel={lsp}: cycles per world-step (mean over worlds | max), {p[:, 8].mean():.2f} iterations")
    for k, name in enumerate(NAMES[:8]):
      print(f"  {name:28s} {p[:, k].mean():10.0f} | {p[:, k].max():10.0f}")
    for k, name in ((10, "  (updates: rows + list)"), (11, "  (updates: J^T f, tiles, H store)")):
      print(f"  {name:34s} {p[:, k].mean():10.0f} | {p[:, k].max():10.0f}")
    print(f"  {'total':28s} {(p[:, :8].sum(axis=1) + p[:, 10] + p[:, 11]).mean():10.0f} | {(p[:, :8].sum(axis=1) + p[:, 10] + p[:, 11]).max():10.0f}")
