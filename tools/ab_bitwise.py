"""Bitwise comparison of library variants on the GPU box: the same seeded rollout (G1 velocity-flat, task events, 60 control steps)
through every gpurun_prof/ab_*.so, in a subprocess each; prints a checksum of the final state per library.
  python tools/ab_bitwise.py            (driver)      python tools/ab_bitwise.py --one      (one library: MJLAB_AMD_LIB)"""
import glob
import hashlib
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

if "--one" in sys.argv:
  import torch

  from mjlab_amd import robots
  from mjlab_amd.rollout import VELOCITY_TASK_EVENTS, PhysicsRollout, g1_action_scale
  from mjlab_amd.sim import Simulation, SimulationCfg

  scene = os.environ.get("SCENE", "g1_velocity_flat")
  model = robots.load_model(scene)
  sim = Simulation(1024, SimulationCfg(njmax=300), model, "cuda:0")
  roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=42, substeps_per_call=4, control_kernel=True, **VELOCITY_TASK_EVENTS["g1"])
  for _ in range(60):
    roll.step(roll.random_action())
  torch.cuda.synchronize()
  h = hashlib.sha256()
  for f in ("qpos", "qvel", "qacc", "qacc_warmstart", "efc_force"):
    h.update(getattr(sim.data, f).cpu().numpy().tobytes())
  print("CHECKSUM", h.hexdigest()[:16], float(sim.data.qpos.abs().sum()))
else:
  for lib in sorted(glob.glob(str(ROOT / "gpurun_prof" / "ab_*.so"))):
    r = subprocess.run([sys.executable, __file__, "--one"], env={**os.environ, "MJLAB_AMD_LIB": lib}, capture_output=True, text=True, timeout=300)
    line = next((x for x in r.stdout.splitlines() if x.startswith("CHECKSUM")), r.stderr[-300:])
    print(Path(lib).name, line)
