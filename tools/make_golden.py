"""Generate tests/golden/*.npz: seeded inputs and fp64-oracle outputs for each scene.

The reference holds no numeric golden vectors for the step and cannot be run here (its
engine, mujoco_warp/mujoco, is not installed) -- SURVEY.md section 8c.  These fixtures are
therefore REGRESSION vectors produced by this repository's own fp64 oracle; they pin the
oracle (CPU test) and the HIP path (GPU test) to each other over time, not to upstream.
That holds doubly for the elliptic-cone fixtures (`*_elliptic.npz`, round 5): the cone model itself is restated from MuJoCo's
documentation and unpinned -- REGRESSION fixtures, not parity evidence.
"""

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from mjlab_amd import robots  # noqa: E402
from oracle.oracle import OracleSim  # noqa: E402

OUT_FIELDS = ("qpos", "qvel", "qacc", "xpos", "xquat", "cvel", "subtree_com", "sensordata", "actuator_force", "qfrc_bias")


def golden_inputs(model, nworld, seed):
  rng = np.random.default_rng(seed)
  base = model.key_qpos[0] if model.nkey else model.qpos0
  qpos = np.tile(base, (nworld, 1))
  qvel = rng.normal(0, 0.3, size=(nworld, model.nv))
  for j in range(model.njnt):
    qa = model.jnt_qposadr[j]
    if model.jnt_type[j] == 0:
      qpos[:, qa + 2] += rng.uniform(-0.04, 0.02, size=nworld)
      q = qpos[:, qa + 3 : qa + 7] + rng.normal(0, 0.05, size=(nworld, 4))
      qpos[:, qa + 3 : qa + 7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    else:
      qpos[:, qa] += rng.normal(0, 0.1, size=nworld)
  ctrl = np.zeros((nworld, model.nu))
  if model.nu:
    jn = model.actuator_trnid[:, 0]
    ctrl = qpos[:, model.jnt_qposadr[jn]] + rng.normal(0, 0.2, size=(nworld, model.nu))
  origins = getattr(model, "terrain_origins", None)
  if origins is not None:  # terrain scenes: spread the worlds over the sub-terrains (drawn last: other scenes keep their inputs)
    r, c = rng.integers(0, origins.shape[0], nworld), rng.integers(0, origins.shape[1], nworld)
    qpos[:, :3] += origins[r, c]
    qpos[:, :2] += rng.uniform(-1.5, 1.5, size=(nworld, 2))
  return qpos, qvel, ctrl


def models():
  out = {n: robots.load_model(n) for n in robots.SCENES}
  out["mixed"] = robots.mixed_model()
  out["box"] = robots.box_model()
  # elliptic friction cones (MujocoCfg.cone = "elliptic"; round 5): the G1 as the velocity task builds it, and the mixed scene with impratio 2
  import copy

  from mjlab_amd.mjcf import CONE_ELLIPTIC

  for base, name, impratio in (("g1_velocity_flat", "g1_velocity_flat_elliptic", 1.0), ("mixed", "mixed_elliptic", 2.0)):
    m = copy.deepcopy(out[base])
    m.opt.cone, m.opt.impratio = CONE_ELLIPTIC, impratio
    out[name] = m
  return out


def main():
  gold = ROOT / "tests" / "golden"
  gold.mkdir(exist_ok=True)
  only = sys.argv[1:]  # (names: regenerate these fixtures only)
  for name, model in models().items():
    if only and name not in only:
      continue
    nworld, nstep = 4, 5
    qpos, qvel, ctrl = golden_inputs(model, nworld, seed=7)
    s = OracleSim(model, nworld, njmax=300, precision="f64")
    s.qpos[:], s.qvel[:], s.ctrl[:] = qpos, qvel, ctrl
    s.forward()
    blob = {"in_qpos": qpos, "in_qvel": qvel, "in_ctrl": ctrl, "nstep": np.array(nstep)}
    for f in OUT_FIELDS + ("nefc", "ncon"):
      blob["fwd_" + f] = getattr(s, f).copy()
    s.step(nstep)
    s.forward()
    for f in OUT_FIELDS + ("nefc", "ncon"):
      blob["step_" + f] = getattr(s, f).copy()
    np.savez_compressed(gold / f"{name}.npz", **blob)
    print(name, "nefc after forward", s.nefc.ravel())


if __name__ == "__main__":
  main()
