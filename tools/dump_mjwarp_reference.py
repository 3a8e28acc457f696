"""Record golden step vectors from the PINNED upstream engine (mujoco_warp) -- to be run on a
machine that has the reference's dependencies; it cannot run in the build container.

The reference's hot path is ``mjwarp.step`` / ``mjwarp.forward``
(reference src/mjlab/sim/sim.py:136,139,187,195), pinned at
``mujoco_warp @ 486642c3fa262a989b482e0e506716d5793d61a9`` + ``mujoco 3.3.7.dev811775910``
(reference pyproject.toml:94-96).  Neither is installable here, so the oracle in
``oracle/`` is "parity unpinned" (DESIGN.md section 3).  This script closes that gap without
changing the build: it compiles the same scenes with upstream ``mujoco`` through the
reference's own ``Scene`` / task configuration, seeds the same states as
``tools/make_golden.py`` and writes ``tests/golden_upstream/<scene>.npz`` in the same format
as ``tests/golden/*.npz`` (``in_{qpos,qvel,ctrl}``, ``fwd_<field>`` after ``forward``,
``step_<field>`` after ``nstep`` x ``step`` + ``forward``; fields = make_golden.OUT_FIELDS).

  python tools/dump_mjwarp_reference.py --reference /path/to/mjlab

Once such files exist, ``tests/test_golden.py::test_*_upstream`` compares both the oracle and
the HIP path against them at the north_star tolerance (1e-5 relative on state, fp32).
"""

from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

SCENES = {
  "g1_velocity_flat": "Mjlab-Velocity-Flat-Unitree-G1",
  "g1_tracking_flat": "Mjlab-Tracking-Flat-Unitree-G1",
  "go1_velocity_flat": "Mjlab-Velocity-Flat-Unitree-Go1",
}


def main() -> None:
  ap = argparse.ArgumentParser()
  ap.add_argument("--reference", required=True, help="checkout of mujocolab/mjlab with its pinned deps installed")
  ap.add_argument("--nworld", type=int, default=4)
  ap.add_argument("--seed", type=int, default=7)
  ap.add_argument("--out", default=str(ROOT / "tests" / "golden_upstream"))
  args = ap.parse_args()

  try:
    import mujoco
    import mujoco_warp as mjwarp
    import warp as wp
  except ImportError as e:  # the expected outcome in the build container
    raise SystemExit(f"upstream engine not importable here ({e}); run this where mjlab's pinned deps are installed")

  sys.path.insert(0, str(Path(args.reference) / "src"))
  from mjlab.scene import Scene  # type: ignore
  from mjlab.tasks.registry import load_cfg_from_registry  # type: ignore

  from make_golden import OUT_FIELDS, golden_inputs, models  # same seeded inputs as the oracle fixtures

  ours = models()
  out = Path(args.out)
  out.mkdir(parents=True, exist_ok=True)
  for name, task in SCENES.items():
    cfg = load_cfg_from_registry(task, "env_cfg_entry_point")
    cfg.scene.num_envs = args.nworld
    scene = Scene(cfg.scene, device="cuda:0")
    cfg.sim.mujoco.edit_spec(scene.spec) if hasattr(cfg.sim.mujoco, "edit_spec") else None
    mjm = scene.compile()
    mjd = mujoco.MjData(mjm)
    mujoco.mj_forward(mjm, mjd)
    qpos, qvel, ctrl = golden_inputs(ours[name], args.nworld, args.seed)
    assert qpos.shape[1] == mjm.nq and qvel.shape[1] == mjm.nv, "model mismatch between the two compilers"
    m = mjwarp.put_model(mjm)
    d = mjwarp.put_data(mjm, mjd, nworld=args.nworld, nconmax=cfg.sim.nconmax, njmax=cfg.sim.njmax)
    wp.copy(d.qpos, wp.array(qpos.astype(np.float32)))
    wp.copy(d.qvel, wp.array(qvel.astype(np.float32)))
    wp.copy(d.ctrl, wp.array(ctrl.astype(np.float32)))
    mjwarp.forward(m, d)
    nstep = 5
    rec = {"in_qpos": qpos, "in_qvel": qvel, "in_ctrl": ctrl, "nstep": np.array(nstep)}
    for f in OUT_FIELDS + ("nefc",):
      rec["fwd_" + f] = getattr(d, f).numpy()
    for _ in range(nstep):
      mjwarp.step(m, d)
    mjwarp.forward(m, d)
    for f in OUT_FIELDS + ("nefc",):
      rec["step_" + f] = getattr(d, f).numpy()
    np.savez_compressed(out / f"{name}.npz", **rec)
    print("wrote", out / f"{name}.npz")


if __name__ == "__main__":
  main()
