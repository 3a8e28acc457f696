"""Record golden vectors from the PINNED upstream engine (mujoco_warp) -- to be run on a machine that has the reference's
dependencies; it cannot run in the build container.

The reference's hot path is ``mjwarp.step`` / ``mjwarp.forward`` (reference src/mjlab/sim/sim.py:136,139,187,195), pinned at
``mujoco_warp @ 486642c3fa262a989b482e0e506716d5793d61a9`` + ``mujoco 3.3.7.dev811775910`` (reference pyproject.toml:94-96).
Neither is installable here, so the oracle in ``oracle/`` is "parity unpinned" (DESIGN.md section 3).  This script closes that
gap in ONE command on a machine that has them:

  python tools/dump_mjwarp_reference.py --reference /path/to/mjlab            # -> tests/golden_upstream/*.npz

``--dry-run`` runs the SAME code path here -- the reference's task registry, Scene and MujocoCfg over this repository's
``mujoco`` shim, and tools/fake_mjwarp.py (``mujoco_warp`` + ``warp`` backed by the fp32 oracle) in place of the engine -- and
writes tests/golden_upstream_dryrun/ (committed: the GPU box has no reference tree, and the HIP-side consumers read it there).  tests/test_upstream_dryrun.py does that in CI and feeds the output through
the very comparison code the upstream tests use, so every line of this tool and of its consumers has executed before the real run.

Per scene it writes two files, each with ``ls_parallel`` on (the reference's setting, sim/sim.py:89,111) AND off (the seeded-state file also
with ``MujocoCfg.cone = "elliptic"``: records ``elllsp1_*`` / ``elllsp0_*``, what would pin this repository's elliptic path):

  <scene>.npz            the seeded states of tools/make_golden.py (4 worlds), same keys as tests/golden/<scene>.npz
  <scene>_rollout.npz    the ROLLOUT STATES the parity gate uses: tests/golden/rollout_states_<scene>.npz (256 worlds reached
                         by a GPU rollout with random actions, falls and resets; exported by tools/export_rollout_states.py and
                         committed), with per-world model fields (friction, torso com, joint zero offsets) applied

and in both the upstream mjModel arrays of every field in this repository's catalogue (``model_<field>``: what
tests/test_golden.py::test_compiled_model_matches_upstream compares this repository's MJCF compiler against).

Keys: ``in_<qpos|qvel|ctrl|qacc_warmstart>``, ``model_<field>``, ``dr_<field>`` (per-world model fields, rollout file), and
for P in (``lsp1``, ``lsp0``): ``P_fwd_<field>`` after forward(), ``P_step_<field>`` after nstep x step() + forward().  Fields =
make_golden.OUT_FIELDS + the intermediates ``qacc_smooth qfrc_smooth qfrc_constraint qM nefc ncon solver_niter
efc_J efc_D efc_aref efc_pos efc_force contact_dist contact_pos contact_frame contact_geom`` (whatever of them the pinned
engine exposes per world; arrays it keeps in another layout are stored raw under ``P_raw_<name>`` with their shape).

tests/test_golden.py consumes all of it: the oracle (CPU) and the HIP path (GPU) under both ``literal_termination`` and both
``warmstart_at_advance`` settings, at north_star's tolerance.
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
INTERMEDIATES = ("qacc_smooth", "qfrc_smooth", "qfrc_constraint", "qM", "nefc", "ncon", "solver_niter")
EFC = ("J", "D", "aref", "pos", "force")
CONTACT = ("dist", "pos", "frame", "geom")
NSTEP = 5


def model_arrays(mjm) -> dict:
  """Every mjModel array this repository's catalogue names (include/mjlab_fields.h), as upstream compiled it."""
  from mjlab_amd import native  # layout only: no GPU needed

  out = {}
  try:
    fields = [f.name for f in native.layouts()[0]]
  except Exception:  # noqa: BLE001  (no built library on that machine: fall back to the header)
    import re

    fields = re.findall(r"X\((\w+), ", (ROOT / "include" / "mjlab_fields.h").read_text().split("MJLAB_DATA_REAL_FIELDS")[0])
  for f in fields:
    if hasattr(mjm, f):
      out["model_" + f] = np.asarray(getattr(mjm, f)).copy()
  for f in ("nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite", "nsensor"):
    out["model_" + f] = np.asarray(getattr(mjm, f))
  for f in ("timestep", "gravity", "impratio", "tolerance", "ls_tolerance", "iterations", "ls_iterations", "integrator", "cone", "solver"):
    out["model_opt_" + f] = np.asarray(getattr(mjm.opt, f))
  out["model_stat_meaninertia"] = np.asarray(mjm.stat.meaninertia)
  return out


def record(d, prefix: str, nworld: int, fields) -> dict:
  rec = {}

  def put(key, arr):
    a = np.array(arr.numpy() if hasattr(arr, "numpy") else arr)  # a COPY: host-resident engines hand out live views
    if a.ndim >= 1 and a.shape[0] == nworld:
      rec[f"{prefix}_{key}"] = a
    else:  # a layout this script does not know (pooled constraints of older engines, ...): keep it, labelled
      rec[f"{prefix.split('_')[0]}_raw_{prefix.split('_', 1)[1]}_{key}"] = a

  for f in fields:
    if hasattr(d, f):
      put(f, getattr(d, f))
  for f in EFC:
    if hasattr(d, "efc") and hasattr(d.efc, f):
      put("efc_" + f, getattr(d.efc, f))
  for f in CONTACT:
    if hasattr(d, "contact") and hasattr(d.contact, f):
      put("contact_" + f, getattr(d.contact, f))
  return rec


def run_case(mjwarp, wp, mjm, mjd, cfg, states: dict, dr: dict, ls_parallel: bool, fields, tag: str = "") -> dict:
  nworld = states["qpos"].shape[0]
  m = mjwarp.put_model(mjm)
  m.opt.ls_parallel = ls_parallel  # reference src/mjlab/sim/sim.py:111
  d = mjwarp.put_data(mjm, mjd, nworld=nworld, nconmax=cfg.sim.nconmax, njmax=cfg.sim.njmax)
  for f, v in dr.items():  # per-world model fields, expanded like the reference's expand_model_fields (sim/randomization.py)
    setattr(m, f, wp.array(np.ascontiguousarray(v.astype(np.float32)), dtype=getattr(m, f).dtype))
  for f, v in states.items():
    wp.copy(getattr(d, f), wp.array(np.ascontiguousarray(v.astype(np.float32))))
  p = tag + ("lsp1" if ls_parallel else "lsp0")  # tag "ell": the same scene compiled with MujocoCfg.cone = "elliptic"
  mjwarp.forward(m, d)
  rec = record(d, p + "_fwd", nworld, fields)
  for f, v in states.items():  # one step()-chain from the SAME state and warm start (forward() overwrote qacc_warmstart)
    wp.copy(getattr(d, f), wp.array(np.ascontiguousarray(v.astype(np.float32))))
  for _ in range(NSTEP):
    mjwarp.step(m, d)
  mjwarp.forward(m, d)
  rec.update(record(d, p + "_step", nworld, fields))
  return rec


def load_engine(dry_run: bool, reference_src: Path):
  """-> (mujoco, mjwarp, wp).  Real run: the pinned wheels.  ``--dry-run``: this repository's ``mujoco`` shim, the gym / warp /
  prettytable stubs of tools/reference_env.py and tools/fake_mjwarp.py (``mujoco_warp`` backed by the fp32 CPU oracle), so that
  every line below executes in the build container; its output pins nothing to upstream."""
  if dry_run:
    import fake_mjwarp
    import reference_env

    reference_env.install_stubs(reference_src)
    mjwarp, wp = fake_mjwarp.install()
    import mujoco

    return mujoco, mjwarp, wp
  try:
    import mujoco
    import mujoco_warp as mjwarp
    import warp as wp
  except ImportError as e:  # the expected outcome in the build container
    raise SystemExit(f"upstream engine not importable here ({e}); run this where mjlab's pinned deps are installed, or pass --dry-run")
  sys.path.insert(0, str(reference_src))
  return mujoco, mjwarp, wp


def main() -> None:
  ap = argparse.ArgumentParser()
  ap.add_argument("--reference", required=True, help="checkout of mujocolab/mjlab with its pinned deps installed")
  ap.add_argument("--nworld", type=int, default=4)
  ap.add_argument("--seed", type=int, default=7)
  ap.add_argument("--out", default=None, help="default: tests/golden_upstream (tests/golden_upstream_dryrun with --dry-run)")
  ap.add_argument("--device", default=None, help="default: cuda:0 (cpu with --dry-run)")
  ap.add_argument("--dry-run", action="store_true", help="run the whole tool over the oracle-backed fake engine (CI; pins nothing)")
  ap.add_argument("--rollout-worlds", type=int, default=0, help="use only the first N rollout states (0 = all)")
  args = ap.parse_args()
  device = args.device or ("cpu" if args.dry_run else "cuda:0")
  out = Path(args.out or (ROOT / "tests" / ("golden_upstream_dryrun" if args.dry_run else "golden_upstream")))

  mujoco, mjwarp, wp = load_engine(args.dry_run, Path(args.reference) / "src")
  # the reference's own way from a task id to its configuration (scripts/train.py:18,129) and from there to the compiled model
  # (envs/manager_based_env.py:63-70): importing mjlab.tasks registers every task with gymnasium
  import mjlab.tasks  # type: ignore  # noqa: F401
  from mjlab.scene import Scene  # type: ignore
  from mjlab.third_party.isaaclab.isaaclab_tasks.utils.parse_cfg import load_cfg_from_registry  # type: ignore

  from make_golden import OUT_FIELDS, golden_inputs, models  # same seeded inputs as the oracle fixtures

  ours = models()
  out.mkdir(parents=True, exist_ok=True)
  fields = tuple(OUT_FIELDS) + INTERMEDIATES
  for name, task in SCENES.items():
    cfg = load_cfg_from_registry(task, "env_cfg_entry_point")
    cfg.scene.num_envs = args.nworld
    scene = Scene(cfg.scene, device=device)
    cfg.sim.mujoco.edit_spec(scene.spec)  # reference envs/manager_based_env.py:63: the task's MujocoCfg goes into the spec before compile()
    mjm = scene.compile()
    mjd = mujoco.MjData(mjm)
    mujoco.mj_forward(mjm, mjd)
    marr = model_arrays(mjm)
    # ---- (1) the seeded states of tests/golden/<scene>.npz
    qpos, qvel, ctrl = golden_inputs(ours[name], args.nworld, args.seed)
    assert qpos.shape[1] == mjm.nq and qvel.shape[1] == mjm.nv, "model mismatch between the two compilers"
    states = {"qpos": qpos, "qvel": qvel, "ctrl": ctrl}
    rec = {"in_" + k: v for k, v in states.items()}
    rec.update(marr, nstep=np.array(NSTEP), dry_run=np.array(int(args.dry_run)))
    for lsp in (True, False):
      rec.update(run_case(mjwarp, wp, mjm, mjd, cfg, states, {}, lsp, fields))
    # ---- (1b) the same scene and states with elliptic friction cones (MujocoCfg.cone, reference sim/sim.py:49,52; no task sets it: this
    # is what would pin this repository's elliptic path, round 5): records under "elllsp1_*" / "elllsp0_*"
    import copy

    cfg_e = copy.deepcopy(cfg)  # (the velocity tasks share one module-level SimulationCfg: never edit it in place)
    cfg_e.sim.mujoco.cone = "elliptic"
    scene_e = Scene(cfg_e.scene, device=device)
    cfg_e.sim.mujoco.edit_spec(scene_e.spec)
    mjm_e = scene_e.compile()
    mjd_e = mujoco.MjData(mjm_e)
    mujoco.mj_forward(mjm_e, mjd_e)
    rec["ell_model_opt_cone"] = np.asarray(mjm_e.opt.cone)
    for lsp in (True, False):
      rec.update(run_case(mjwarp, wp, mjm_e, mjd_e, cfg_e, states, {}, lsp, fields, tag="ell"))
    np.savez_compressed(out / f"{name}.npz", **rec)
    print("wrote", out / f"{name}.npz")
    # ---- (2) the rollout states of the parity gate, exported from a GPU run and committed
    src = ROOT / "tests" / "golden" / f"rollout_states_{name}.npz"
    if not src.exists():
      print("no", src, "(tools/export_rollout_states.py on the GPU box): rollout file skipped")
      continue
    z = np.load(src)
    sl = slice(0, args.rollout_worlds or None)
    states = {k: z[k][sl] for k in ("qpos", "qvel", "ctrl", "qacc_warmstart")}
    dr = {k[3:]: z[k][sl] for k in z.files if k.startswith("dr_")}
    rec = {"in_" + k: v for k, v in states.items()}
    rec.update({"dr_" + k: v for k, v in dr.items()})
    rec.update(marr, nstep=np.array(NSTEP), dry_run=np.array(int(args.dry_run)))
    for lsp in (True, False):
      rec.update(run_case(mjwarp, wp, mjm, mjd, cfg, states, dr, lsp, fields))
    np.savez_compressed(out / f"{name}_rollout.npz", **rec)
    print("wrote", out / f"{name}_rollout.npz")


if __name__ == "__main__":
  main()
