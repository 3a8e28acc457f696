"""Where do the bytes go?  Expected HBM traffic per array and stage against the measured counters.

  python tools/traffic_budget.py [profiles/<tag>/hbm_traffic.csv] [scene]       -> table on stdout (committed as profiles/<tag>/traffic_budget.md)

EXPECTED: every array a stage reads from / writes to global memory (table below: read off the stage sources,
mjlab_amd/csrc/stage_*.h), sized from the model (include/mjlab_fields.h counts) and the measured mean row / contact counts of the
benchmark's rollout (nefc 31.5, ncon 9.6 per world: profiles/r03_v12/phases.txt); row arrays are counted at the rows in use.
Reads that are served by the L2 (an array written earlier in the same launch by the same CU's XCD, re-read within microseconds)
never reach HBM: the columns give the bytes REQUESTED, and FETCH_SIZE x 2 (calibrated, profiles/calibration.json) counts what
missed.  Writes all reach HBM eventually (WRITE_SIZE, exact).

MEASURED: per stage kernel from the --pmc passes of a build with one kernel per stage (tools/gpu_profile.sh runs the stage
kernels in its informational pass), per launch of `nworld` worlds.

What the table answers: which arrays make up the 7.4 x between the public contract (10 156 B per world and step) and the
traffic, and how much of the written bytes nothing outside the step ever reads.
"""

from __future__ import annotations

import csv
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

NEFC, NCON, NITER = 31.5, 9.6, 3.93  # measured means of the benchmark rollout (G1 velocity-flat, 4096 worlds)

# stage -> (reads, writes): mjData arrays by name; "J*k" = the row array re-read k times (served by L2 after the first)
STAGES = {
  "k_position": (["qpos", "qvel", "sh_qpos", "sh_qvel"],
                 ["xpos", "xquat", "xmat", "xipos", "ximat", "xanchor", "xaxis", "geom_xpos", "geom_xmat", "site_xpos", "site_xmat", "subtree_com", "cinert", "cdof",
                  "qM", "xorigin", "geom_xrel", "subtree_crel", "xipos_rel"]),
  "k_collision": (["geom_xrel", "geom_xmat", "xorigin"],
                  ["contact_dist", "contact_pos", "contact_frame", "contact_includemargin", "contact_friction", "contact_solref", "contact_solimp", "contact_dim",
                   "contact_geom", "contact_efc_address", "contact_prel", "ncon"]),
  "k_velocity": (["qpos", "qvel", "ctrl", "qfrc_applied", "xfrc_applied", "cinert", "cdof", "subtree_crel", "xipos_rel"],
                 ["cvel", "cdof_dot", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "actuator_force", "qfrc_smooth"]),
  "k_constraint": (["qpos", "qvel", "cdof", "subtree_crel", "contact_dist", "contact_frame", "contact_includemargin", "contact_friction", "contact_solref", "contact_solimp",
                    "contact_dim", "contact_geom", "contact_prel"],
                   ["efc_J", "efc_pos", "efc_margin", "efc_D", "efc_aref", "efc_type", "efc_id", "contact_efc_address", "nefc", "nf", "sensordata"]),
  "k_solve_integrate<36>": (["qM", "qfrc_smooth", "qacc_warmstart", "efc_D", "efc_aref", "qvel", "qpos", "actuator_force", "nefc"],
                            ["qacc_smooth", "qacc", "qacc_warmstart", "qfrc_constraint", "efc_force", "qvel", "qpos", "time", "solver_niter"]),
}
J_PASSES = 1 + 1 + 2 * NITER  # warm start (one pass, two vectors), initial Hessian pass, per iteration J v + J^T f / Hessian
# arrays the reference reads through sim.data (SURVEY 8b minimum export set) or that are mjData state
PUBLIC = {"qpos", "qvel", "qacc", "qacc_warmstart", "xpos", "xquat", "xmat", "xipos", "subtree_com", "cvel", "geom_xpos", "geom_xmat", "site_xpos", "site_xmat",
          "sensordata", "actuator_force", "time"}


def sizes(scene: str) -> dict:
  """bytes per world of every data field (rows in use for the row arrays)."""
  from mjlab_amd import _abi, native, robots

  m = robots.load_model(scene)
  try:
    fields = native.layouts()[1]
  except Exception:  # noqa: BLE001
    raise SystemExit("needs the built library for the field catalogue (python -c 'import __graft_entry__ as g; g.build()')")
  out = {}
  for f in fields:
    n = _abi.count_of(f.count, m, 1, 1)
    if f.count == "njmaxnv":
      n = NEFC * m.nv
    elif f.count == "njmax":
      n = NEFC
    elif f.count == "nconmax":
      n = NCON
    out[f.name] = 4.0 * n * f.ncol
  ns = int(getattr(m, "nstaticgeom", 0))
  for k in ("geom_xpos", "geom_xmat", "geom_xrel"):  # static geoms are posed once, not per pass
    out[k] *= (m.ngeom - ns) / max(m.ngeom, 1)
  out["site_xpos"] *= (m.nsite - int(getattr(m, "nstaticsite", 0))) / max(m.nsite, 1)
  out["site_xmat"] *= (m.nsite - int(getattr(m, "nstaticsite", 0))) / max(m.nsite, 1)
  return out


def main() -> None:
  path = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "profiles" / "r03_v12" / "hbm_traffic.csv"
  scene = sys.argv[2] if len(sys.argv) > 2 else "g1_velocity_flat"
  nworld = 4096
  meas = {r["kernel"]: r for r in csv.DictReader(open(path))}
  fcol = next(k for k in next(iter(meas.values())) if k.startswith("fetch_bytes_per_launch"))
  wcol = next(k for k in next(iter(meas.values())) if k.startswith("write_bytes_per_launch"))
  sz = sizes(scene)
  print(f"# Traffic budget, {scene}, per world and pass (bytes); measured: {path.relative_to(ROOT) if path.is_absolute() else path}\n")
  print("| stage | requested reads | of which J re-reads (L2) | expected writes | of which nothing outside the step reads | measured fetch (x2 cal.) | measured write | write - expected |")
  print("|---|---|---|---|---|---|---|---|")
  tot = [0.0] * 7
  private_rows = []
  for stage, (reads, writes) in STAGES.items():
    r = sum(sz[a] for a in reads)
    jre = 0.0
    if stage.startswith("k_solve"):
      jre = sz["efc_J"] * J_PASSES
      r += jre
    w = sum(sz[a] for a in writes)
    priv = sum(sz[a] for a in writes if a not in PUBLIC)
    private_rows += [(stage, a, sz[a]) for a in writes if a not in PUBLIC]
    mf = float(meas[stage][fcol]) / nworld if stage in meas else float("nan")
    mw = float(meas[stage][wcol]) / nworld if stage in meas else float("nan")
    print(f"| {stage} | {r:.0f} | {jre:.0f} | {w:.0f} | {priv:.0f} | {mf:.0f} | {mw:.0f} | {mw - w:.0f} |")
    for i, v in enumerate((r, jre, w, priv, mf, mw, mw - w)):
      tot[i] += v
  print(f"| **sum of the five stages** | {tot[0]:.0f} | {tot[1]:.0f} | {tot[2]:.0f} | {tot[3]:.0f} | {tot[4]:.0f} | {tot[5]:.0f} | {tot[6]:.0f} |")
  ck = next((k for k in meas if k.startswith("k_control_step")), None)
  if ck:
    f, w = float(meas[ck][fcol]) / nworld / 5, float(meas[ck][wcol]) / nworld / 5
    print(f"| {ck}, per pass (5 passes per launch) | | | | | {f:.0f} | {w:.0f} | |")
  print("\nWritten arrays that nothing outside the step reads (hand-over between stages + diagnostics), largest first:\n")
  print("| stage | array | bytes per world and pass |\n|---|---|---|")
  for stage, a, b in sorted(private_rows, key=lambda x: -x[2])[:14]:
    print(f"| {stage} | {a} | {b:.0f} |")
  print("\n`write - expected` is what the arrays do not explain: register spills to scratch (39 spilled VGPRs x 256 B per wave = 10 KB per spill "
        "round of a wave; private_segment 120 B per lane) and partial-line write amplification.")


if __name__ == "__main__":
  main()
