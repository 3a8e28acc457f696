"""profiles/parity_literals.json: every literal of the GPU parity gate (tests/test_gpu_parity_gate.py) and of smoke()
(__graft_entry__.py), the spread of the statistic it bounds over every gate report on record, and the commit that last touched it.

VERDICT round 5, item 5: the literals moved in round 5 (for a measured reason: tools/parity_seeds.py); from now on moving one is a
visible act -- tests/test_parity_literals.py (CPU) fails when a literal in the test files is WIDER than this file says, and this file
changes only by running this script, whose diff shows old value, new value and the measurements next to it.

  python tools/make_parity_literals.py            (rewrites profiles/parity_literals.json from the current sources and reports)
"""
from __future__ import annotations

import collections
import glob
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tools"))

KIN = ("xpos", "xquat", "xipos", "subtree_com", "geom_xpos", "site_xpos")
VEL = ("cvel", "qfrc_bias", "actuator_force", "qfrc_smooth", "qacc_smooth")
# literal key -> (fields, statistic index 0 median / 1 p99 / 2 max) in a report's first table
STAT = {"kin_max": (KIN, 2), "qM_max": (("qM",), 2), "vel_max": (VEL, 2), "efc_J_max": (("efc_J",), 2), "efc_J_p99": (("efc_J",), 1),
        "efc_pos_abs_max": (("efc_pos_abs_m",), 2), "efc_D_p99": (("efc_D",), 1), "efc_aref_p99": (("efc_aref",), 1), "qacc_med": (("qacc", "qfrc_constraint"), 0),
        "qacc_p99": (("qacc",), 1), "qacc_max": (("qacc",), 2), "qfc_p99": (("qfrc_constraint",), 1), "qfc_max": (("qfrc_constraint",), 2),
        "step_qpos_p99": (("step_qpos",), 1), "step_qpos_max": (("step_qpos",), 2), "step_qvel_p99": (("step_qvel",), 1), "step_qvel_max": (("step_qvel",), 2)}
LOWER = ("same_frac",)  # literals that are FLOORS (a smaller value is the wider one); everything else is a ceiling


def scene_class(scene: str) -> str:
  return "rough" if scene.endswith("rough") else ("tracking" if "tracking" in scene else "flat")


def reports():
  """Every recorded gate report: (class, {field: (median, p99, max)}, {regular field: max}, unexplained_max, off_count, {elem field: fraction})."""
  files = sorted(set(glob.glob(str(ROOT / "profiles" / "r0*" / "parity_gate*.txt")) + glob.glob(str(ROOT / "profiles" / "r0*" / "*" / "parity_gate*.txt"))))
  out = []
  for f in files:
    cur = None
    for line in open(f, errors="ignore"):
      m = re.match(r"== (\w+): (\d+) worlds after (\d+) control steps.*oracle (f\d+)", line)
      if m:
        cur = {"file": str(Path(f).relative_to(ROOT)), "cls": scene_class(m.group(1)), "scene": m.group(1), "fields": {}, "regular": {}, "elem": {}, "table": 0}
        out.append(cur)
        continue
      if cur is None:
        continue
      if line.strip().startswith("field "):
        cur["table"] = 1
      elif line.strip().startswith("element-wise"):
        cur["table"] = 2
      m = re.match(r"\s+(\w+)\s+([\d.e+-]+)\s+([\d.e+-]+)\s+([\d.e+-]+)(?:\s+([\d.]+))?", line)
      if m and cur["table"] == 1:
        cur["fields"][m.group(1)] = tuple(float(m.group(k)) for k in (2, 3, 4))
      elif m and cur["table"] == 2 and m.group(5):
        cur["elem"][m.group(1)] = float(m.group(5))
      m = re.search(r"below its cap: efc_J ([\d.e+-]+), qacc ([\d.e+-]+), qfrc_constraint ([\d.e+-]+), step_qpos ([\d.e+-]+), step_qvel ([\d.e+-]+)", line)
      if m:
        cur["regular"] = dict(zip(("efc_J", "qacc", "qfrc_constraint", "step_qpos", "step_qvel"), map(float, m.groups())))
      m = re.search(r"qacc off by more than 1e-5 in (\d+) worlds.*unexplained \d+ \(worst ([\d.e+-]+)\)", line)
      if m:
        cur["off"], cur["unexplained_max"] = int(m.group(1)), float(m.group(2))
  return out


def spread(vals):
  vals = sorted(vals)
  if not vals:
    return None
  return {"reports": len(vals), "min": vals[0], "median": vals[len(vals) // 2], "max": vals[-1]}


def blame(path: Path, lineno: int) -> str:
  r = subprocess.run(["git", "blame", "-L", f"{lineno},{lineno}", "--porcelain", str(path)], capture_output=True, text=True, cwd=ROOT)
  h = r.stdout.split()[0] if r.returncode == 0 and r.stdout else "uncommitted"
  return "uncommitted" if set(h) == {"0"} else h[:7]


def line_of(src: list[str], block_start: int, key: str) -> int:
  for i in range(block_start, min(len(src), block_start + 40)):
    if re.search(rf'["\']?{re.escape(key)}["\']?\s*[:=]', src[i]):
      return i + 1
  return block_start + 1


def main():
  import test_gpu_parity_gate as gate

  import __graft_entry__ as entry

  reps = reports()
  gate_path = ROOT / "tests" / "test_gpu_parity_gate.py"
  src = gate_path.read_text().split("\n")
  start = {name: next(i for i, l in enumerate(src) if l.startswith(name + " =")) for name in ("FLAT", "ROUGH", "TRACKING", "GRID", "REGULAR", "ELEM_FLOOR", "FRICTIONLOSS")}
  out = {"_about": "Frozen literals of the GPU parity gate and of smoke(); written by tools/make_parity_literals.py, enforced by tests/test_parity_literals.py "
                   "(a literal in the sources may be TIGHTER than here, never wider).  measured = the bounded statistic over every gate report under profiles/ "
                   "(per scene class; relative error per world unless the name says otherwise).", "reports_on_record": len(reps), "literals": {}}
  L = out["literals"]
  for name, cls in (("FLAT", "flat"), ("ROUGH", "rough"), ("TRACKING", "tracking"), ("GRID", None), ("FRICTIONLOSS", "flat")):
    table = getattr(gate, name)
    for k, v in table.items():
      if not isinstance(v, (int, float)):
        continue
      if name == "TRACKING" and gate.FLAT.get(k) == v:
        continue  # inherited from FLAT: frozen there
      e = {"value": v, "bound": "floor" if k in LOWER else "ceiling", "last_changed": blame(gate_path, line_of(src, start[name], k))}
      sel = [r for r in reps if cls is None or r["cls"] == cls]
      if k in STAT:
        fields, idx = STAT[k]
        e["measured"] = spread([max(r["fields"][f][idx] for f in fields if f in r["fields"]) for r in sel if any(f in r["fields"] for f in fields)])
      elif k == "unexplained_max":
        e["measured"] = spread([r["unexplained_max"] for r in sel if "unexplained_max" in r])
      elif k == "off_frac":
        e["measured"] = spread([r["off"] / 1024.0 for r in sel if "off" in r])
      L[f"{name}.{k}"] = e
  for cls, table in gate.REGULAR.items():
    for k, v in table.items():
      L[f"REGULAR.{cls}.{k}"] = {"value": v, "bound": "ceiling", "last_changed": blame(gate_path, line_of(src, start["REGULAR"], cls)),
                                 "measured": spread([r["regular"][k] for r in reps if r["cls"] == cls and k in r["regular"]])}
  for cls, table in gate.ELEM_FLOOR.items():
    for k, v in table.items():
      L[f"ELEM_FLOOR.{cls}.{k}"] = {"value": v, "bound": "floor", "last_changed": blame(gate_path, line_of(src, start["ELEM_FLOOR"], cls))}
  ep = ROOT / "__graft_entry__.py"
  esrc = ep.read_text().split("\n")
  eline = next(i for i, l in enumerate(esrc) if l.startswith("SMOKE_TOLERANCES")) + 1
  for par, tols in entry.SMOKE_TOLERANCES:
    for k, v in tols.items():
      L[f"SMOKE.{'grid' if par else 'exact'}.{k}"] = {"value": v, "bound": "ceiling", "last_changed": blame(ep, eline),
                                                      "measured": "GPUTEST_r05 smoke tail: exact qacc 2.9e-6 / qvel 4.7e-6, grid qacc 5.9e-6 / qvel 5.9e-5 / qpos 5.6e-7"}
  (ROOT / "profiles" / "parity_literals.json").write_text(json.dumps(out, indent=1) + "\n")
  print(f"{len(L)} literals, {len(reps)} reports -> profiles/parity_literals.json")


if __name__ == "__main__":
  main()
