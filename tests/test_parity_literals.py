"""The parity gate's literals are frozen in profiles/parity_literals.json: a literal in
tests/test_gpu_parity_gate.py or __graft_entry__.py may be tighter than the file says, never wider, and a literal the file does not
know fails too.  Loosening one means running tools/make_parity_literals.py -- a diff that shows the old value, the new one and the
measured spread of the statistic it bounds (north_star: "within 1e-5 rel fp32")."""

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _current():
  import test_gpu_parity_gate as gate

  import __graft_entry__ as entry

  cur = {}
  for name in ("FLAT", "ROUGH", "TRACKING", "GRID", "FRICTIONLOSS"):
    for k, v in getattr(gate, name).items():
      if isinstance(v, (int, float)) and not (name == "TRACKING" and gate.FLAT.get(k) == v):
        cur[f"{name}.{k}"] = v
  for cls, table in gate.REGULAR.items():
    for k, v in table.items():
      cur[f"REGULAR.{cls}.{k}"] = v
  for cls, table in gate.ELEM_FLOOR.items():
    for k, v in table.items():
      cur[f"ELEM_FLOOR.{cls}.{k}"] = v
  for par, tols in entry.SMOKE_TOLERANCES:
    for k, v in tols.items():
      cur[f"SMOKE.{'grid' if par else 'exact'}.{k}"] = v
  return cur


def test_no_literal_is_wider_than_the_frozen_file():
  frozen = json.loads((ROOT / "profiles" / "parity_literals.json").read_text())["literals"]
  cur = _current()
  assert set(cur) == set(frozen), sorted(set(cur) ^ set(frozen))
  wider = {}
  for k, v in cur.items():
    f = frozen[k]
    if (f["bound"] == "ceiling" and v > f["value"]) or (f["bound"] == "floor" and v < f["value"]):
      wider[k] = (v, f["value"])
  assert not wider, f"literals wider than profiles/parity_literals.json (run tools/make_parity_literals.py to move them on purpose): {wider}"


def test_north_star_tolerance_is_the_line_of_the_bulk():
  """What must never move: the median (1e-5), the flat scenes' p99 (1.5e-5) and the kinematic literals (1e-6)."""
  frozen = json.loads((ROOT / "profiles" / "parity_literals.json").read_text())["literals"]
  assert frozen["FLAT.qacc_med"]["value"] <= 1e-5 and frozen["FLAT.qacc_p99"]["value"] <= 1.5e-5 and frozen["FLAT.kin_max"]["value"] <= 1e-6
  assert frozen["SMOKE.exact.qacc"]["value"] <= 1e-5
