"""Calibration factors from one tools/ubench.sh run (gpurun_out/<tag>): what FETCH_SIZE / WRITE_SIZE report for
known byte counts, and how many cycles a wave64 VALU instruction occupies a SIMD (wave-side s_memtime and SQ counters).
Writes profiles/calibration.json when called with --commit (read by tools/collect_profile.py and bench.py)."""
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def pmc(path: Path) -> dict:
  out: dict = {}
  if not path.exists():
    return out
  for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
    out.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
  return out


def main(tag: str, commit: bool = False) -> None:
  src = ROOT / "gpurun_out" / tag
  rows = [json.loads(x) for x in open(src / "ubench.jsonl") if x.startswith("{")]
  fetch, write = pmc(src / "pmc_FETCH_SIZE" / "pmc_counter_collection.csv"), pmc(src / "pmc_WRITE_SIZE" / "pmc_counter_collection.csv")
  sq: dict = {}
  for i in (1, 2):
    for k, d in pmc(src / f"pmc_SQ{i}" / "pmc_counter_collection.csv").items():
      sq.setdefault(k, {}).update(d)
  cal: dict = {"source": f"gpurun_out/{tag} (tools/ubench.sh)", "stream": {}, "valu": {}}
  print("== streaming kernels: known bytes vs FETCH_SIZE / WRITE_SIZE (KiB as reported) ==")
  for r in rows:
    if "read_bytes" not in r:
      continue
    k = r["kernel"]
    f_kib, w_kib = fetch.get(k, {}).get("FETCH_SIZE"), write.get(k, {}).get("WRITE_SIZE")
    ff = r["read_bytes"] / (f_kib * 1024) if f_kib and r["read_bytes"] else None
    wf = r["write_bytes"] / (w_kib * 1024) if w_kib and r["write_bytes"] else None
    cal["stream"][k] = {"read_bytes": r["read_bytes"], "write_bytes": r["write_bytes"], "GBps": r["GBps"], "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
                        "fetch_factor": ff, "write_factor": wf}
    print(f"{k:18s} {r['GBps']:8.1f} GB/s  read {r['read_bytes']/1e6:8.1f} MB  FETCH_SIZE {f_kib}  -> factor {ff}   write {r['write_bytes']/1e6:8.1f} MB  WRITE_SIZE {w_kib} -> factor {wf}")
  print("== VALU kernels: cycles per wave64 instruction ==")
  for r in rows:
    if "waves_per_simd" not in r:
      continue
    k, wps = r["kernel"], r["waves_per_simd"]
    line = f"{k:20s} wps {wps}  per wave {r['cycles_per_inst_wave']:6.2f} cyc/inst   per SIMD {r['cycles_per_inst_simd']:6.3f} cyc/inst   {r['inst_per_us_per_simd']:8.1f} inst/us/SIMD"
    ent = cal["valu"].setdefault(k, {})
    ent[f"wps{wps}"] = {"cycles_per_inst_wave": r["cycles_per_inst_wave"], "cycles_per_inst_simd": r["cycles_per_inst_simd"], "inst_per_us_per_simd": r["inst_per_us_per_simd"]}
    if wps == 4 and k in sq:
      d = sq[k]
      insts, busy = d.get("SQ_INSTS_VALU"), d.get("SQ_BUSY_CYCLES")
      if insts and busy:
        simd_cycles = busy / 32.0 * 1024.0  # SQ_BUSY_CYCLES is summed over the 32 shader engines; 1024 SIMDs
        ent["sq"] = {c: d.get(c) for c in ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA")}
        ent["sq_simd_cycles_per_valu_inst"] = simd_cycles / insts
        line += f"   SQ: INSTS_VALU {insts:.3g} BUSY_CYCLES {busy:.3g} -> {simd_cycles / insts:.3f} SIMD-cycles per VALU inst; ACTIVE_INST_VALU {d.get('SQ_ACTIVE_INST_VALU')}"
    print(line)
  if commit:
    (ROOT / "profiles" / "calibration.json").write_text(json.dumps(cal, indent=1) + "\n")


if __name__ == "__main__":
  main(sys.argv[1], "--commit" in sys.argv)
