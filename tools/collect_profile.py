"""Condense one tools/gpu_round.sh output directory (gpurun_out/<tag>) into profiles/<tag>/:

  kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (kernel names truncated)
  hbm_traffic.csv    per-kernel FETCH_SIZE / WRITE_SIZE per launch from the --pmc passes
  bench.json         the bench.py line of the same build
  phases.txt         shader-clock phase breakdown of the solve kernel (profiling build)
and refreshes profiles/traffic.json (read by bench.py for roofline.traffic).

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE
and WRITE_SIZE are collected in separate --pmc passes, are reported in KiB, and on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is left as reported
(uncalibrated per the guide).
"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
from summarize_profile import main as summarize  # noqa: E402


def short(name: str) -> str:
  return name.split("(")[0].replace("void ", "").strip()


def pmc_mean(path: Path) -> dict:
  acc = collections.defaultdict(list)
  for r in csv.DictReader(open(path)):
    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
  return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main(tag: str, scene: str = "g1_velocity_flat") -> None:
  src, dst = ROOT / "gpurun_out" / tag, ROOT / "profiles" / tag
  dst.mkdir(parents=True, exist_ok=True)
  summarize(src / "prof" / "trace_kernel_stats.csv", dst / "kernel_stats.csv")
  fetch = pmc_mean(src / "pmc_FETCH_SIZE" / "pmc_counter_collection.csv")
  write = pmc_mean(src / "pmc_WRITE_SIZE" / "pmc_counter_collection.csv")
  rows = []
  for k in sorted(fetch):
    if not k.startswith("k_"):
      continue
    f_kib, n = fetch[k]
    w_kib = write.get(k, (0.0, 0))[0]
    rows.append((k, n, f_kib, 2.0 * f_kib * 1024, w_kib * 1024))
  with open(dst / "hbm_traffic.csv", "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KiB_raw,fetch_bytes_per_launch_corrected_x2,write_bytes_per_launch\n")
    for r in rows:
      f.write(f"{r[0]},{r[1]},{r[2]:.1f},{r[3]:.0f},{r[4]:.0f}\n")
  for name, out in (("bench.json", "bench.json"), ("phases.log", "phases.txt")):
    if (src / name).exists():
      shutil.copy(src / name, dst / out)
  tj = ROOT / "profiles" / "traffic.json"
  traffic = json.loads(tj.read_text()) if tj.exists() else {}
  ent = {"source": f"profiles/{tag}/hbm_traffic.csv"}
  for r in rows:
    ent[r[0].replace("<", "_").replace(">", "").replace(", ", "_") + "_bytes_per_launch"] = r[3] + r[4]
  solve = [r for r in rows if r[0].startswith("k_solve")]
  if solve:
    ent["solve_integrate_bytes_per_launch"] = solve[0][3] + solve[0][4]
  sub = [r for r in rows if r[0].startswith("k_control_step")] or [r for r in rows if r[0].startswith("k_substep") and "true" in r[0]]  # forward() is <.., false>
  if sub:
    ent["substep_bytes_per_launch"] = sub[0][3] + sub[0][4]
  ent["all_stage_kernels_bytes_per_step"] = sum(r[3] + r[4] for r in rows)
  # VALU issue share of the dominant kernel (bench.py roofline.valu_busy): a wave64 VALU instruction occupies
  # its SIMD16 for 4 cycles; 256 CUs x 4 SIMDs; SQ_BUSY_CYCLES is summed over the 32 shader engines
  sq = src / "pmc_SQ" / "pmc_counter_collection.csv"
  if sq.exists():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sq)):
      acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = []
    for k, d in sorted(acc.items()):
      if not k.startswith("k_"):
        continue
      mean = {c: sum(v) / len(v) for c, v in d.items()}
      lines.append(k + "\n" + "\n".join(f"   {c:28s} {v:16.0f}" for c, v in sorted(mean.items())))
      if "SQ_INSTS_VALU" in mean and mean.get("SQ_BUSY_CYCLES"):
        busy = 4.0 * mean["SQ_INSTS_VALU"] / (mean["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
        lines.append(f"   {'valu_busy (4 x INSTS_VALU / (BUSY_CYCLES / 32 x 1024 SIMDs))':28s} {busy:16.3f}")
        if k.startswith("k_solve"):
          ent["solve_integrate_valu_busy"] = busy
        if k.startswith("k_control_step") or (k.startswith("k_substep") and "true" in k and "substep_valu_busy" not in ent):
          ent["substep_valu_busy"] = busy
    (dst / "sq_counters.txt").write_text("\n".join(lines) + "\n")
  traffic[scene] = ent
  tj.write_text(json.dumps(traffic, indent=1) + "\n")
  print(open(dst / "hbm_traffic.csv").read())
  print(open(dst / "kernel_stats.csv").read()[:1200])


if __name__ == "__main__":
  main(*sys.argv[1:])
