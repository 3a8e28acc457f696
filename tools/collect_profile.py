"""Condense one tools/gpu_round.sh output directory (gpurun_out/<tag>) into profiles/<tag>/:

  kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (kernel names truncated)
  hbm_traffic.csv    per-kernel FETCH_SIZE / WRITE_SIZE per launch from the --pmc passes
  bench.json         the bench.py line of the same build
  phases.txt         shader-clock phase breakdown of the solve kernel (profiling build)
and refreshes profiles/traffic.json (read by bench.py for roofline.traffic).

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE
and WRITE_SIZE are collected in separate --pmc passes and are reported in KiB.  The factors
applied to them are the ones MEASURED on known byte counts with tools/ubench.hip through the
same passes (profiles/calibration.json, profiles/README.md "calibration"): FETCH_SIZE x 2.0
for 4 B/lane AND 16 B/lane reads alike (the guide states it for 16 B/lane only), WRITE_SIZE
x 1.0.  VALU occupancy: SQ_INSTS_VALU weighted by the issue cycles of the kernel's static
instruction mix (tools/code_object.py valu_class: v_fma 2, v_pk_fma 4, DPP 4, readlane 5 --
measured), with the all-2-cycle figure beside it as the lower bound.
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

CAL = ROOT / "profiles" / "calibration.json"
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 64 lanes x 2 flop / 2 cycles x 2.4 GHz


def factors() -> tuple[float, float]:
  """(FETCH_SIZE factor, WRITE_SIZE factor) measured by tools/ubench.sh; the 4 B/lane streaming kernels are this library's pattern."""
  if CAL.exists():
    st = json.loads(CAL.read_text()).get("stream", {})
    f, w = st.get("k_ub_read4", {}).get("fetch_factor"), st.get("k_ub_write4", {}).get("write_factor")
    if f and w:
      return float(f), float(w)
  return 2.0, 1.0


def static_mix(kernel_short: str) -> dict | None:
  """Static VALU mix of the built kernel whose demangled name starts like `kernel_short` (k_control_step<36> ...)."""
  import re

  from code_object import kernels, valu_mix

  m = re.match(r"(k_\w+?)(?:<(\d+)(?:, (true|false))?>)?$", kernel_short)
  if not m:
    return None
  base, nvp, flag = m.groups()
  lib = ROOT / "mjlab_amd" / "csrc" / "libmjlab_amd.so"
  if not lib.exists():
    return None
  global _KCACHE
  try:
    _KCACHE
  except NameError:
    _KCACHE = kernels(lib)
  for name, md in _KCACHE.items():
    if f"{len(base)}{base}" not in name:
      continue
    if nvp and f"ILi{nvp}E" not in name:
      continue
    if flag and ("Lb1" if flag == "true" else "Lb0") not in name:
      continue
    return valu_mix(md)
  return None


def short(name: str) -> str:
  return name.split("(")[0].replace("void ", "").strip()


def pmc_mean(path: Path) -> dict:
  acc, n = collections.defaultdict(list), collections.defaultdict(int)
  for r in csv.DictReader(open(path)):
    k = short(r["Kernel_Name"])
    acc[k].append(float(r["Counter_Value"]))
    n[k] += int(float(r.get("Launches") or 1))  # files condensed by tools/reduce_pmc.py carry the mean and the launch count
  return {k: (sum(v) / len(v), n[k]) for k, v in acc.items()}


def main(tag: str, scene: str = "g1_velocity_flat") -> None:
  src, dst = ROOT / "gpurun_out" / tag, ROOT / "profiles" / tag
  dst.mkdir(parents=True, exist_ok=True)
  summarize(src / "prof" / "trace_kernel_stats.csv", dst / "kernel_stats.csv")
  fetch = pmc_mean(src / "pmc_FETCH_SIZE" / "pmc_counter_collection.csv")
  write = pmc_mean(src / "pmc_WRITE_SIZE" / "pmc_counter_collection.csv")
  rows = []
  ffac, wfac = factors()
  for k in sorted(fetch):
    if not k.startswith("k_"):
      continue
    f_kib, n = fetch[k]
    w_kib = write.get(k, (0.0, 0))[0]
    rows.append((k, n, f_kib, ffac * f_kib * 1024, wfac * w_kib * 1024))
  with open(dst / "hbm_traffic.csv", "w") as f:
    f.write(f"kernel,launches,FETCH_SIZE_KiB_raw,fetch_bytes_per_launch_x{ffac:.3f},write_bytes_per_launch_x{wfac:.3f}\n")
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
  acc = collections.defaultdict(lambda: collections.defaultdict(list))
  for sqdir in ("pmc_SQ", "pmc_SQ1", "pmc_SQ2"):
    sq = src / sqdir / "pmc_counter_collection.csv"
    if sq.exists():
      for r in csv.DictReader(open(sq)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
  if acc:
    stats = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(src / "prof" / "trace_kernel_stats.csv"))} if (src / "prof" / "trace_kernel_stats.csv").exists() else {}
    lines = []
    for k, d in sorted(acc.items()):
      if not k.startswith("k_"):
        continue
      mean = {c: sum(v) / len(v) for c, v in d.items()}
      lines.append(k + "\n" + "\n".join(f"   {c:28s} {v:16.0f}" for c, v in sorted(mean.items())))
      if "SQ_INSTS_VALU" in mean and mean.get("SQ_BUSY_CYCLES"):
        simd_cycles = mean["SQ_BUSY_CYCLES"] / 32.0 * 1024.0  # SQ_BUSY_CYCLES is summed over the 32 shader engines; 1024 SIMDs
        mix = static_mix(k) or {}
        cpi = mix.get("cycles_per_valu_inst", 2.0)
        mfma = mean.get("SQ_INSTS_MFMA", 0.0)  # counted inside SQ_INSTS_VALU as well (tools/ubench.hip k_ub_mfma)
        valu = mean["SQ_INSTS_VALU"] - mfma
        busy_lo = (2.0 * valu + 32.0 * mfma) / simd_cycles
        busy = (cpi * valu + 32.0 * mfma) / simd_cycles
        flops = valu * mix.get("flops_per_valu_inst_wave", 0.0) + mfma * 2048.0
        lines.append(f"   {'valu_busy, all VALU at 2 cycles (lower bound)':60s} {busy_lo:8.3f}")
        lines.append(f"   {'valu_busy, static mix %.2f cycles per VALU inst' % cpi:60s} {busy:8.3f}")
        lines.append(f"   {'issued fp32 operations per launch (static mix x SQ_INSTS_VALU + 2048 x MFMA)':60s} {flops:16.0f}")
        ns = next((v for kk, v in stats.items() if kk == k), None)
        if ns:
          lines.append(f"   {'-> TFLOP/s at %.1f us per launch (peak %.1f)' % (ns / 1e3, PEAK_FP32_TFLOPS):60s} {flops / ns / 1e3:8.2f}")
        key = "solve_integrate" if k.startswith("k_solve") else ("substep" if k.startswith("k_control_step") or (k.startswith("k_substep") and "true" in k and "substep_valu_busy" not in ent) else None)
        if key:
          ent[key + "_valu_busy"] = busy
          ent[key + "_valu_busy_2cycle_lower_bound"] = busy_lo
          ent[key + "_flops_per_launch"] = flops
          ent[key + "_valu_cycles_per_inst_static_mix"] = cpi
    (dst / "sq_counters.txt").write_text("\n".join(lines) + "\n")
  traffic[scene] = ent
  tj.write_text(json.dumps(traffic, indent=1) + "\n")
  print(open(dst / "hbm_traffic.csv").read())
  print(open(dst / "kernel_stats.csv").read()[:1200])


if __name__ == "__main__":
  main(*sys.argv[1:])
