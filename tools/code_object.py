"""The gfx950 code objects inside a built library: extract them from the HIP fat binary (one clang offload bundle per
translation unit), read the kernels' register / scratch / LDS footprint from the AMDGPU metadata notes and count
instruction classes in the disassembly.  Runs anywhere the ROCm LLVM tools are installed (no GPU).

  python tools/code_object.py [library.so] [kernel-name-substring]
"""
from __future__ import annotations

import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_objects(lib: Path) -> list[bytes]:
  """The amdgcn ELF images of every bundle in the library's .hip_fatbin section."""
  with tempfile.TemporaryDirectory() as td:
    fat = Path(td) / "fatbin"
    subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(lib)], check=True, capture_output=True)
    blob = fat.read_bytes()
  out = []
  pos = blob.find(MAGIC)
  while pos >= 0:
    (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
    p = pos + len(MAGIC) + 8
    for _ in range(n):
      off, size, tlen = struct.unpack_from("<QQQ", blob, p)
      triple = blob[p + 24 : p + 24 + tlen].decode()
      p += 24 + tlen
      if "amdgcn" in triple and size:
        out.append(blob[pos + off : pos + off + size])
    pos = blob.find(MAGIC, pos + 1)
  return out


# Issue cost (SIMD cycles per wave64 instruction) and issued fp32 operations per lane of the VALU instruction classes.  Cycles:
# measured with tools/ubench.hip on one MI355X at 4 waves per SIMD (profiles/r03_ubench/report.txt, profiles/README.md
# "calibration"): v_fma_f32 2 (the guide's figure, MI355X_MICROARCH.md:52-53), v_pk_fma_f32 4 (1.87 x the v_fma rate: NO
# throughput gain per flop over v_fma_f32), DPP-modified VALU 4 (1.62 x), v_readlane_b32 -> SGPR -> v_fma pair 7 (readlane 5),
# v_mfma_f32_16x16x4_f32 32 (29-34 measured).  Transcendentals (quarter rate, 8) are the guide's figure, not measured here.
def valu_class(op: str, line: str) -> tuple[str, float, float]:
  """-> (class, cycles, flops per lane) of one disassembled VALU instruction."""
  if "mfma" in op:
    return "mfma", 32.0, 2048.0 / 64.0
  dpp = "dpp" in line
  if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
    return "readlane", 5.0, 0.0
  if op.startswith("v_pk_fma_f32"):
    return "pk_fma", 4.0, 4.0
  if op.startswith(("v_pk_mul_f32", "v_pk_add_f32")):
    return "pk_other", 4.0, 2.0
  if op.startswith(("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")):
    return "trans", 8.0, 1.0
  if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mad_f32", "v_mac_f32")):
    return ("dpp" if dpp else "fma"), (4.0 if dpp else 2.0), 2.0
  if op.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_max_f32", "v_min_f32", "v_med3_f32", "v_mul_legacy_f32")):
    return ("dpp" if dpp else "f32_1op"), (4.0 if dpp else 2.0), 1.0
  if op.endswith(("_b64", "_u64", "_i64", "_f64")) or "u64" in op:
    return "wide", 4.0, 0.0
  return ("dpp" if dpp else "other"), (4.0 if dpp else 2.0), 0.0


def kernels(lib: Path) -> dict[str, dict]:
  """kernel name (demangled-ish: the mangled symbol) -> metadata fields + instruction counts."""
  res: dict[str, dict] = {}
  with tempfile.TemporaryDirectory() as td:
    for k, img in enumerate(device_objects(lib)):
      f = Path(td) / f"dev{k}.co"
      f.write_bytes(img)
      notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(f)], capture_output=True, text=True).stdout
      for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
          continue
        md = {key: int(v) for key, v in re.findall(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)", blk)}
        res[name.group(1)] = md
      dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(f)], capture_output=True, text=True).stdout
      cur = None
      for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
          cur = m.group(1) if m.group(1) in res else None
          continue
        if cur is None:
          continue
        ins = line.split()
        if len(ins) < 1:
          continue
        op = ins[0]
        c = res[cur].setdefault("insts", {})
        for key, pat in (("mfma", "v_mfma"), ("dpp", "_dpp"), ("pk_fma", "v_pk_fma_f32"), ("setprio", "s_setprio"), ("scratch", "scratch_"), ("readlane", "v_readlane")):
          if pat in op or (key == "dpp" and "dpp" in line):
            c[key] = c.get(key, 0) + 1
        c["total"] = c.get("total", 0) + 1
        if op.startswith("v_"):
          cls, cyc, fl = valu_class(op, line)
          v = res[cur].setdefault("valu", {})
          e = v.setdefault(cls, [0, 0.0, 0.0])  # count, cycles, lane-flops
          e[0] += 1; e[1] += cyc; e[2] += fl
  return res


def valu_mix(md: dict) -> dict:
  """Static mix of a kernel's VALU instructions: average issue cycles and issued fp32 operations (x 64 lanes) per VALU
  instruction (MFMA kept apart: SQ_INSTS_MFMA counts it separately)."""
  v = {k: e for k, e in md.get("valu", {}).items() if k != "mfma"}
  n = sum(e[0] for e in v.values())
  return {"valu_insts_static": n, "cycles_per_valu_inst": sum(e[1] for e in v.values()) / max(n, 1),
          "flops_per_valu_inst_wave": 64.0 * sum(e[2] for e in v.values()) / max(n, 1),
          "classes": {k: e[0] for k, e in sorted(md.get("valu", {}).items())}}


if __name__ == "__main__":
  lib = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parents[1] / "mjlab_amd" / "csrc" / "libmjlab_amd.so"
  sub = sys.argv[2] if len(sys.argv) > 2 else ""
  for name, md in sorted(kernels(lib).items()):
    if sub in name:
      print(name, {k: v for k, v in md.items() if k not in ("insts", "valu")}, md.get("insts", {}), valu_mix(md))
