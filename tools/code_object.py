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
  return res


if __name__ == "__main__":
  lib = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parents[1] / "mjlab_amd" / "csrc" / "libmjlab_amd.so"
  sub = sys.argv[2] if len(sys.argv) > 2 else ""
  for name, md in sorted(kernels(lib).items()):
    if sub in name:
      print(name, {k: v for k, v in md.items() if k != "insts"}, md.get("insts", {}))
