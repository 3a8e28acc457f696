"""Static check of built gfx950 code objects for the miscompile behind round 5's "memory aperture violation" (DESIGN.md section 7).

hipcc (ROCm 7.2's LLVM) can place a register-allocator spill STORE at the top of a join / loop-exit block AHEAD of the
`s_or_b64 exec, exec, sN` that gives the lanes back.  Arriving from a divergent loop's exit (or over the `s_cbranch_execz` around it) EXEC
is zero there: the store writes nothing and the reload returns stale scratch memory -- zeros in a fresh process, another kernel's spills
later.  The pattern is sporadic (of the 27 fused cone kernels at four waves per SIMD two sizes carry it, at three waves one), so the build
checks every cone translation unit and falls back to fewer waves per SIMD where it appears (mjlab_amd/native.py), and the CPU suite
asserts that no kernel of the finished library carries it (tests/test_code_object.py).

The analysis: a forward may-analysis of "EXEC may be zero" over each kernel's control-flow graph (seeds: the fall-through of
`s_cbranch_execnz`, the taken edge of `s_cbranch_execz`; cleared by any write of EXEC); every scratch / global / flat / buffer / LDS
instruction reached in that state is recorded.  The `s_cbranch_execz` the compiler puts around every divergent region makes that a wide
net, so the FATAL class is the precise shape: a scratch STORE on such a path that is followed, inside its basic block and before any
other write of EXEC, by `s_or_b64 exec, exec, sN`.  CLI: tools/exec_zero_check.py.
"""
from __future__ import annotations

import re
import struct
import subprocess
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


MEM = ("scratch_", "global_", "flat_", "buffer_", "ds_")


def disassemble(img: bytes) -> str:
  with tempfile.TemporaryDirectory() as td:
    f = Path(td) / "dev.co"
    f.write_bytes(img)
    return subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(f)], capture_output=True, text=True).stdout


def functions(dis: str) -> dict[str, list[tuple[int, str]]]:
  """name -> [(address, instruction text)] in address order."""
  out: dict[str, list[tuple[int, str]]] = {}
  cur = None
  for line in dis.splitlines():
    m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
    if m:
      cur = out.setdefault(m.group(2), [])
      continue
    if cur is None:
      continue
    m = re.match(r"^\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
    if m and m.group(1):
      cur.append((int(m.group(2), 16), m.group(1).strip()))
  return out


def writes_exec(ins: str) -> bool:
  op, _, rest = ins.partition(" ")
  if "saveexec" in op:
    return True
  dst = rest.split(",")[0].strip()
  return dst in ("exec", "exec_lo", "exec_hi") and op.startswith("s_")


def analyse(insts: list[tuple[int, str]]) -> list[tuple[int, str, str]]:
  """-> [(address, instruction, how EXEC got to zero)] for memory instructions that may execute with EXEC == 0."""
  index = {a: i for i, (a, _) in enumerate(insts)}
  n = len(insts)
  zero_in: list[str | None] = [None] * n  # reason string when EXEC may be zero on entry
  work: list[int] = []

  def push(i: int, why: str):
    if 0 <= i < n and zero_in[i] is None:
      zero_in[i] = why
      work.append(i)

  def target(a: int, ins: str) -> int | None:
    m = re.match(r"s_c?branch\S*\s+(\d+)", ins)
    if not m:
      return None
    simm = int(m.group(1))
    if simm >= 0x8000:
      simm -= 0x10000
    return index.get(a + 4 + 4 * simm)

  for i, (a, ins) in enumerate(insts):
    if ins.startswith("s_cbranch_execnz"):
      push(i + 1, f"fall-through of s_cbranch_execnz at {a:#x} (divergent loop exit)")
    elif ins.startswith("s_cbranch_execz"):
      t = target(a, ins)
      if t is not None:
        push(t, f"taken s_cbranch_execz at {a:#x}")
  # basic-block leaders: branch targets and fall-throughs of branches
  leader = set()
  for i, (a, ins) in enumerate(insts):
    if ins.startswith(("s_branch", "s_cbranch")):
      t = target(a, ins)
      if t is not None:
        leader.add(t)
      leader.add(i + 1)

  def before_restore(i: int) -> bool:
    """instruction i is followed, inside its basic block and before any other write of EXEC, by `s_or_b64 exec, exec, sN`: it sits in
    the prologue of a join / loop-exit block, ahead of the instruction that gives the lanes back."""
    for j in range(i + 1, n):
      if j in leader:
        return False
      ins = insts[j][1]
      if ins.startswith("s_or_b64 exec, exec,"):
        return True
      if writes_exec(ins) or ins.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
        return False
    return False

  found = []
  seen_report = set()
  while work:
    i = work.pop()
    why = zero_in[i]
    a, ins = insts[i]
    if ins.startswith(MEM) and i not in seen_report:
      seen_report.add(i)
      found.append((a, ins, why + ("; AHEAD OF THE EXEC RESTORE of its block" if before_restore(i) else "")))
    if writes_exec(ins):
      continue  # EXEC rewritten: state unknown (not zero for our purpose)
    if ins.startswith("s_endpgm"):
      continue
    if ins.startswith("s_branch"):
      t = target(a, ins)
      if t is not None:
        push(t, why)
      continue
    if ins.startswith("s_cbranch_execnz"):  # EXEC == 0 here: not taken
      push(i + 1, why)
      continue
    if ins.startswith("s_cbranch_execz"):  # EXEC == 0 here: taken
      t = target(a, ins)
      if t is not None:
        push(t, why)
      continue
    if ins.startswith("s_cbranch"):  # scc / vcc branches: both ways
      t = target(a, ins)
      if t is not None:
        push(t, why)
    push(i + 1, why)
  return sorted(found)


def fatal(hits):
  """The hits that are the miscompile: scratch STORES on a path where EXEC may be zero, ahead of their block's `s_or_b64 exec, exec, sN`."""
  return [h for h in hits if h[1].startswith("scratch_store") and "AHEAD OF THE EXEC RESTORE" in h[2]]


def check(path: Path, sub: str = "") -> dict[str, list[tuple[int, str, str]]]:
  data = path.read_bytes()
  imgs = [data] if path.suffix == ".co" else device_objects(path)
  from concurrent.futures import ThreadPoolExecutor

  res = {}
  with ThreadPoolExecutor(max_workers=8) as pool:  # (llvm-objdump per code object: subprocesses, the threads only wait)
    for dis in pool.map(disassemble, imgs):
      for name, insts in functions(dis).items():
        if sub in name and insts:
          res[name] = analyse(insts)
  return res




def fatal_hits(path: Path, sub: str = "") -> dict[str, list[tuple[int, str, str]]]:
  """kernel name -> the miscompiled spill stores of that kernel in `path` (a library, an object or a code object); {} = clean."""
  return {name: fatal(hits) for name, hits in check(Path(path), sub).items() if fatal(hits)}
