"""Static check of the gfx950 code objects for memory instructions that execute with EXEC == 0.

Why this exists (round 6): the fused elliptic-cone kernels built at four waves per SIMD faulted ("memory aperture
violation") when a 32-dof model ran after a 36-dof one.  rocgdb on the faulting wave showed a 64-bit index reloaded from a spill slot
that held another kernel's leftovers; the disassembly shows why -- hipcc (ROCm 7.2's LLVM) placed the spill STORE of that value in the
exit block of a divergent loop, in front of the `s_or_b64 exec, exec, sN` that restores the lanes:

    loop:  ...
           s_andn2_b64 exec, exec, s[4:5]        ; lanes leave the loop as they finish
           s_cbranch_execnz loop
           s_mov_b64 s[64:65], 0x100
           scratch_store_dwordx2 off, v[24:25], off offset:280     ; EXEC == 0: nothing is written
           s_or_b64 exec, exec, s[2:3]
    ...    scratch_load_dwordx2 v[4:5], off, off offset:280        ; whatever the slot held before this kernel started

A store under an empty EXEC mask is a no-op, so the reload returns stale scratch memory: zeros in a fresh process (the tests "passed
alone"), another kernel's spills after a kernel with a different frame ran.  This tool finds that pattern without a GPU: a forward
may-analysis of "EXEC is known to be zero" over each kernel's control-flow graph (the fall-through of `s_cbranch_execnz`, the taken
edge of `s_cbranch_execz`), cleared by any write to EXEC; every scratch / global / flat / buffer / LDS instruction reached in that
state is recorded.  The `s_cbranch_execz` the compiler puts around every divergent region makes that a wide net (at kernel entry EXEC is
never zero, yet the path exists), so the FATAL class is the precise shape of the miscompile: a scratch STORE on such a path that is
followed, inside its basic block and before any other write of EXEC, by `s_or_b64 exec, exec, sN` -- spill code in the prologue of a
join / loop-exit block, ahead of the instruction that gives the lanes back.  tests/test_code_object.py asserts there is none in the
shipped library.

  python tools/exec_zero_check.py [library.so | object.o | code object] [kernel-name-substring]
"""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import code_object  # noqa: E402

LLVM = code_object.LLVM
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
  imgs = [data] if path.suffix == ".co" else code_object.device_objects(path)
  from concurrent.futures import ThreadPoolExecutor

  res = {}
  with ThreadPoolExecutor(max_workers=8) as pool:  # (llvm-objdump per code object: subprocesses, the threads only wait)
    for dis in pool.map(disassemble, imgs):
      for name, insts in functions(dis).items():
        if sub in name and insts:
          res[name] = analyse(insts)
  return res


if __name__ == "__main__":
  lib = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parents[1] / "mjlab_amd" / "csrc" / "libmjlab_amd.so"
  sub = sys.argv[2] if len(sys.argv) > 2 else ""
  bad = 0
  for name, hits in sorted(check(lib, sub).items()):
    stores = fatal(hits)
    maybe = [h for h in hits if h[1].startswith("scratch_") and h not in stores]
    print(f"{name}: spill stores ahead of their block's EXEC restore: {len(stores)}; other scratch instructions on a may-be-zero path: {len(maybe)}; "
          f"other memory instructions: {len(hits) - len(stores) - len(maybe)}")
    for a, ins, why in stores:
      print(f"    {a:#x}: {ins}    <- {why}")
    if "-v" in sys.argv:
      for a, ins, why in maybe:
        print(f"      ({a:#x}: {ins}    <- {why})")
    bad += len(stores)
  print("FATAL (spill stores that write nothing):", bad)
  sys.exit(1 if bad else 0)
