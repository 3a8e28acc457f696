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

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd.code_check import analyse, check, disassemble, fatal, functions, writes_exec  # noqa: E402,F401  (the analysis lives in the package: the build uses it)

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
