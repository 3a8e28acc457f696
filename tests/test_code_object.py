"""What the built library contains for gfx950 (no GPU needed: the ROCm LLVM tools read the code objects).  Guards the
properties the design rests on: every kernel fits 4 waves per SIMD (<= 128 VGPRs, launch bound 64 x 4), uses dynamic LDS
only, the solver kernels really carry the fp32 MFMA Hessian, DPP reductions, packed fp32 FMAs and the wave-priority
instruction, and scratch (spill) space stays small."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))

from mjlab_amd import native  # noqa: E402


@pytest.fixture(scope="module")
def kernels():
  import code_object

  if not (code_object.LLVM / "llvm-objdump").exists():
    pytest.skip("ROCm LLVM tools not installed")
  native.lib()  # builds the library if it is missing
  return code_object.kernels(ROOT / "mjlab_amd" / "csrc" / "libmjlab_amd.so")


def test_every_kernel_fits_four_waves_per_simd(kernels):
  assert len(kernels) >= 40  # 9 padded sizes x (solve, forward, step, control step) + the size-independent kernels
  for name, md in kernels.items():
    if "k_poison_scratch" in name:  # the diagnostic that dirties scratch on purpose (extras.h): its frame must exceed every other kernel's
      assert md["private_segment_fixed_size"] >= 1280, (name, md)
      continue
    if "_cone" in name and "k_constraint_cone" not in name:  # elliptic cones (stage_cone.h, kernels.h): kernels of their own off the measured path
      # The fused ones are built for four waves per SIMD where hipcc compiles that budget correctly, else for three, else for the spill-free
      # two (native.py picks per translation unit with mjlab_amd/code_check.py; the round-5 fault was ONE miscompiled spill store, not
      # spilling as such: DESIGN.md section 7; test_no_spill_store_executes_ahead_of_its_exec_restore below); k_solve_cone: two, spill-free.
      assert md["group_segment_fixed_size"] == 0 and md["private_segment_fixed_size"] <= 1024, (name, md)
      assert md["vgpr_spill_count"] == 0 or md["vgpr_count"] <= 168, (name, md)
      continue
    assert md["vgpr_count"] <= 128, (name, md)
    if "k_command_motion_sample" in name:  # k_command_motion_sample / _sampler: environment terms, ONE workgroup per launch -- the histogram /
      assert md["group_segment_fixed_size"] <= 4 * 4096 + 2 * 4 * 256 + 16, (name, md)  # distribution (MJLAB_MOTION_SAMPLE_MAX_BINS floats) and two reduction arrays are static LDS
      continue
    if "k_masked_sums" in name:  # an environment term: four waves per summed vector, their partial sums meet in 16 bytes of static LDS
      assert md["group_segment_fixed_size"] <= 16 and md["vgpr_spill_count"] == 0, (name, md)
      continue
    if "k_chol_selftest" in name:  # the factorization's diagnostic (nvp_inst.hip, part 2): its factor block is static LDS
      assert md["group_segment_fixed_size"] <= 4 * 64 * 69 and md["vgpr_spill_count"] == 0, (name, md)
      continue
    assert md["group_segment_fixed_size"] == 0, (name, md)  # LDS is laid out per model at launch (mjlab_lds_bytes)
    # scratch per lane: the 64-dof instantiations (beyond every model of the reference) spill the most
    assert md["private_segment_fixed_size"] <= (1024 if "ILi64E" in name else 512), (name, md)


def test_g1_kernels_are_cdna4_code(kernels):
  selftest = [md for n, md in kernels.items() if "k_chol_selftestILi36E" in n]
  assert len(selftest) == 1 and selftest[0]["insts"].get("mfma", 0) == 51  # the diagnostic runs the tile factorization itself: 51 MFMAs per factor site
  g1 = {n: md for n, md in kernels.items() if "ILi36E" in n and "_cone" not in n and "k_chol_selftest" not in n}  # (the elliptic-cone kernels have their own allocation and limits: above)
  names = " ".join(g1)
  assert all(k in names for k in ("k_solve_integrate", "k_substep", "k_control_step"))
  for name, md in g1.items():
    ins = md["insts"]
    # J^T D J (6 upper 16 x 16 tiles x 4 row groups per block) and, since round 5, the LDL^T of the Hessian taken apart in those
    # tiles (51 MFMAs per factor site: common.h, chol_factor_tiles)
    assert ins.get("mfma", 0) >= 24 + 2 * 51, (name, ins)
    assert ins.get("dpp", 0) >= 200, (name, ins)      # wave / row reductions without LDS
    assert ins.get("pk_fma", 0) >= 100, (name, ins)   # packed fp32 multiply-adds (the LDS-broadcast sweep of round 2-4 had 300 more: PGS keeps it)
    assert ins.get("setprio", 0) >= 4, (name, ins)    # wave issue priority by the world's constraint rows
    assert md["private_segment_fixed_size"] <= 320, (name, md)  # (k_substep<36, false>, the forward() kernel: 288 since the literal-cost switch)
  ctrl = next(md for n, md in g1.items() if "k_control_step" in n)
  assert ctrl["vgpr_count"] == 128 and ctrl["vgpr_spill_count"] <= 24, ctrl  # (16 since the tiles are factored where they lie; 39-41 before)


def test_exec_zero_checker_recognises_the_round5_miscompile():
  """tools/exec_zero_check.py on the instruction sequence of the faulting build (k_substep_cone<32, true> at four waves per SIMD,
  profiles/r06_fault): the spill store in the loop's exit block, ahead of the EXEC restore, is the one FATAL hit; the same store
  after the restore is not."""
  import exec_zero_check as z

  def prog(store_first):
    body = [
      "s_mov_b64 s[2:3], exec", "s_and_b64 s[0:1], s[2:3], s[0:1]", "s_mov_b64 exec, s[0:1]", "s_cbranch_execz 7",
      "global_load_dword v7, v[4:5], off", "v_add_u32_e32 v6, 64, v6", "v_cmp_le_i32_e32 vcc, s6, v6", "s_or_b64 s[4:5], vcc, s[4:5]",
      "ds_write_b32 v9, v7", "s_andn2_b64 exec, exec, s[4:5]", "s_cbranch_execnz 65529",
    ]
    tail = ["s_mov_b64 s[64:65], 0x100"]
    tail += ["scratch_store_dwordx2 off, v[24:25], off offset:280", "s_or_b64 exec, exec, s[2:3]"] if store_first else \
            ["s_or_b64 exec, exec, s[2:3]", "scratch_store_dwordx2 off, v[24:25], off offset:280"]
    tail += ["s_load_dwordx2 s[0:1], s[68:69], 0x80", "s_endpgm"]
    return [(0x1000 + 4 * i, ins) for i, ins in enumerate(body + tail)]

  bad = z.fatal(z.analyse(prog(True)))
  assert len(bad) == 1 and "offset:280" in bad[0][1], bad
  assert z.fatal(z.analyse(prog(False))) == []


def test_no_spill_store_executes_ahead_of_its_exec_restore():
  """The defect behind round 5's "memory aperture violation" (DESIGN.md section 7): a VGPR spill store that the compiler placed in a join /
  loop-exit block in front of `s_or_b64 exec, exec, sN` executes with EXEC == 0 and writes nothing; the reload then returns stale scratch.
  No kernel of the shipped library may contain that pattern (the dynamic counterpart is tests/test_gpu_scratch.py)."""
  import code_object
  import exec_zero_check as z

  if not (code_object.LLVM / "llvm-objdump").exists():
    pytest.skip("ROCm LLVM tools not installed")
  native.lib()
  res = z.check(ROOT / "mjlab_amd" / "csrc" / "libmjlab_amd.so")
  assert len(res) >= 40
  bad = {name: z.fatal(hits) for name, hits in res.items() if z.fatal(hits)}
  assert not bad, bad
  # what the build chose for the cone units (written by native.build): the G1's size runs at four waves per SIMD
  import json

  choice = json.loads((ROOT / "mjlab_amd" / "csrc" / "libmjlab_amd.cone_waves_per_simd.json").read_text())
  assert set(choice.values()) <= {2, 3, 4} and choice["nvp_36_2"] == 4, choice
  got = {n: md["vgpr_count"] for n, md in code_object.kernels(ROOT / "mjlab_amd" / "csrc" / "libmjlab_amd.so").items() if "k_control_step_coneILi36E" in n}
  assert list(got.values()) == [128], got  # (and the library really is that build)
