"""Loader / builder of the HIP extension (mjlab_amd/csrc/libmjlab_amd.so).

The product path has NO CPU fallback: if the shared library is missing or does not load,
importing it raises, and ``Simulation`` refuses to construct on a non-GPU device.
"""

from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from pathlib import Path

from . import _abi

CSRC = Path(__file__).parent / "csrc"
LIB_PATH = Path(os.environ.get("MJLAB_AMD_LIB", CSRC / "libmjlab_amd.so"))
SOURCES = [CSRC / "mjlab_amd.hip", CSRC / "nvp_inst.hip"]
HEADERS = [Path(__file__).parents[1] / "include" / "mjlab_amd.h", Path(__file__).parents[1] / "include" / "mjlab_fields.h",
           *sorted(CSRC.glob("*.h"))]  # kernels.h includes the stage files
NVP_SIZES = (8, 16, 20, 24, 32, 36, 40, 48, 64)  # padded dof counts the solve / substep / control kernels are instantiated for
HIPCC_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=on", "--offload-arch=gfx950", "-fPIC"]

ABI_VERSION = 4  # include/mjlab_amd.h MJLAB_ABI_VERSION

STAGE_POSITION, STAGE_COLLISION, STAGE_VELOCITY, STAGE_CONSTRAINT, STAGE_SOLVE, STAGE_INTEGRATE = 1, 2, 4, 8, 16, 32
STAGE_FORWARD, STAGE_STEP = 31, 63


def build(force: bool = False, verbose: bool = False, out: Path | None = None, defines: tuple[str, ...] = (), jobs: int | None = None,
          only_size: int | None = None) -> Path:
  """Compile the HIP extension for gfx950 (cross-compiles without a GPU): mjlab_amd.hip (C ABI + the kernels that
  do not depend on the padded dof count) and nvp_inst.hip twice per padded size, in parallel, then one link.
  ``out`` / ``defines`` build a variant somewhere else (profiling build: ``defines=("MJLAB_PROFILE",)``)."""
  from concurrent.futures import ThreadPoolExecutor

  out = Path(out) if out is not None else LIB_PATH
  newest_src = max(p.stat().st_mtime for p in SOURCES + HEADERS)
  if out.exists() and not force and not defines and out.stat().st_mtime >= newest_src:
    return out
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  objdir = CSRC / "build" / (out.stem + ("_" + "_".join(defines) if defines else ""))
  objdir.mkdir(parents=True, exist_ok=True)
  # One builder at a time per output (torchrun: every rank may find the library stale at once -- ADVICE round 4): an exclusive
  # file lock around compile + link, the link goes to a temporary name and is moved into place atomically, and a rank that waited
  # for the lock finds the finished library and returns.
  import fcntl

  lock = open(objdir / ".build.lock", "w")
  fcntl.flock(lock, fcntl.LOCK_EX)
  try:
    if out.exists() and not force and not defines and out.stat().st_mtime >= newest_src:
      return out
    return _build_locked(out, objdir, hipcc, verbose, defines, jobs, only_size)
  finally:
    fcntl.flock(lock, fcntl.LOCK_UN)
    lock.close()


def _build_locked(out: Path, objdir: Path, hipcc: str, verbose: bool, defines: tuple[str, ...], jobs: int | None, only_size: int | None) -> Path:
  from concurrent.futures import ThreadPoolExecutor

  dflags = [f"-D{x}" for x in defines] + ([f"-DMJLAB_NVP_ONLY={only_size}"] if only_size else [])
  dflags += os.environ.get("MJLAB_HIPCC_EXTRA", "").split()  # compiler-flag experiments
  sizes = (only_size,) if only_size else NVP_SIZES  # only_size: an experiment library for models of one padded size (A/B runs)
  units = [(CSRC / "mjlab_amd.hip", objdir / "abi.o", [])] + [
    (CSRC / "nvp_inst.hip", objdir / f"nvp_{n}_{part}.o", [f"-DMJLAB_NVP={n}", f"-DMJLAB_NVP_PART={part}"]) for n in reversed(sizes) for part in (1, 0, 2)
  ]  # largest first: the 64-dof instantiations are the critical path; part 2 = the elliptic-cone kernels, a unit of their own so that
  # the inliner sees the pyramid's kernels (parts 0 and 1: the measured path) among exactly the callers they always had

  cone_choice: dict[str, int] = {}

  def compile_one(unit):
    src, obj, extra = unit
    # The elliptic-cone kernels (part 2) run 27 % faster at four waves per SIMD than at two (bench.py --cone elliptic: 2.10 against 1.65 M
    # env-steps/s, profiles/r06_cone) -- but at that register budget hipcc miscompiles a spill store in SOME instantiations (it executes
    # under EXEC == 0: DESIGN.md section 7, mjlab_amd/code_check.py).  So every cone unit is built for four waves, its code object is
    # checked, and a unit that carries the pattern is rebuilt for three, then for the spill-free two (kernels.h: CONE_WAVES' default).
    cone = "-DMJLAB_NVP_PART=2" in extra and not any(x.startswith("-DMJLAB_CONE_WPE") for x in dflags)
    for wpe in ((4, 3, 0) if cone else (0,)):
      cmd = [hipcc, *HIPCC_FLAGS, *dflags, *extra, *([f"-DMJLAB_CONE_WPE={wpe}"] if wpe else []), "-c", str(src), "-o", str(obj)]
      if verbose:
        print(" ".join(cmd), flush=True)
      subprocess.run(cmd, check=True)
      if not wpe:
        break
      from . import code_check

      bad = code_check.fatal_hits(obj)
      if not bad:
        cone_choice[obj.stem] = wpe
        break
      if verbose:
        print(f"{obj.name}: {sum(map(len, bad.values()))} spill store(s) ahead of their EXEC restore at {wpe} waves per SIMD ({', '.join(bad)}): rebuilding with fewer", flush=True)
    if cone:
      cone_choice.setdefault(obj.stem, 2)

  with ThreadPoolExecutor(max_workers=jobs or min(len(units), os.cpu_count() or 4)) as pool:
    list(pool.map(compile_one, units))
  tmp = out.with_name(out.name + f".tmp{os.getpid()}")
  cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp), *[str(u[1]) for u in units]]
  if verbose:
    print(" ".join(cmd), flush=True)
  subprocess.run(cmd, check=True)
  os.replace(tmp, out)  # (a process that has the old library mapped keeps its inode; nobody ever sees a half-written file)
  if cone_choice:
    import json

    out.with_name(out.stem + ".cone_waves_per_simd.json").write_text(  # next to the library: travels with it, like the .so git-ignored
json.dumps(dict(sorted(cone_choice.items())), indent=1) + "\n")
  return out


class NativeLibraryError(RuntimeError):
  pass


_LIB: ctypes.CDLL | None = None


def lib() -> ctypes.CDLL:
  """The loaded extension; raises NativeLibraryError (never falls back) when unavailable."""
  global _LIB
  if _LIB is not None:
    return _LIB
  stale = LIB_PATH.exists() and "MJLAB_AMD_LIB" not in os.environ and shutil.which("hipcc") is not None \
    and LIB_PATH.stat().st_mtime < max(p.stat().st_mtime for p in SOURCES + HEADERS)
  if stale and not os.environ.get("MJLAB_AMD_NO_AUTOBUILD"):  # a library older than its sources is rebuilt, never driven with newer struct mirrors
    build()
  if not LIB_PATH.exists():
    if os.environ.get("MJLAB_AMD_NO_AUTOBUILD"):
      raise NativeLibraryError(f"{LIB_PATH} is missing; run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
      build()
    except Exception as e:  # noqa: BLE001
      raise NativeLibraryError(f"could not build {LIB_PATH}: {e}") from e
  try:
    L = ctypes.CDLL(str(LIB_PATH))
  except OSError as e:
    raise NativeLibraryError(f"could not load {LIB_PATH}: {e}") from e
  L.mjlab_last_error.restype = ctypes.c_char_p
  L.mjlab_model_layout.restype = ctypes.c_char_p
  L.mjlab_data_layout.restype = ctypes.c_char_p
  L.mjlab_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
  L.mjlab_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
  L.mjlab_forward_masked.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
  L.mjlab_entity_readback.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
  L.mjlab_masked_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]
  L.mjlab_interval_push.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
  vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
  L.mjlab_event_reset_root_state_uniform.argtypes = [vp, ci, ci, vp, ci, ci, ci, vp, vp, ci, vp, vp, ci, vp, vp, vp]
  L.mjlab_event_reset_joints_by_scale.argtypes = [vp, ci, vp, ci, ci, vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp]
  L.mjlab_event_push_by_setting_velocity.argtypes = [vp, ci, ci, ci, vp, cf, vp, vp, ci, vp, ci, vp, ci, vp, vp]
  L.mjlab_command_uniform_velocity.argtypes = [vp, vp]
  L.mjlab_command_motion_write.argtypes = [vp, vp, ci, ci, vp, ci, ci, vp, vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, vp, cf, cf, vp]
  L.mjlab_masked_fill_rows.argtypes = [vp, ci, vp, ci, vp]
  L.mjlab_masked_sums.argtypes = [vp, ci, vp, ci, vp, vp]
  L.mjlab_reward_accumulate.argtypes = [vp, vp, vp, ci, ci, cf, vp, vp, vp, ci, vp]
  L.mjlab_command_motion_frame.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
  L.mjlab_copy_batch.argtypes = [vp, ci, vp]
  L.mjlab_command_motion_metrics.argtypes = [vp, vp]
  L.mjlab_command_motion_sample.argtypes = [vp, vp]
  L.mjlab_command_motion_sampler.argtypes = [vp, vp]
  L.mjlab_flag_to_mask.argtypes = [vp, ci, vp, vp]
  L.mjlab_log_finish.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp]
  L.mjlab_command_motion_relative.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, ci, vp, vp, ci, vp]
  L.mjlab_control_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
  L.mjlab_forward_stages.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
  L.mjlab_tile_field.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
  L.mjlab_selftest.argtypes = [ctypes.c_void_p]
  L.mjlab_chol_selftest.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
  L.mjlab_poison_scratch.argtypes = [ctypes.c_int, ctypes.c_void_p]
  L.mjlab_lds_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
  if L.mjlab_abi_version() != ABI_VERSION:
    raise NativeLibraryError(f"{LIB_PATH}: ABI version {L.mjlab_abi_version()}, this package speaks {ABI_VERSION}; rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
  # the two host structs and the control-step structs are mirrored by hand in ctypes (_abi.Option / _abi.Sizes, rollout._Control /
  # _MotionReset): a library built from other headers would be driven with shifted fields -- refuse it here instead
  from .env_terms import MotionMetricsArgs, MotionSampleArgs, MotionSamplerArgs, MotionTables, VelocityCommand
  from .rollout import _Control as Control, _MotionReset as MotionReset  # (imported here: rollout imports this module)

  for what, mine, theirs in (("mjlab_option_t", ctypes.sizeof(_abi.Option), L.mjlab_sizeof_option()), ("mjlab_sizes_t", ctypes.sizeof(_abi.Sizes), L.mjlab_sizeof_sizes()),
                             ("mjlab_control_t", ctypes.sizeof(Control), L.mjlab_sizeof_control()), ("mjlab_motion_reset_t", ctypes.sizeof(MotionReset), L.mjlab_sizeof_motion_reset()),
                             ("mjlab_velocity_command_t", ctypes.sizeof(VelocityCommand), L.mjlab_sizeof_velocity_command()),
                             ("mjlab_motion_tables_t", ctypes.sizeof(MotionTables), L.mjlab_sizeof_motion_tables()),
                             ("mjlab_motion_metrics_t", ctypes.sizeof(MotionMetricsArgs), L.mjlab_sizeof_motion_metrics()),
                             ("mjlab_motion_sample_t", ctypes.sizeof(MotionSampleArgs), L.mjlab_sizeof_motion_sample()),
                             ("mjlab_motion_sampler_t", ctypes.sizeof(MotionSamplerArgs), L.mjlab_sizeof_motion_sampler())):
    if mine != theirs:
      raise NativeLibraryError(f"{LIB_PATH}: sizeof({what}) is {theirs} in the library, {mine} in the Python mirror; rebuild the library")
  _LIB = L
  return L


EXPORTED_SYMBOLS = (
  "mjlab_abi_version", "mjlab_last_error", "mjlab_model_layout", "mjlab_data_layout", "mjlab_sizeof_model",
  "mjlab_sizeof_data", "mjlab_sizeof_option", "mjlab_sizeof_sizes", "mjlab_step", "mjlab_forward", "mjlab_forward_masked", "mjlab_entity_readback", "mjlab_masked_reset", "mjlab_interval_push", "mjlab_control_step", "mjlab_sizeof_control", "mjlab_sizeof_motion_reset", "mjlab_event_reset_root_state_uniform", "mjlab_event_reset_joints_by_scale", "mjlab_event_push_by_setting_velocity", "mjlab_command_uniform_velocity", "mjlab_sizeof_velocity_command", "mjlab_command_motion_write", "mjlab_command_motion_relative", "mjlab_command_motion_frame", "mjlab_copy_batch", "mjlab_command_motion_metrics", "mjlab_sizeof_motion_metrics", "mjlab_command_motion_sample", "mjlab_sizeof_motion_sample", "mjlab_command_motion_sampler", "mjlab_sizeof_motion_sampler", "mjlab_flag_to_mask", "mjlab_log_finish", "mjlab_sizeof_motion_tables", "mjlab_reward_accumulate", "mjlab_masked_fill_rows", "mjlab_masked_sums", "mjlab_forward_stages", "mjlab_tile_field", "mjlab_lds_bytes", "mjlab_selftest", "mjlab_chol_selftest", "mjlab_poison_scratch",
)  # fmt: skip


def layouts():
  L = lib()
  mf = _abi.parse_layout(L.mjlab_model_layout().decode())
  df = _abi.parse_layout(L.mjlab_data_layout().decode())
  MS, DS = _abi.make_model_struct(mf), _abi.make_data_struct(df)
  if ctypes.sizeof(MS) != L.mjlab_sizeof_model() or ctypes.sizeof(DS) != L.mjlab_sizeof_data():
    raise NativeLibraryError("struct layout mismatch between Python and the HIP extension")
  return mf, df, MS, DS


def check(rc: int, what: str) -> None:
  if rc != 0:
    raise RuntimeError(f"{what} failed: {lib().mjlab_last_error().decode()}")


if __name__ == "__main__":  # python -m mjlab_amd.native [--out lib.so] [-DNAME ...]
  import sys

  a = sys.argv[1:]
  o = Path(a[a.index("--out") + 1]) if "--out" in a else None
  only = int(a[a.index("--only") + 1]) if "--only" in a else None
  print(build(force=True, verbose="-v" in a, out=o, defines=tuple(x[2:] for x in a if x.startswith("-D")), only_size=only))
