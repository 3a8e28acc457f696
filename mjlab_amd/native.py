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
SOURCES = [CSRC / "mjlab_amd.hip"]
HEADERS = [Path(__file__).parents[1] / "include" / "mjlab_amd.h", Path(__file__).parents[1] / "include" / "mjlab_fields.h",
           *sorted(CSRC.glob("*.h"))]  # the stage files are included by mjlab_amd.hip (one translation unit)

STAGE_POSITION, STAGE_COLLISION, STAGE_VELOCITY, STAGE_CONSTRAINT, STAGE_SOLVE, STAGE_INTEGRATE = 1, 2, 4, 8, 16, 32
STAGE_FORWARD, STAGE_STEP = 31, 63


def build(force: bool = False, verbose: bool = False) -> Path:
  """Compile the HIP extension for gfx950 (cross-compiles without a GPU)."""
  newest_src = max(p.stat().st_mtime for p in SOURCES + HEADERS)
  if LIB_PATH.exists() and not force and LIB_PATH.stat().st_mtime >= newest_src:
    return LIB_PATH
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  cmd = [
    hipcc, "-O3", "-std=c++17", "-ffp-contract=on", "--offload-arch=gfx950", "-shared", "-fPIC",
    "-o", str(LIB_PATH), *[str(s) for s in SOURCES],
  ]  # fmt: skip
  if verbose:
    print(" ".join(cmd))
  subprocess.run(cmd, check=True)
  return LIB_PATH


class NativeLibraryError(RuntimeError):
  pass


_LIB: ctypes.CDLL | None = None


def lib() -> ctypes.CDLL:
  """The loaded extension; raises NativeLibraryError (never falls back) when unavailable."""
  global _LIB
  if _LIB is not None:
    return _LIB
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
  L.mjlab_control_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
  L.mjlab_forward_stages.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
  L.mjlab_tile_field.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
  L.mjlab_selftest.argtypes = [ctypes.c_void_p]
  L.mjlab_lds_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
  if L.mjlab_abi_version() != 1:
    raise NativeLibraryError("ABI version mismatch")
  _LIB = L
  return L


EXPORTED_SYMBOLS = (
  "mjlab_abi_version", "mjlab_last_error", "mjlab_model_layout", "mjlab_data_layout", "mjlab_sizeof_model",
  "mjlab_sizeof_data", "mjlab_step", "mjlab_forward", "mjlab_forward_masked", "mjlab_entity_readback", "mjlab_masked_reset", "mjlab_interval_push", "mjlab_control_step", "mjlab_forward_stages", "mjlab_tile_field", "mjlab_lds_bytes", "mjlab_selftest",
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
