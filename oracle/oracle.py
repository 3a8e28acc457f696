"""ctypes front-end of the CPU oracle (oracle/mjoracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: see the header of oracle/mjoracle.c.
"""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

from mjlab_amd import _abi
from mjlab_amd.mjcf import Model

_DIR = Path(__file__).parent


def build() -> None:
  subprocess.run(["make", "-C", str(_DIR), "-s"], check=True)


def _load(precision: str) -> ctypes.CDLL:
  path = _DIR / f"libmjoracle_{precision}.so"
  if not path.exists():
    build()
  lib = ctypes.CDLL(str(path))
  lib.mjo_model_layout.restype = ctypes.c_char_p
  lib.mjo_data_layout.restype = ctypes.c_char_p
  return lib


class OracleSim:
  """Batch of independent worlds stepped on the CPU by the oracle.

  Data arrays are numpy, shape ``(nworld, ...)`` like the device tensors of
  ``mjlab_amd.sim.Simulation`` (and like ``mjwarp.Data`` in the reference).
  """

  def __init__(self, model: Model, nworld: int = 1, nconmax: int | None = None, njmax: int | None = None, precision: str = "f64",
               flags: int = 0, ls_parallel: bool = False):
    self.model = model
    self.nworld = nworld
    self.lib = _load(precision)
    self.real = np.float64 if precision == "f64" else np.float32
    assert self.lib.mjo_sizeof_real() == np.dtype(self.real).itemsize
    self.nconmax, self.njmax = _abi.default_capacities(model, nconmax, njmax)
    mfields = _abi.parse_layout(self.lib.mjo_model_layout().decode())
    dfields = _abi.parse_layout(self.lib.mjo_data_layout().decode())
    MS, DS = _abi.make_model_struct(mfields), _abi.make_data_struct(dfields)
    assert ctypes.sizeof(MS) == self.lib.mjo_sizeof_model()
    assert ctypes.sizeof(DS) == self.lib.mjo_sizeof_data()
    self._m = MS()
    self._m.size = _abi.fill_sizes(model, nworld, self.nconmax, self.njmax)
    self._m.opt = _abi.fill_option(model)
    if ls_parallel:
      flags |= _abi.OPT_LS_PARALLEL  # mujoco_warp's parallel grid search instead of MuJoCo's exact iterative one (default)
    self._m.opt.flags |= flags  # MJLAB_OPT_LITERAL_TERMINATION / MJLAB_OPT_WARMSTART_AT_ADVANCE (fold is a device-side mechanism)
    self.mfield: dict[str, np.ndarray] = {}
    for f in mfields:
      if f.kind == "i":
        arr = _abi.model_int_array(model, f.name)
      else:
        arr = np.ascontiguousarray(getattr(model, f.name), dtype=self.real)
        setattr(self._m, f.name + "_ws", 0)
      self.mfield[f.name] = arr
      setattr(self._m, f.name, arr.ctypes.data)
    self._d = DS()
    self.dfield: dict[str, np.ndarray] = {}
    for f in dfields:
      n = _abi.count_of(f.count, model, self.nconmax, self.njmax)
      shape = (nworld, n, f.ncol) if f.ncol > 1 else (nworld, n)
      arr = np.zeros(shape, dtype=np.int32 if f.kind == "i" else self.real)
      self.dfield[f.name] = arr
      setattr(self._d, f.name, arr.ctypes.data)
    self.reset()

  def __getattr__(self, name: str) -> np.ndarray:
    df = self.__dict__.get("dfield", {})
    if name in df:
      return df[name]
    raise AttributeError(name)

  def expand_model_field(self, name: str) -> np.ndarray:
    """Per-world copy of a real model field (the oracle-side analogue of expand_model_fields)."""
    base = np.ascontiguousarray(getattr(self.model, name), dtype=self.real)
    arr = np.ascontiguousarray(np.broadcast_to(base, (self.nworld,) + base.shape)).copy()
    self.mfield[name] = arr
    setattr(self._m, name, arr.ctypes.data)
    setattr(self._m, name + "_ws", int(base.size))
    if name == "dof_frictionloss":
      self._m.opt.flags |= _abi.OPT_FRICTIONLOSS  # values may now be written: build the rows (as Simulation does)
    if name in ("geom_pos", "geom_quat", "body_pos", "body_quat"):
      self._m.size.nstaticgeom = 0  # static geoms may now differ per world: pose them every pass (as Simulation does)
    if name in ("site_pos", "site_quat", "body_pos", "body_quat"):
      self._m.size.nstaticsite = 0
    return arr

  def reset(self, key: int | None = None) -> None:
    for a in self.dfield.values():
      a[...] = 0
    m = self.model
    self.qpos[:] = m.qpos0 if key is None else m.key_qpos[key]
    if key is not None:
      self.qvel[:] = m.key_qvel[key]
      self.ctrl[:] = m.key_ctrl[key]
    self.lib.mjo_static_geoms(ctypes.byref(self._m), ctypes.byref(self._d))

  def forward(self, nthread: int = 1) -> None:
    self.lib.mjo_run(ctypes.byref(self._m), ctypes.byref(self._d), 0, nthread)

  def step(self, nstep: int = 1, nthread: int = 1) -> None:
    self.lib.mjo_run(ctypes.byref(self._m), ctypes.byref(self._d), nstep, nthread)
