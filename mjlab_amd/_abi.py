"""ctypes mirror of include/mjlab_fields.h, built from the library's own layout string.

The C side is the single source of truth: ``*_model_layout()`` / ``*_data_layout()``
return ``"kind:name:ncol:count,"`` items in struct order, and the ctypes ``Structure``
classes are generated from them, so adding a field to the header cannot silently
desynchronise the two sides.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass

from .mjcf import TCAND_MAX, Model

_SIZE_FIELDS = (
  "nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite", "nsensor", "nsensordata", "npair",
  "nlevel", "nworld", "nconmax", "njmax",
  "nstaticgeom", "geom_lds0", "nstaticsite", "nterrain", "ntgeom", "ntcell", "ntcellp1", "ntitem", "tgrid_nx", "tgrid_ny",
)  # fmt: skip


class Sizes(ctypes.Structure):
  _fields_ = [(n, ctypes.c_int) for n in _SIZE_FIELDS]


class Option(ctypes.Structure):
  _fields_ = [
    ("timestep", ctypes.c_double),
    ("gravity", ctypes.c_double * 3),
    ("impratio", ctypes.c_double),
    ("tolerance", ctypes.c_double),
    ("ls_tolerance", ctypes.c_double),
    ("meaninertia", ctypes.c_double),
    ("tgrid_x0", ctypes.c_double),
    ("tgrid_y0", ctypes.c_double),
    ("tgrid_cell", ctypes.c_double),
    ("ls_parallel_min_step", ctypes.c_double),
    ("iterations", ctypes.c_int),
    ("ls_iterations", ctypes.c_int),
    ("integrator", ctypes.c_int),
    ("cone", ctypes.c_int),
    ("flags", ctypes.c_int),
    ("solver", ctypes.c_int),
  ]


# mjlab_option_t.flags (include/mjlab_fields.h)
OPT_FOLD_FORWARD, OPT_LITERAL_TERMINATION, OPT_WARMSTART_AT_ADVANCE, OPT_FUSE_PRESOLVE, OPT_FUSE_STEP, OPT_FRICTIONLOSS, OPT_LS_PARALLEL, OPT_WORLD_FRAME = 1, 2, 4, 8, 16, 32, 64, 128
OPT_LS_LITERAL_COST = 256
# mjlab_data_t.overflow bits
OVF_NCONMAX, OVF_NJMAX, OVF_TCAND = 1, 2, 4


@dataclass(frozen=True)
class FieldSpec:
  kind: str  # "i" int32 | "r" real
  name: str
  ncol: int
  count: str  # size symbol


def parse_layout(layout: str) -> list[FieldSpec]:
  out = []
  for item in layout.strip(",").split(","):
    kind, name, ncol, count = item.split(":")
    out.append(FieldSpec(kind, name, int(ncol), count))
  return out


def make_model_struct(fields: list[FieldSpec]) -> type[ctypes.Structure]:
  fl: list[tuple] = [("size", Sizes), ("opt", Option)]
  for f in fields:
    fl.append((f.name, ctypes.c_void_p))
    if f.kind == "r":
      fl.append((f.name + "_ws", ctypes.c_int))
  return type("ModelStruct", (ctypes.Structure,), {"_fields_": fl})


def make_data_struct(fields: list[FieldSpec]) -> type[ctypes.Structure]:
  return type("DataStruct", (ctypes.Structure,), {"_fields_": [(f.name, ctypes.c_void_p) for f in fields]})


def count_of(sym: str, m: Model, nconmax: int, njmax: int) -> int:
  if sym == "one":
    return 1
  if sym == "nlevelp1":
    return m.nlevel + 1
  if sym == "nvnv":
    return m.nv * m.nv
  if sym == "njmaxnv":
    return njmax * m.nv
  if sym == "nconmax":
    return nconmax
  if sym == "njmax":
    return njmax
  return int(getattr(m, sym))


def fill_sizes(m: Model, nworld: int, nconmax: int, njmax: int) -> Sizes:
  s = Sizes()
  for n in _SIZE_FIELDS[:11]:
    setattr(s, n, int(getattr(m, n)))
  s.nworld, s.nconmax, s.njmax = nworld, nconmax, njmax
  for n in _SIZE_FIELDS[14:]:
    setattr(s, n, int(getattr(m, n)))
  return s


def fill_option(m: Model) -> Option:
  o = Option()
  o.timestep = m.opt.timestep
  o.gravity[:] = m.opt.gravity
  o.impratio = m.opt.impratio
  o.tolerance = m.opt.tolerance
  o.ls_tolerance = m.opt.ls_tolerance
  o.meaninertia = m.meaninertia
  o.tgrid_x0, o.tgrid_y0, o.tgrid_cell = m.tgrid_x0, m.tgrid_y0, m.tgrid_cell
  o.ls_parallel_min_step = float(getattr(m.opt, "ls_parallel_min_step", 1.0e-6))
  o.iterations = m.opt.iterations
  o.ls_iterations = m.opt.ls_iterations
  o.integrator = m.opt.integrator
  o.cone = m.opt.cone
  o.solver = m.opt.solver
  import numpy as np

  o.flags = OPT_FRICTIONLOSS if np.any(np.asarray(m.dof_frictionloss) != 0) else 0
  return o


def model_int_array(m: Model, name: str):
  """Host int32 array for a model int field (body_dofmask is split into lo/hi words)."""
  import numpy as np

  if name == "body_dofmask":
    mask = m.body_dofmask.astype(np.uint64)
    lo = (mask & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
    hi = (mask >> np.uint64(32)).astype(np.uint32).view(np.int32)
    return np.ascontiguousarray(np.stack([lo, hi], axis=1))
  return np.ascontiguousarray(getattr(m, name), dtype=np.int32)


def default_capacities(m: Model, nconmax: int | None, njmax: int | None) -> tuple[int, int]:
  """Per-world contact / constraint-row capacities.

  The reference passes ``nconmax`` (a pool shared by all worlds upstream) and ``njmax``
  (rows per world) to ``mjwarp.put_data`` (src/mjlab/sim/sim.py:113-119).  Here both
  capacities are per world: ``njmax`` rows and ``njmax`` contacts (a contact yields at
  least one row), bounded by what the model can ever produce.
  """
  max_rows = 2 * int((m.jnt_limited != 0).sum()) + 4 * 4 * (m.npair + m.ntgeom * TCAND_MAX)
  if njmax is None:
    njmax = max(1, min(max_rows, 512))
  njmax = max(1, min(int(njmax), max(max_rows, 1)))
  ncon = max(1, min(njmax, 4 * max(m.npair + m.ntgeom * TCAND_MAX, 1)))
  return ncon, njmax
