"""Device-side mjModel / mjData: torch tensors + the ctypes structs the C ABI takes.

What ``mjwarp.put_model`` / ``mjwarp.put_data`` do for the reference (src/mjlab/sim/sim.py:107-119):
every field of include/mjlab_fields.h becomes one torch tensor on the GPU and one pointer in
``mjlab_model_t`` / ``mjlab_data_t``.  Shared by ``Simulation`` and the ``mjwarp``-shaped facade
(mjwarp_compat.py).
"""

from __future__ import annotations

import numpy as np
import torch

from . import _abi, native
from .mjcf import SOL_PGS, Model

EXTRA_MODEL_FIELDS = ("geom_rgba",)  # DR-able host fields not consumed by the kernels


def shape_view(f: _abi.FieldSpec, flat: torch.Tensor, n: int) -> torch.Tensor:
  """(nworld, n * ncol) storage -> the shape mjData documents for the field."""
  nw = flat.shape[0]
  if f.count == "one":
    return flat.view(nw) if f.ncol == 1 else flat.view(nw, f.ncol)
  if f.count == "nvnv":
    nv = int(round(n**0.5))
    return flat.view(nw, nv, nv)
  if f.count == "njmaxnv":
    return flat  # (nworld, njmax * nv), reshaped by users that need it
  if f.ncol == 1:
    return flat.view(nw, n)
  if f.ncol == 9 and f.name.endswith("xmat") or f.name in ("ximat",):
    return flat.view(nw, n, 3, 3)
  return flat.view(nw, n, f.ncol)


def upload_model(model: Model, num_envs: int, nconmax: int, njmax: int, dev: torch.device):
  """-> (mjlab_model_t, base tensors, per-world views).  Float fields are stored once
  (``(1, n...)``, world stride 0) and viewed as ``(num_envs, n...)`` broadcasts until expanded."""
  mf, _, MS, _ = native.layouts()
  with torch.cuda.device(dev):
    m = MS()
    m.size = _abi.fill_sizes(model, num_envs, nconmax, njmax)
    m.opt = _abi.fill_option(model)
    base: dict[str, torch.Tensor] = {}
    view: dict[str, torch.Tensor] = {}
    for f in mf:
      if f.kind == "i":
        t = torch.from_numpy(_abi.model_int_array(model, f.name)).to(dev)
        base[f.name] = t
        view[f.name] = t
        setattr(m, f.name, t.data_ptr())
      else:
        host = np.ascontiguousarray(getattr(model, f.name), dtype=np.float32)
        t = torch.from_numpy(host).to(dev).unsqueeze(0).contiguous()
        base[f.name] = t
        view[f.name] = t.expand(num_envs, *t.shape[1:])
        setattr(m, f.name, t.data_ptr())
        setattr(m, f.name + "_ws", 0)
    for name in EXTRA_MODEL_FIELDS:
      host = np.ascontiguousarray(getattr(model, name), dtype=np.float32)
      t = torch.from_numpy(host).to(dev).unsqueeze(0).contiguous()
      base[name] = t
      view[name] = t.expand(num_envs, *t.shape[1:])
  return m, base, view


def alloc_data(model: Model, num_envs: int, nconmax: int, njmax: int, dev: torch.device):
  """-> (mjlab_data_t, tensors): zeroed ``(num_envs, ...)`` arrays, ``qpos`` at ``qpos0``."""
  _, df, _, DS = native.layouts()
  with torch.cuda.device(dev):
    d = DS()
    data: dict[str, torch.Tensor] = {}
    for f in df:
      n = _abi.count_of(f.count, model, nconmax, njmax)
      dtype = torch.int32 if f.kind == "i" else torch.float32
      if f.name == "efc_B" and model.opt.solver != SOL_PGS:
        # njmax x nv floats per world that only the dual solver reads (172 MB at 4096 G1 worlds): NULL for the primal solvers
        # (the library refuses MJLAB_SOL_PGS with a NULL efc_B)
        setattr(d, f.name, None)
        data[f.name] = torch.zeros((num_envs, 0), dtype=dtype, device=dev)
        continue
      flat = torch.zeros((num_envs, n * f.ncol), dtype=dtype, device=dev)
      setattr(d, f.name, flat.data_ptr())
      data[f.name] = shape_view(f, flat, n)
    data["qpos"][:] = torch.from_numpy(model.qpos0.astype(np.float32)).to(dev)
    # activation state: na = 0 for every supported actuator (no dynamics), kept for API parity
    # (reference entity/data.py reads data.act)
    data["act"] = torch.zeros((num_envs, int(getattr(model, "na", 0))), dtype=torch.float32, device=dev)
  return d, data


def expand_field(m_struct, base: dict, view: dict, model: Model, name: str, num_envs: int, dev: torch.device, stream: int) -> bool:
  """Per-world copy of one float model field (reference sim/randomization.py:20-55).  Returns False
  when it already is per world."""
  lib = native.lib()
  mfields = {f.name for f in native.layouts()[0]}
  if name not in base:
    host = np.ascontiguousarray(getattr(model, name), dtype=np.float32)
    base[name] = torch.from_numpy(host).to(dev).unsqueeze(0).contiguous()
  b = base[name]
  if b.dtype != torch.float32:
    raise ValueError(f"Field '{name}' is an integer topology field and cannot be per-world")
  if b.shape[0] == num_envs:
    return False
  nelem = b[0].numel()
  dst = torch.empty((num_envs, *b.shape[1:]), dtype=b.dtype, device=dev)
  native.check(lib.mjlab_tile_field(dst.data_ptr(), b.data_ptr(), nelem, num_envs, 4, stream), "mjlab_tile_field")
  base[name] = dst
  view[name] = dst
  if name in mfields:
    setattr(m_struct, name, dst.data_ptr())
    setattr(m_struct, name + "_ws", int(nelem))
  return True
