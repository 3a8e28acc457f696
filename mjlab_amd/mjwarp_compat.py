"""``mujoco_warp``-shaped entry points over the C ABI: the four calls the reference makes into its
engine (src/mjlab/sim/sim.py:107 ``put_model``, :113-119 ``put_data``, :136/195 ``step``, :139/187
``forward``) and the field expansion of sim/randomization.py:20-55, with the same argument order.

  from mjlab_amd import mjwarp_compat as mjwarp
  m = mjwarp.put_model(mj_model)                                  # mujoco.MjModel or mjlab_amd.mjcf.Model
  d = mjwarp.put_data(mj_model, mj_data, nworld=4096, nconmax=None, njmax=300)
  mjwarp.step(m, d); mjwarp.forward(m, d)
  d.qpos, m.geom_friction ...                                      # torch tensors, (nworld, ...) like mjwarp's arrays

The arrays ARE torch tensors (no ``wp.to_torch`` wrapping needed); launches go to PyTorch's current
stream on the tensors' device.  ``Simulation`` (sim.py) is the full mirror of the reference class;
this module is the seam a maintainer patches when they keep the reference's own ``Simulation``.
"""

from __future__ import annotations

import ctypes
from types import SimpleNamespace
from typing import Any

import torch

from . import _abi, device_state, native
from .mjcf import Model as HostModel


def _host(mjm: Any) -> HostModel:
  if isinstance(mjm, HostModel):
    return mjm
  from .from_mujoco import model_from_mujoco

  return model_from_mujoco(mjm)


class Model:
  """Device model: attribute access yields the field's tensor.  Like mjwarp's, float fields have a
  leading world dimension of size 1 until ``expand_model_fields`` gives them per-world storage
  (reference sim/sim_data.py:20-26 broadcasts that dimension in its bridge); int topology fields
  have none."""

  def __init__(self, host: HostModel, device: torch.device) -> None:
    from .sim import check_supported

    check_supported(host)
    self.__dict__.update(host=host, device=device, nworld=1, _expanded=set())
    struct, base, view = device_state.upload_model(host, 1, 1, 1, device)
    self.__dict__.update(struct=struct, _base=base, _view=view)
    struct.opt.flags |= _abi.OPT_FRICTIONLOSS  # nothing on this seam sees a later write to dof_frictionloss: always read it
    # the reference sets wp_model.opt.ls_parallel (sim/sim.py:111): read at every step() / forward() (_sync_opt).
    # "Forward folded into the next step" stays
    # OFF on this seam: nothing here sees writes to model arrays between forward() and step()
    # (Simulation hooks its model bridge for that); set struct.opt.flags |= _abi.OPT_FOLD_FORWARD to opt in.
    self.__dict__["opt"] = SimpleNamespace(**host.opt.__dict__, ls_parallel=False)

  def __getattr__(self, name: str) -> Any:
    view = self.__dict__.get("_view", {})
    if name in view:
      return self._base[name]
    if hasattr(self.host, name):
      return getattr(self.host, name)
    raise AttributeError(name)

  def __setattr__(self, name: str, value: Any) -> None:
    raise AttributeError("model fields are fixed-address device arrays: write into them (m.field[...] = v)")


class Data:
  """Device data: one ``(nworld, ...)`` tensor per mjData field."""

  def __init__(self, host: HostModel, nworld: int, nconmax: int, njmax: int, device: torch.device) -> None:
    struct, tensors = device_state.alloc_data(host, nworld, nconmax, njmax, device)
    self.__dict__.update(struct=struct, _tensors=tensors, nworld=nworld, nconmax=nconmax, njmax=njmax, device=device, _static_done=False)

  def __getattr__(self, name: str) -> Any:
    t = self.__dict__.get("_tensors", {})
    if name in t:
      return t[name]
    raise AttributeError(name)

  def __setattr__(self, name: str, value: Any) -> None:
    if name == "_static_done":
      self.__dict__[name] = value
      return
    raise AttributeError("data fields are fixed-address device arrays: write into them (d.field[...] = v)")


def put_model(mjm: Any, device: str | torch.device | None = None) -> Model:
  dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
  if dev.type != "cuda" or not torch.cuda.is_available():
    raise RuntimeError("mjlab_amd needs a ROCm GPU device; there is no CPU fallback")
  return Model(_host(mjm), dev)


def put_data(mjm: Any, mjd: Any = None, nworld: int = 1, nconmax: int | None = None, njmax: int | None = None,
             device: str | torch.device | None = None) -> Data:  # fmt: skip
  """``mjd`` (a host mjData) seeds qpos / qvel / ctrl of every world when given, like upstream."""
  host = _host(mjm)
  dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
  ncon, nj = _abi.default_capacities(host, nconmax, njmax)
  d = Data(host, nworld, ncon, nj, dev)
  if mjd is not None:
    import numpy as np

    for f in ("qpos", "qvel", "ctrl"):
      v = getattr(mjd, f, None)
      if v is not None and np.asarray(v).size:
        getattr(d, f)[:] = torch.as_tensor(np.asarray(v), dtype=torch.float32, device=dev)
  return d


def _bind(m: Model, d: Data) -> None:
  """The model struct carries the batch sizes: (re)bind it to this data's."""
  s = m.struct.size
  if (s.nworld, s.nconmax, s.njmax) != (d.nworld, d.nconmax, d.njmax):
    s.nworld, s.nconmax, s.njmax = d.nworld, d.nconmax, d.njmax
    m.__dict__["nworld"] = d.nworld


def _stream(d: Data) -> int:
  return torch.cuda.current_stream(d.device).cuda_stream


def _static_geoms(m: Model, d: Data) -> None:
  """First use of a Data: pose the static geoms (world / terrain bodies) once."""
  if d._static_done:
    return
  keep, keep_s = m.struct.size.nstaticgeom, m.struct.size.nstaticsite
  m.struct.size.nstaticgeom = m.struct.size.nstaticsite = 0
  flags = m.struct.opt.flags
  m.struct.opt.flags = flags | _abi.OPT_WORLD_FRAME  # stored world poses must not depend on where the robots are right now
  native.check(native.lib().mjlab_forward_stages(ctypes.byref(m.struct), ctypes.byref(d.struct), native.STAGE_POSITION, _stream(d)), "mjlab_forward_stages")
  m.struct.opt.flags = flags
  m.struct.size.nstaticgeom, m.struct.size.nstaticsite = keep, keep_s
  d._static_done = True


def _sync_opt(m: Model) -> None:
  """`wp_model.opt.ls_parallel = cfg.ls_parallel` (reference sim/sim.py:111) is a plain attribute write: mirror it into the flags."""
  on = bool(getattr(m.opt, "ls_parallel", False))
  m.struct.opt.flags = (m.struct.opt.flags | _abi.OPT_LS_PARALLEL) if on else (m.struct.opt.flags & ~_abi.OPT_LS_PARALLEL)
  m.struct.opt.ls_parallel_min_step = float(getattr(m.opt, "ls_parallel_min_step", 1.0e-6))


def forward(m: Model, d: Data) -> None:
  with torch.cuda.device(d.device):
    _bind(m, d)
    _sync_opt(m)
    _static_geoms(m, d)
    native.check(native.lib().mjlab_forward(ctypes.byref(m.struct), ctypes.byref(d.struct), _stream(d)), "mjlab_forward")


def step(m: Model, d: Data) -> None:
  with torch.cuda.device(d.device):
    _bind(m, d)
    _sync_opt(m)
    _static_geoms(m, d)
    native.check(native.lib().mjlab_step(ctypes.byref(m.struct), ctypes.byref(d.struct), 1, _stream(d)), "mjlab_step")


def expand_model_fields(m: Model, nworld: int, fields_to_expand: list[str]) -> None:
  """Per-world copies of the listed float fields (reference sim/randomization.py:20-55)."""
  invalid = [f for f in fields_to_expand if not hasattr(m.host, f)]
  if invalid:
    raise ValueError(f"Fields not found in model: {invalid}")
  if nworld == 1:
    return
  moves_static = [f for f in fields_to_expand if f in ("geom_pos", "geom_quat", "body_pos", "body_quat")]
  if moves_static:
    if m.host.nterrain:
      raise NotImplementedError(f"per-world {moves_static} with a box terrain: terrain boxes are static and shared by all worlds")
    m.struct.size.nstaticgeom = 0
  if any(f in ("site_pos", "site_quat", "body_pos", "body_quat") for f in fields_to_expand):
    m.struct.size.nstaticsite = 0
  m.__dict__["nworld"] = nworld
  with torch.cuda.device(m.device):
    for name in fields_to_expand:
      if device_state.expand_field(m.struct, m._base, m._view, m.host, name, nworld, m.device, torch.cuda.current_stream(m.device).cuda_stream):
        m._expanded.add(name)
