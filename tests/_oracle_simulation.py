"""TEST INFRASTRUCTURE: a ``Simulation``-shaped object whose step() / forward() are the CPU ORACLE's (fp32 build), with
``sim.model`` / ``sim.data`` Bridges over host tensors that alias the oracle's own arrays.

It exists so that the whole reference environment stack (tools/reference_env.py: the reference's ManagerBasedRlEnv, Scene,
Entity, managers and task configs over this package's ``mujoco`` shim) can be exercised in the CPU test suite; the product
path is ``mjlab_amd.sim.Simulation`` on a GPU and never imports this file.  Same constructor signature and the same reference
surface as ``mjlab_amd.sim.Simulation`` (reference sim/sim.py:94-198).
"""

from __future__ import annotations

import numpy as np
import torch

from mjlab_amd import _abi, device_state
from mjlab_amd.nan_guard import NanGuard, NanGuardCfg
from mjlab_amd.sim_data import Bridge
from oracle.oracle import OracleSim


class OracleSimulation:
  def __init__(self, num_envs: int, cfg, model, device: str = "cpu") -> None:
    assert str(device) == "cpu"
    self.cfg, self.device, self.num_envs = cfg, "cpu", num_envs
    self._mj_model = model
    self.nconmax, self.njmax = _abi.default_capacities(model, getattr(cfg, "nconmax", None), getattr(cfg, "njmax", None))
    # the line search the configuration asks for (the reference's SimulationCfg: ls_parallel=True, sim/sim.py:89), like Simulation
    self.ls_parallel = bool(getattr(cfg, "ls_parallel", True))
    self.ora = OracleSim(model, num_envs, nconmax=self.nconmax, njmax=self.njmax, precision="f32", ls_parallel=self.ls_parallel)
    mfields = _abi.parse_layout(self.ora.lib.mjo_model_layout().decode())
    dfields = _abi.parse_layout(self.ora.lib.mjo_data_layout().decode())
    self._mfields = {f.name: f for f in mfields}
    mview: dict[str, torch.Tensor] = {}
    for f in mfields:
      t = torch.from_numpy(self.ora.mfield[f.name])
      mview[f.name] = t if f.kind == "i" else t.unsqueeze(0).expand(num_envs, *t.shape)
    for name in device_state.EXTRA_MODEL_FIELDS:
      t = torch.from_numpy(np.ascontiguousarray(getattr(model, name), dtype=np.float32)).unsqueeze(0)
      mview[name] = t.expand(num_envs, *t.shape[1:])
    dview: dict[str, torch.Tensor] = {}
    for f in dfields:
      n = _abi.count_of(f.count, model, self.nconmax, self.njmax)
      flat = torch.from_numpy(self.ora.dfield[f.name].reshape(num_envs, -1))
      dview[f.name] = device_state.shape_view(f, flat, n)
    dview["act"] = torch.zeros((num_envs, int(getattr(model, "na", 0))))
    scal = {k: int(getattr(model, k)) for k in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "nsensor", "nsensordata")}
    self._model_view, self._data = mview, dview
    self._model_bridge = Bridge("sim.model", mview, {**scal, "opt": model.opt, "nworld": num_envs})
    self._data_bridge = Bridge("sim.data", dview, {"nworld": num_envs, "njmax": self.njmax, "nconmax": self.nconmax})
    self.nan_guard = NanGuard(getattr(cfg, "nan_guard", NanGuardCfg()), num_envs, model)
    self.nthread = 8
    self.step_calls = self.forward_calls = 0
    self.forward()

  mj_model = property(lambda self: self._mj_model)
  host_model = property(lambda self: self._mj_model)
  data = property(lambda self: self._data_bridge)
  model = property(lambda self: self._model_bridge)

  def create_graph(self) -> None:
    pass

  def expand_model_fields(self, fields: list[str]) -> None:
    invalid = [f for f in fields if not hasattr(self._mj_model, f)]
    if invalid:
      raise ValueError(f"Fields not found in model: {invalid}")
    for name in fields:
      if name in self._mfields:
        if self._mfields[name].kind != "r":
          raise ValueError(f"Field '{name}' is an integer topology field and cannot be per-world")
        self._model_view[name] = torch.from_numpy(self.ora.expand_model_field(name))
      else:
        self._model_view[name] = self._model_view[name].clone().contiguous()

  def reset(self) -> None:
    pass

  def forward(self, env_mask=None) -> None:
    """``env_mask`` as in mjlab_amd.Simulation.forward (extension): only the marked worlds are recomputed (the others keep every
    array: the oracle forwards all worlds, the unmarked rows are put back)."""
    keep = None
    if env_mask is not None:
      m = torch.as_tensor(env_mask).bool()
      if not bool(m.any()):
        return
      if not bool(m.all()):
        rows = (~m).nonzero().flatten()
        keep = [(t, rows, t[rows].clone()) for t in self._data.values() if t.dim() >= 1 and t.shape[0] == m.numel()]
    self.forward_calls += 1
    self.ora.forward(nthread=self.nthread)
    if keep is not None:
      for t, rows, old in keep:
        t[rows] = old

  def step(self, nsubstep: int = 1) -> None:
    self.step_calls += nsubstep
    with self.nan_guard.watch(self.data):
      self.ora.step(nsubstep, nthread=self.nthread)

  def close(self) -> None:
    pass
