"""Opt-in NaN guard around ``Simulation.step`` (reference surface: ``utils/nan_guard.py:18-58``
``NanGuardCfg`` / ``NanGuard``, used at ``sim/sim.py:129,191`` as ``with self.nan_guard.watch(self.data):``
and configured by ``scripts/train.py:56-58`` through ``cfg.env.sim.nan_guard.enabled / .output_dir``).

Disabled (the default) every call returns immediately.  Enabled, the history lives ON THE DEVICE: a ring
of ``buffer_size`` slots x ``max_envs_to_capture`` worlds x ``nq + nv + na`` floats is filled by two strided
device copies per step (no host transfer, unlike the reference's per-env ``.cpu()`` loop), and the
non-finite test is one fused reduction into a device flag.  Only that flag crosses to the host -- every
``check_every`` steps (1 = the reference's behaviour: a sync per step, which is why this is a debugging aid).
On the first detection the ring is copied out once and written in the reference's dump layout
(``nan_dump_<timestamp>.npz``: ``states_step_%06d`` arrays in mjSTATE_PHYSICS order + ``_metadata``), so the
reference's ``scripts/nan_viz.py`` reads it; the model goes next to it as this package's ``Model`` ``.npz``
(``mujoco.mj_saveModel`` needs the wheel).
"""

from __future__ import annotations

from contextlib import contextmanager
from dataclasses import dataclass
from datetime import datetime
from pathlib import Path
from typing import Any, Iterator

import numpy as np
import torch


@dataclass
class NanGuardCfg:
  """Field names and defaults of the reference's ``NanGuardCfg`` (utils/nan_guard.py:18-25)."""

  enabled: bool = False
  buffer_size: int = 100
  output_dir: str = "/tmp/mjlab/nan_dumps"
  max_envs_to_capture: int = 5
  check_every: int = 1  # extension: read the device flag back every n-th step only


class NanGuard:
  WATCHED = ("qpos", "qvel", "qacc", "qacc_warmstart")  # utils/nan_guard.py:104

  def __init__(self, cfg: NanGuardCfg, num_envs: int, model: Any) -> None:
    self.enabled = bool(cfg.enabled)
    self.num_envs = num_envs
    if not self.enabled:
      return
    self.buffer_size = int(cfg.buffer_size)
    self.output_dir = Path(cfg.output_dir)
    self.max_envs_to_capture = int(cfg.max_envs_to_capture)
    self.num_to_capture = min(num_envs, self.max_envs_to_capture)
    self.check_every = max(1, int(getattr(cfg, "check_every", 1)))
    self.model = model
    self._nq, self._nv, self._na = int(model.nq), int(model.nv), int(getattr(model, "na", 0))
    self.state_size = self._nq + self._nv + self._na  # mj_stateSize(mjSTATE_PHYSICS)
    self.step_counter = 0
    self._ring: torch.Tensor | None = None  # allocated on the device of the first watched data
    self._slot_step = [-1] * self.buffer_size
    self._bad: torch.Tensor | None = None  # per-world "has been non-finite since the last read-back"
    self._dumped = False
    if self.num_to_capture < num_envs:
      print(f"[NanGuard] keeping {self.num_to_capture} of {num_envs} envs (max_envs_to_capture)")

  # -- per step ----------------------------------------------------------------------------------
  def capture(self, data: Any) -> None:
    """Pre-step [qpos, qvel, act] of the captured worlds into the next ring slot (device to device)."""
    if not self.enabled:
      return
    n, nq, nv = self.num_to_capture, self._nq, self._nv
    if self._ring is None:
      self._ring = torch.zeros(self.buffer_size, n, self.state_size, dtype=data.qpos.dtype, device=data.qpos.device)
      self._bad = torch.zeros(data.qpos.shape[0], dtype=torch.bool, device=data.qpos.device)
    slot = self._ring[self.step_counter % self.buffer_size]
    slot[:, :nq].copy_(data.qpos[:n])
    slot[:, nq:nq + nv].copy_(data.qvel[:n])
    if self._na:
      slot[:, nq + nv:].copy_(data.act[:n])
    self._slot_step[self.step_counter % self.buffer_size] = self.step_counter
    self.step_counter += 1

  @contextmanager
  def watch(self, data: Any) -> Iterator[None]:
    self.capture(data)
    yield
    self.check_and_dump(data)

  def check_and_dump(self, data: Any) -> bool:
    """True when a dump was written by this call."""
    if not self.enabled or self._dumped or self._bad is None:
      return False
    for name in self.WATCHED:
      self._bad |= ~torch.isfinite(getattr(data, name)).all(dim=-1)
    if self.step_counter % self.check_every:
      return False
    if not bool(self._bad.any()):  # the one host sync
      return False
    self._dump(torch.nonzero(self._bad).view(-1).cpu().numpy().tolist())
    self._dumped = True
    return True

  # -- dump --------------------------------------------------------------------------------------
  def _dump(self, nan_env_ids: list[int]) -> None:
    assert self._ring is not None
    self.output_dir.mkdir(parents=True, exist_ok=True)
    stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    path = self.output_dir / f"nan_dump_{stamp}.npz"
    model_path = self.output_dir / f"model_{stamp}.npz"
    if hasattr(self.model, "save"):
      self.model.save(model_path)
    ring = self._ring.to(torch.float64).cpu().numpy()
    order = sorted((s, i) for i, s in enumerate(self._slot_step) if s >= 0)
    out: dict[str, Any] = {f"states_step_{s:06d}": ring[i] for s, i in order}
    out["_metadata"] = np.array(
      {
        "num_envs_total": self.num_envs,
        "num_envs_captured": self.num_to_capture,
        "nan_env_ids": nan_env_ids[: self.max_envs_to_capture],
        "state_size": self.state_size,
        "buffer_size": len(order),
        "detection_step": self.step_counter,
        "timestamp": stamp,
        "model_file": model_path.name,
        "note": "rows = [qpos, qvel, act] (mjSTATE_PHYSICS) of the captured envs before each step; model = mjlab_amd.mjcf.Model npz",
      },
      dtype=object,
    )
    np.savez_compressed(path, **out)
    print(f"[NanGuard] non-finite state at step {self.step_counter} in envs {nan_env_ids[:10]}; {len(order)} states -> {path}")
