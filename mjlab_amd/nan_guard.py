"""NaN guard of ``Simulation.step`` (reference src/mjlab/utils/nan_guard.py, used at
src/mjlab/sim/sim.py:129,191: ``with self.nan_guard.watch(self.data): <step>``).

Disabled by default and then free.  When enabled it keeps the last ``buffer_size`` pre-step
physics states (mjSTATE_PHYSICS = qpos, qvel, act) of the first ``max_envs_to_capture``
environments and, the first time qpos / qvel / qacc / qacc_warmstart of any environment turns
NaN or Inf, writes them to ``nan_dump_<timestamp>.npz`` (same keys and metadata as the reference)
next to the model.  The model is saved as this package's ``.npz`` (``Model.load`` reads it back);
the reference writes an ``.mjb`` through ``mujoco.mj_saveModel``, which needs the mujoco wheel.
Checking costs a host sync per step -- it is a debugging aid, like upstream.
"""

from __future__ import annotations

from collections import deque
from contextlib import contextmanager
from dataclasses import dataclass
from datetime import datetime
from pathlib import Path
from typing import Any, Iterator

import numpy as np
import torch

from .mjcf import Model


@dataclass
class NanGuardCfg:
  enabled: bool = False
  buffer_size: int = 100
  output_dir: str = "/tmp/mjlab/nan_dumps"
  max_envs_to_capture: int = 5  # Max number of NaN envs to save.


class NanGuard:
  def __init__(self, cfg: NanGuardCfg, num_envs: int, model: Model) -> None:
    self.enabled = cfg.enabled
    self.num_envs = num_envs
    if not self.enabled:
      return
    self.buffer_size = cfg.buffer_size
    self.output_dir = Path(cfg.output_dir)
    self.max_envs_to_capture = cfg.max_envs_to_capture
    self.num_to_capture = min(num_envs, cfg.max_envs_to_capture)
    self.buffer: deque = deque(maxlen=self.buffer_size)
    self.step_counter = 0
    self._dumped = False  # one dump per run
    if self.num_to_capture < num_envs:
      print(f"[NanGuard] Capturing only {self.num_to_capture}/{num_envs} envs (limited by nan_guard_max_envs={self.max_envs_to_capture})")
    self.model = model
    self.state_size = int(model.nq + model.nv + getattr(model, "na", 0))  # mj_stateSize(mjSTATE_PHYSICS)

  def capture(self, data: Any) -> None:
    """Pre-step state of the first ``num_to_capture`` environments: [qpos, qvel, act] per row."""
    if not self.enabled:
      return
    n = self.num_to_capture
    parts = [data.qpos[:n], data.qvel[:n]]
    if getattr(self.model, "na", 0) > 0:
      parts.append(data.act[:n])
    states = torch.cat(parts, dim=1).to(torch.float64).cpu().numpy()
    self.buffer.append({"step": self.step_counter, "states": states})
    self.step_counter += 1

  @contextmanager
  def watch(self, data: Any) -> Iterator[None]:
    self.capture(data)
    yield
    self.check_and_dump(data)

  def check_and_dump(self, data: Any) -> bool:
    if not self.enabled or self._dumped:
      return False
    bad = torch.zeros(data.qpos.shape[0], dtype=torch.bool, device=data.qpos.device)
    for t in (data.qpos, data.qvel, data.qacc, data.qacc_warmstart):
      bad |= ~torch.isfinite(t).all(dim=-1)
    if not bool(bad.any()):
      return False
    self._dump_buffer(torch.where(bad)[0].cpu().numpy().tolist())
    self._dumped = True
    return True

  def _dump_buffer(self, nan_env_ids: list[int]) -> None:
    self.output_dir.mkdir(parents=True, exist_ok=True)
    timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    filename = self.output_dir / f"nan_dump_{timestamp}.npz"
    model_filename = self.output_dir / f"model_{timestamp}.npz"
    self.model.save(model_filename)
    out: dict[str, Any] = {f"states_step_{item['step']:06d}": item["states"] for item in self.buffer}
    out["_metadata"] = np.array(
      {
        "num_envs_total": self.num_envs,
        "num_envs_captured": self.num_to_capture,
        "nan_env_ids": nan_env_ids[: self.max_envs_to_capture],
        "state_size": self.state_size,
        "buffer_size": len(self.buffer),
        "detection_step": self.step_counter,
        "timestamp": timestamp,
        "model_file": model_filename.name,
        "note": "Rows are [qpos, qvel, act] (mjSTATE_PHYSICS order) of the captured envs before each step; "
        "the model is saved as mjlab_amd Model .npz (mjlab_amd.mjcf.Model.load).",
      },
      dtype=object,
    )
    np.savez_compressed(filename, **out)
    print(f"[NanGuard] Detected NaN/Inf at step {self.step_counter}")
    print(f"[NanGuard] NaN/Inf found in envs: {nan_env_ids[:10]}...")
    print(f"[NanGuard] Dumped {len(self.buffer)} states to: {filename}")
    print(f"[NanGuard] Saved model to: {model_filename}")
