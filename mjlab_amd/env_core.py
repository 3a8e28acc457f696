"""The functional core of ``GraphedRlEnv`` (mjlab_amd/graphed_env.py): the pieces of the reference's control step that the wrapper
RESTATES -- mask based, capturable -- as functions of plain tensors and small specs, with no reference object in sight:

  * the bookkeeping of ``ManagerBasedRlEnv._reset_idx`` (reference envs/manager_based_rl_env.py:214-249 and the managers' ``reset()``:
    managers/reward_manager.py:60-74, command_manager.py:44-53,128-134, termination_manager.py:73-85, action_manager.py:101-110,
    event_manager.py:139-148): masked sums for the log, masked fills of the buffers -- ``ResetBookkeeping``;
  * ``extras["log"]`` as the reference leaves it between steps with resets, and its sharded form -- ``LogBook``;
  * ``RewardManager.compute``'s accumulation (managers/reward_manager.py:77-89) -- ``reward_accumulate``;
  * ``ObservationManager.compute`` for groups of plain terms (managers/observation_manager.py:144-188) -- ``assemble_observation``;
  * ``UniformVelocityCommand._update_command`` (tasks/velocity/mdp/velocity_command.py:88-102) -- ``update_uniform_velocity``.

The wrapper extracts the tensors from the reference's environment object and calls these; ``tools/make_graphed_golden.py`` records
the reference's OWN eager results for the same inputs (tests/golden/graphed_core_*.npz) and ``tests/test_graphed_core_golden.py``
replays them through this module -- on the CPU and, without the reference tree, on the MI355X, where the fills / sums / accumulation
also run as their HIP launches (mjlab_amd/env_terms.py) against the same recorded truth.
"""

from __future__ import annotations

from typing import Any

import torch


# ------------------------------------------------------------------------------------------------- masked fills and sums
class TorchMaskedFill:
  """``buf[mask] = value`` for a list of (tensor, value) pairs (rows of any trailing shape); the torch twin of env_terms.MaskedFill."""

  def __init__(self, items: list) -> None:
    self.items = list(items)

  def __call__(self, mask: torch.Tensor) -> None:
    for t, value in self.items:
      m = mask.reshape((-1,) + (1,) * (t.dim() - 1))
      if isinstance(value, torch.Tensor):  # (a scalar tensor read at every call: the step's value under a capture)
        torch.where(m, value.to(t.dtype), t, out=t)
      else:
        t.masked_fill_(m, bool(value) if t.dtype == torch.bool else value)


class TorchMaskedSums:
  """``out[i] = vectors[i][mask].sum()`` (bools counted), ``out[-1] = mask.sum()``: ONE stacked reduction; the torch twin of
  env_terms.MaskedSums (same output layout, float32)."""

  def __init__(self, vectors: list) -> None:
    self.vectors = list(vectors)

  def __call__(self, mask: torch.Tensor) -> torch.Tensor:
    cols = [v.to(torch.float32) for v in self.vectors] + [torch.ones_like(mask, dtype=torch.float32)]
    return (torch.stack(cols, dim=1) * mask[:, None]).sum(dim=0)


class ResetBookkeeping:
  """The masked sums the managers' ``reset()`` log and the masked fills they perform, for one environment.

  `fills`: (tensor, value) pairs; `vectors`: the summed per-environment vectors in the order [episode reward sums | command
  metrics updated in place | termination flags]; `rkeys` / `mkeys` / `tkeys` name them.  ``fused``: the two HIP launches of
  mjlab_amd/env_terms.py (device tensors) instead of the torch chains."""

  def __init__(self, fills: list, vectors: list, rkeys: list, mkeys: list, tkeys: list, fused: bool) -> None:
    self.rkeys, self.mkeys, self.tkeys = list(rkeys), list(mkeys), list(tkeys)
    assert len(vectors) == len(rkeys) + len(mkeys) + len(tkeys)
    if fused:
      from . import env_terms

      self.fill, self.sums = env_terms.MaskedFill(fills), env_terms.MaskedSums(vectors)
    else:
      self.fill, self.sums = TorchMaskedFill(fills), TorchMaskedSums(vectors)

  def log_entries(self, out: torch.Tensor) -> dict:
    """key -> (raw masked sum, kind) in the reference's key names; ``out`` = ``self.sums(mask)``."""
    log: dict = {}
    kr, km = len(self.rkeys), len(self.mkeys)
    for k, key in enumerate(self.rkeys):
      log["Episode_Reward/" + key] = (out[k], "sum_len")  # / resets / max_episode_length_s
    for k, (name, key) in enumerate(self.mkeys):
      log[f"Metrics/{name}/{key}"] = (out[kr + k], "sum")  # / resets
    for k, key in enumerate(self.tkeys):
      log["Episode_Termination/" + key] = (out[kr + km + k], "count")
    return log


# ------------------------------------------------------------------------------------------------------------ extras["log"]
class LogBook:
  """``extras["log"]`` as the reference leaves it.  ``publish(log, mask)`` takes key -> (value, kind) with the RAW masked sums of a
  step: kind "sum_len" (episode reward sums: / resets / max_episode_length_s, managers/reward_manager.py:65-70), "sum" (command
  metrics: / resets, managers/command_manager.py:128-134), "count" (terminations per term), "state" (curriculum state).

  * ``_reset_idx`` -- and with it the managers' reset() logging -- runs only in a step in which some environment reset
    (envs/manager_based_rl_env.py:121-127), so between two such steps the dict keeps the numbers of the last one.  Here the masked
    sums are evaluated every step (a capture cannot skip them); the scalars go through ONE ``where(any reset, new, previous)`` into
    a persistent vector, and ``self.pub`` is a persistent dict of 0-dim views of it (the same objects across replays and resets;
    counts are float32 like everything else in the vector).
  * SHARDED (``world > 1``): the raw sums and the reset count of this rank are parked in ``self.raw`` and the caller
    all-reduces them (sum) before ``finish()`` divides, so every rank logs the numbers of the GLOBAL batch: mean over all ranks'
    reset environments, total termination counts, mean curriculum state."""

  def __init__(self, max_episode_length_s: float, device: Any, world: int = 1) -> None:
    self.max_len, self.device, self.world = float(max_episode_length_s), device, int(world)
    self.keys: list | None = None
    self.vec = self.raw = self.div = self.scale = None
    self.pub: dict = {}
    self.first = False
    self._ptrs: list | None = None  # source addresses the launch form's pointer table was built for (_publish_fused)

  def publish(self, log: dict, mask: torch.Tensor, count: torch.Tensor | None = None) -> dict:
    """``count``: the number of environments in `mask` as a device float scalar, where the caller has it (the masked sums' last output)."""
    keys = [k for k, (v, _) in log.items() if v.dim() == 0]
    if self._publish_fused(log, keys, count):
      for k, (v, _) in log.items():
        if v.dim() != 0:
          self.pub[k] = v
      return self.pub
    raw = torch.stack([log[k][0].to(torch.float32) for k in keys] + [(mask.sum() if count is None else count).to(torch.float32)])
    first = False
    if self.vec is None or self.keys != keys:
      kinds = [log[k][1] for k in keys]
      self.keys = keys
      self.div = torch.tensor([kd in ("sum_len", "sum") for kd in kinds], device=self.device)
      self.scale = torch.tensor([1.0 / self.max_len if kd == "sum_len" else (1.0 / self.world if kd == "state" else 1.0) for kd in kinds], device=self.device)
      self.vec = torch.zeros(len(keys), device=self.device)
      self.raw = torch.zeros(len(keys) + 1, device=self.device)
      self.pub = {k: self.vec[i] for i, k in enumerate(keys)}
      self._ptrs = None  # (the launch form's tables follow the new vectors at its next call)
      first = True
    if self.world > 1:
      self.raw.copy_(raw)  # finished by the caller after the all-reduce: finish(self.raw, self.first)
      self.first = first or self.first
    else:
      self.finish(raw, first)
    for k, (v, _) in log.items():  # (non-scalar curriculum state: passed through as it is)
      if v.dim() != 0:
        self.pub[k] = v
    return self.pub

  def _publish_fused(self, log: dict, keys: list, count: torch.Tensor | None) -> bool:
    """One HIP launch (``mjlab_log_finish``) straight from the scalars the log names -- no stack, no division / where chain -- when this
    is a single-process log on the GPU whose values are float32 device scalars at fixed addresses (views into the masked sums' output
    and persistent state) and the count is at hand; the vectors exist already (the first publication goes through the torch lines)."""
    if self.world > 1 or count is None or self.vec is None or self.keys != keys or not self.vec.is_cuda:
      return False
    vals = [log[k][0] for k in keys]
    if not all(v.is_cuda and v.dtype == torch.float32 for v in vals) or count.dtype != torch.float32 or not count.is_cuda:
      return False
    ptrs = [v.data_ptr() for v in vals]
    if self._ptrs != ptrs:  # (addresses are part of what a captured launch records: a new table for new addresses)
      if torch.cuda.is_current_stream_capturing():
        return False
      self._ptrs = ptrs
      self._ptr_table = torch.tensor(ptrs, dtype=torch.int64, device=self.device)
      self._div_u8 = self.div.to(torch.uint8).contiguous()
    from . import native

    native.check(native.lib().mjlab_log_finish(self._ptr_table.data_ptr(), self._div_u8.data_ptr(), self.scale.data_ptr(), len(keys), count.data_ptr(), 0,
                                               self.vec.data_ptr(), torch.cuda.current_stream(self.vec.device).cuda_stream), "mjlab_log_finish")
    self.first = False
    return True

  def clear(self) -> None:
    """Zero what has been published so far, keeping the vectors (and the tensors handed out) in place."""
    if self.vec is not None:
      self.vec.zero_()
      self.raw.zero_()
    self.first = False

  def finish(self, raw: torch.Tensor, first: bool) -> None:
    cnt = raw[-1]
    new = torch.where(self.div, raw[:-1] / cnt.clamp(min=1.0), raw[:-1]) * self.scale
    if first:
      self.vec.copy_(new)
    else:
      torch.where(cnt > 0, new, self.vec, out=self.vec)
    self.first = False


# ------------------------------------------------------------------------------------------------------------------ rewards
def reward_accumulate(values: torch.Tensor, weights: torch.Tensor, columns: list, dt: float, reward_buf: torch.Tensor, episode_sums: list,
                      step_reward: torch.Tensor) -> torch.Tensor:
  """RewardManager.compute's loop (managers/reward_manager.py:77-89) over the ACTIVE terms (weight != 0), from their raw outputs
  `values` (k, n): ``value = raw * weight * dt``; ``reward_buf += value``; ``episode_sums[name] += value``;
  ``step_reward[:, i] = value / dt`` -- the same operations in the same order, so the results are the reference's bit for bit.  (The
  torch twin of the mjlab_reward_accumulate launch; idle terms' step_reward columns are zeroed by the caller.)"""
  reward_buf[:] = 0.0
  for k, col in enumerate(columns):
    value = values[k] * weights[k] * dt
    reward_buf += value
    episode_sums[k] += value
    step_reward[:, col] = value / dt
  return reward_buf


# ------------------------------------------------------------------------------------------------------------- observations
def assemble_observation(raw_terms: list, noisy: bool, width: torch.Tensor | None, lo: torch.Tensor | None, U: torch.Tensor | None) -> torch.Tensor:
  """An observation group of plain terms (2-D outputs concatenated along the last dimension, no clip / scale / history; noise none
  or ``UniformNoiseCfg(operation="add")`` with scalar bounds): ONE concatenation of the raw term outputs plus, if the group is
  corrupted, ONE noise block ``U * (n_max - n_min) + n_min`` with per-column bounds (the reference: clone + rand_like + mul + add + add
  per term, then the concatenation).  Without noise the values are the reference's bit for bit."""
  raw = torch.cat(raw_terms, dim=-1)
  return raw + (U * width + lo) if noisy else raw


# ----------------------------------------------------------------------------------------------------------------- commands
def update_uniform_velocity(vel_command_b: torch.Tensor, heading_target: torch.Tensor | None, heading_w: torch.Tensor | None, is_heading_env: torch.Tensor | None,
                            is_standing_env: torch.Tensor, heading_command: bool, stiffness: float, ang_vel_z: Any, wrap_to_pi: Any) -> None:
  """UniformVelocityCommand._update_command (tasks/velocity/mdp/velocity_command.py:88-102), in place on `vel_command_b`: the yaw rate of
  the heading-controlled environments from the heading error (clipped to the ang_vel_z range), zero command for the standing ones."""
  v = vel_command_b
  if heading_command:
    err = wrap_to_pi(heading_target - heading_w)
    yaw = torch.clip(stiffness * err, min=ang_vel_z[0], max=ang_vel_z[1])
    v[:, 2] = torch.where(is_heading_env, yaw, v[:, 2])
  v.masked_fill_(is_standing_env[:, None], 0.0)
