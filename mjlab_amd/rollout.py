"""Minimal control-step driver around ``Simulation`` for benchmarks and tests.

Restates the physics-facing part of ``ManagerBasedRlEnv.step`` (reference
src/mjlab/envs/manager_based_rl_env.py:106-147): process action -> ``decimation`` x
[write ctrl, ``sim.step()``] -> termination check -> masked reset of terminated envs ->
``sim.forward()``.  The MDP managers (rewards, observations, commands) are outside the
physics hot path and are not reproduced; resets are mask-based (no ``nonzero()`` host sync).
"""

from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from . import native
from .mjcf import JNT_FREE, Model
from .sim import Simulation


def _entity_view_struct():
  from .entity_data import _View

  return _View


class _PushRange(ctypes.Structure):
  _fields_ = [("lo", ctypes.c_float * 6), ("hi", ctypes.c_float * 6)]


class _Control(ctypes.Structure):
  """mjlab_control_t (include/mjlab_amd.h)."""

  _fields_ = [
    ("nsubstep", ctypes.c_int), ("forward_mode", ctypes.c_int), ("max_len", ctypes.c_int), ("pad_", ctypes.c_int),
    ("action", ctypes.c_void_p), ("action_offset", ctypes.c_void_p), ("action_scale", ctypes.c_void_p),
    ("key_qpos", ctypes.c_void_p), ("rnd3", ctypes.c_void_p), ("episode_length", ctypes.c_void_p), ("reset_mask", ctypes.c_void_p),
    ("env_origins", ctypes.c_void_p), ("world_order", ctypes.c_void_p), ("push_time_left", ctypes.c_void_p), ("rnd7", ctypes.c_void_p),
    ("min_height", ctypes.c_float), ("min_up_z", ctypes.c_float), ("push_dt", ctypes.c_float),
    ("push_interval_lo", ctypes.c_float), ("push_interval_hi", ctypes.c_float), ("push_range", _PushRange),
    ("readback_on", ctypes.c_int), ("pad2_", ctypes.c_int), ("readback", _entity_view_struct()),
  ]  # fmt: skip


# The velocity task's events that touch the physics state or model (reference
# src/mjlab/tasks/velocity/velocity_env_cfg.py:136-172,219-223 and config/{g1,go1}/flat_env_cfg.py):
# foot friction U(0.3, 1.2) per env and foot geom at startup, a root-velocity kick every U(1, 3) s,
# termination on a 70 degree tilt.  (Go1 keeps the base class's +-1.0 m/s kick; G1-flat sets +-0.5.)
VELOCITY_TASK_EVENTS = {
  "g1": {"friction_range": (0.3, 1.2), "friction_geoms": r"(left|right)_foot[1-7]_collision$",
         "push": {"interval_s": (1.0, 3.0), "velocity": {"x": (-0.5, 0.5), "y": (-0.5, 0.5)}}, "bad_orientation_deg": 70.0},
  "go1": {"friction_range": (0.3, 1.2), "friction_geoms": r"[FR][LR]_foot_collision$",
          "push": {"interval_s": (1.0, 3.0), "velocity": {"x": (-1.0, 1.0), "y": (-1.0, 1.0)}}, "bad_orientation_deg": 70.0},
}  # fmt: skip


class PhysicsRollout:
  def __init__(self, sim: Simulation, action_scale: np.ndarray | float = 0.25, decimation: int = 4,
               episode_length_s: float = 20.0, min_height: float = 0.3, seed: int = 42, key: int = 0,
               masked_forward: bool = False, fused_reset: bool = True, min_up_z: float | None = None,
               max_init_terrain_level: int | None = 5, friction_range: tuple[float, float] | None = None,
               friction_geoms: str | None = None, push: dict | None = None, bad_orientation_deg: float | None = None,
               substeps_per_call: int = 1, control_kernel: bool = False) -> None:
    m: Model = sim.host_model
    dev = sim.data.qpos.device
    self.sim, self.m, self.decimation = sim, m, decimation
    # 1 = the reference's call pattern (write ctrl, sim.step(), `decimation` times); `decimation` = one
    # sim.step(nsubstep=decimation) call: the action is fixed during a control step, so the results are the same
    assert decimation % substeps_per_call == 0
    self.substeps_per_call = substeps_per_call
    # True: the whole control step (action -> ctrl, substeps, termination + reset, forward, push) is ONE
    # library launch (mjlab_control_step); bit-identical to the call sequence below
    self.control_kernel = control_kernel
    self.gen = torch.Generator(device=dev)
    self.gen.manual_seed(seed)
    self.key_qpos = torch.tensor(m.key_qpos[key] if m.nkey else m.qpos0, dtype=torch.float32, device=dev)
    jn = m.actuator_trnid[:, 0]
    self.act_qadr = torch.tensor(m.jnt_qposadr[jn], dtype=torch.long, device=dev)
    self.default_joint = self.key_qpos[self.act_qadr]
    scale = np.broadcast_to(np.asarray(action_scale, dtype=np.float32), (m.nu,)).copy()
    self.action_scale = torch.tensor(scale, device=dev)
    self.has_free = m.njnt > 0 and m.jnt_type[0] == JNT_FREE
    self.max_len = int(round(episode_length_s / (m.opt.timestep * decimation)))
    self.min_height = min_height
    # False = the reference's behaviour (forward on ALL worlds after a reset,
    # envs/manager_based_rl_env.py:128-132); True = only the reset worlds (extension)
    self.masked_forward = masked_forward
    # True: termination test + reset in one library launch (mjlab_masked_reset); False: the same
    # logic as a chain of torch ops (the reference's style)
    self.fused_reset = fused_reset
    n = sim.num_envs
    # Environment origins.  On a generated terrain every env spawns on its sub-terrain's origin
    # (reference terrains/terrain_importer.py:196-229, max_init_terrain_level = 5 in
    # tasks/velocity/velocity_env_cfg.py:35) and the termination is the reference's orientation
    # test (bad_orientation, limit 70 degrees: velocity_env_cfg.py TerminationCfg.fell_over)
    # instead of an absolute height; on the plane all envs share the origin.
    self.env_origins: torch.Tensor | None = None
    origins = getattr(m, "terrain_origins", None)
    if origins is not None and self.has_free:
      from . import terrains

      eo, self.terrain_levels, self.terrain_types = terrains.env_origins_curriculum(
        n, np.asarray(origins), max_init_terrain_level, np.random.default_rng(seed)
      )
      self.env_origins = torch.tensor(eo, dtype=torch.float32, device=dev)
      if min_up_z is None:
        min_up_z, self.min_height = math.cos(math.radians(70.0)), -1.0e9
    if bad_orientation_deg is not None and min_up_z is None:
      # the reference's only state-based termination of the velocity task (bad_orientation, mdp/terminations.py:23-31)
      min_up_z, self.min_height = math.cos(math.radians(bad_orientation_deg)), -1.0e9
    self.min_up_z = -2.0 if min_up_z is None else float(min_up_z)
    # startup event: per-env, per-geom friction (randomize_field "abs" on axis 0 of geom_friction)
    self.friction_geom_ids: np.ndarray | None = None
    if friction_range is not None:
      import re

      pat = re.compile(friction_geoms or ".*")
      gids = np.asarray([g for g, name in enumerate(m.names["geom"]) if name and pat.search(name)], dtype=np.int64)
      if gids.size == 0:
        raise ValueError(f"no geom name matches {friction_geoms!r}")
      sim.expand_model_fields(["geom_friction"])
      lo, hi = friction_range
      vals = lo + (hi - lo) * torch.rand((n, gids.size), device=dev, generator=self.gen)
      sim.model.geom_friction[:, torch.from_numpy(gids).to(dev), 0] = vals
      sim.create_graph()  # pointers changed (the reference re-captures too: manager_based_rl_env.py:102-104)
      self.friction_geom_ids = gids
    # interval event: velocity kicks under a per-env timer (mjlab_interval_push)
    self.push = None
    if push is not None and self.has_free:
      lo_t, hi_t = push["interval_s"]
      rng6 = _PushRange()
      for k, key_ in enumerate(("x", "y", "z", "roll", "pitch", "yaw")):
        rng6.lo[k], rng6.hi[k] = push["velocity"].get(key_, (0.0, 0.0))
      time_left = lo_t + (hi_t - lo_t) * torch.rand((n,), device=dev, generator=self.gen)
      self.push = (float(lo_t), float(hi_t), rng6, time_left)
    self._graph: torch.cuda.CUDAGraph | None = None
    self._action_buf: torch.Tensor | None = None
    self._obs_buf: torch.Tensor | None = None
    # load balance of the control kernel (mjlab_control_t.world_order): see balance_worlds()
    self.readback = None  # an mjlab_amd.entity_data.EntityReadback to be refreshed by the control kernel
    self.world_order: torch.Tensor | None = None
    self._slot_of_rank: torch.Tensor | None = None
    # start at random episode phase like the reference (train.py:109-111 init_at_random_ep_len)
    self.episode_length = torch.randint(0, self.max_len, (n,), device=dev, generator=self.gen).to(torch.int32)
    self._reset_mask = torch.zeros((n,), dtype=torch.int32, device=dev)
    self.reset_all()

  def _sample_reset_qpos(self, n: int) -> torch.Tensor:
    """Keyframe pose with x, y in U(-0.5, 0.5) and yaw in U(-3.14, 3.14)
    (reference src/mjlab/tasks/velocity/velocity_env_cfg.py:136-144)."""
    return self._reset_qpos_from(torch.rand((n, 3), device=self.key_qpos.device, generator=self.gen))

  def _reset_qpos_from(self, rnd3: torch.Tensor) -> torch.Tensor:
    """The reset pose as mjlab_masked_reset computes it from 3 uniforms per world."""
    n = rnd3.shape[0]
    q = self.key_qpos.unsqueeze(0).repeat(n, 1)
    if self.has_free:
      xy = rnd3[:, 0:2] - 0.5
      yaw = (rnd3[:, 2] * 2 - 1) * 3.14
      q[:, 0:2] += xy
      if self.env_origins is not None:
        q[:, 0:3] += self.env_origins
      q[:, 3] = torch.cos(yaw * 0.5)
      q[:, 4:6] = 0.0
      q[:, 6] = torch.sin(yaw * 0.5)
    return q

  def reset_all(self) -> None:
    d = self.sim.data
    n = self.sim.num_envs
    d.qpos[:] = self._sample_reset_qpos(n)
    d.qvel[:] = 0.0
    d.ctrl[:] = self.default_joint
    d.qacc_warmstart[:] = 0.0
    self.sim.forward()

  def step(self, action: torch.Tensor) -> torch.Tensor:
    """One control step; returns the reset mask of shape ``(num_envs,)``: int32 (non-zero = reset) with the fused
    reset -- the array the library wrote, no conversion launch -- and bool with ``fused_reset=False``.

    With ``capture_graph()`` done, the whole control step -- action processing, the
    ``decimation`` physics steps, termination test, masked reset and the forward pass -- is
    ONE hipGraph replay (resets are mask based, so there is no
    ``nonzero()`` host sync and nothing data dependent on the host side)."""
    self._nstep = getattr(self, "_nstep", 0) + 1
    if self._nstep % 16 == 0:  # wave-priority classes follow the batch (a few small device ops, outside the graph)
      self.sim.update_priority_thresholds()
    if self._graph is not None:
      if action is not self._action_buf:  # random_action(out=...) / a policy may write the static buffer directly
        self._action_buf.copy_(action)
      self._graph.replay()
      return self._reset_buf
    return self._step_eager(action)

  def capture_graph(self) -> None:
    """Capture one control step into a hipGraph (the physics kernels are launched directly
    inside the capture, not through the Simulation's own step/forward graphs)."""
    n = self.sim.num_envs
    dev = self.key_qpos.device
    self._action_buf = torch.zeros((n, self.m.nu), device=dev)
    # with the fused reset the mask written by the library IS the result (int32, non-zero = reset): no conversion launch
    self._reset_buf = self._reset_mask if self.fused_reset else torch.zeros((n,), dtype=torch.bool, device=dev)
    sim_graph = self.sim.use_graph
    self.sim.use_graph = False  # nested replays cannot be captured; record the raw launches
    try:
      # warm-up on a side stream (allocator, lazy init), then capture
      st = torch.cuda.Stream(device=dev)
      st.wait_stream(torch.cuda.current_stream(dev))
      with torch.cuda.stream(st):
        self._capture_body()
      torch.cuda.current_stream(dev).wait_stream(st)
      g = torch.cuda.CUDAGraph()
      g.register_generator_state(self.gen)
      with torch.cuda.graph(g):
        self._capture_body()
      self._graph = g
    finally:
      self.sim.use_graph = sim_graph

  def _capture_body(self) -> None:
    r = self._step_eager(self._action_buf)
    if r is not self._reset_buf:
      self._reset_buf.copy_(r)

  def _step_eager(self, action: torch.Tensor) -> torch.Tensor:
    """-> the reset mask of this control step: int32 (non-zero = reset) with the fused reset, bool otherwise."""
    d = self.sim.data
    s = self.sim
    n = s.num_envs
    dev = self.key_qpos.device
    # one draw per control step for everything below: 3 uniforms per world for the reset pose, 7 for the push
    rnd = torch.rand((n * 10,), device=dev, generator=self.gen)
    rnd3, rnd7 = rnd[: 3 * n].view(n, 3), rnd[3 * n :].view(n, 7)
    dt = float(self.m.opt.timestep * self.decimation)
    if self.control_kernel and self.fused_reset:
      c = _Control()
      c.nsubstep, c.forward_mode, c.max_len = self.decimation, 2 if self.masked_forward else 1, self.max_len
      self._action_in = action.contiguous()  # kept alive until the launch has been enqueued / captured
      c.action, c.action_offset, c.action_scale = self._action_in.data_ptr(), self.default_joint.data_ptr(), self.action_scale.data_ptr()
      c.key_qpos, c.rnd3 = self.key_qpos.data_ptr(), rnd3.data_ptr()
      c.episode_length, c.reset_mask = self.episode_length.data_ptr(), self._reset_mask.data_ptr()
      c.env_origins = 0 if self.env_origins is None else self.env_origins.data_ptr()
      c.min_height, c.min_up_z = float(self.min_height), self.min_up_z
      c.world_order = 0 if self.world_order is None else self.world_order.data_ptr()
      if self.readback is not None:  # EntityReadback whose outputs this launch refreshes
        c.readback_on = 1
        ctypes.memmove(ctypes.byref(c.readback), ctypes.byref(self.readback._view), ctypes.sizeof(c.readback))
      if self.push is not None:
        lo_t, hi_t, rng6, time_left = self.push
        c.push_time_left, c.rnd7, c.push_dt, c.push_interval_lo, c.push_interval_hi, c.push_range = time_left.data_ptr(), rnd7.data_ptr(), dt, lo_t, hi_t, rng6
      with torch.cuda.device(dev):
        native.check(s._lib.mjlab_control_step(ctypes.byref(s._m), ctypes.byref(s._d), ctypes.byref(c), s._stream()), "mjlab_control_step")
      return self._reset_mask
    target = self.default_joint + action * self.action_scale
    for _ in range(self.decimation // self.substeps_per_call):
      d.ctrl[:] = target
      self.sim.step(self.substeps_per_call)
    if self.fused_reset:
      native.check(
        s._lib.mjlab_masked_reset(ctypes.byref(s._m), ctypes.byref(s._d), self.key_qpos.data_ptr(), rnd3.data_ptr(),
                                  self.episode_length.data_ptr(), self.max_len, float(self.min_height),
                                  self._reset_mask.data_ptr(), 0 if self.env_origins is None else self.env_origins.data_ptr(),
                                  self.min_up_z, s._stream()),
        "mjlab_masked_reset",
      )
      reset = self._reset_mask
    else:
      self.episode_length.add_(1)
      if self.has_free:
        z0 = self.env_origins[:, 2] if self.env_origins is not None else 0.0
        up_z = 1.0 - 2.0 * (d.qpos[:, 4] ** 2 + d.qpos[:, 5] ** 2)
        fell = (d.qpos[:, 2] - z0 < self.min_height) | (up_z < self.min_up_z)
      else:
        fell = torch.zeros_like(self.episode_length, dtype=torch.bool)
      bad = ~torch.isfinite(d.qpos).all(dim=1)
      reset = fell | bad | (self.episode_length >= self.max_len)
      fresh = self._reset_qpos_from(rnd3)
      rm = reset.unsqueeze(1)
      d.qpos[:] = torch.where(rm, fresh, torch.nan_to_num(d.qpos))
      d.qvel[:] = torch.where(rm, torch.zeros_like(d.qvel), torch.nan_to_num(d.qvel))
      d.qacc_warmstart[:] = torch.where(rm, torch.zeros_like(d.qacc_warmstart), torch.nan_to_num(d.qacc_warmstart))
      self.episode_length.copy_(torch.where(reset, torch.zeros_like(self.episode_length), self.episode_length))
    self.sim.forward(reset if self.masked_forward else None)
    if self.push is not None:  # interval events come after the reset's forward() (manager_based_rl_env.py:134-137)
      lo_t, hi_t, rng6, time_left = self.push
      native.check(
        s._lib.mjlab_interval_push(ctypes.byref(s._m), ctypes.byref(s._d), time_left.data_ptr(), rnd7.data_ptr(),
                                   dt, lo_t, hi_t, ctypes.byref(rng6), 1, s._stream()),
        "mjlab_interval_push",
      )
    return reset

  def balance_worlds(self, simd_stride: int = 1024) -> None:
    """Deal the worlds out over the SIMDs by expected cost (control kernel only; results unchanged).

    The launch ends with its slowest SIMD, whose four waves share its issue slots.  With all nworld workgroups
    resident at once (4096 = 256 CUs x 16), workgroups b, b + 1024, b + 2048, b + 3072 land on the same SIMD
    (XCD = b % 8, CU and wave slot from b / 8 in dispatch order), so the worlds are ranked by the cost proxy
    of their last pass -- rows x (Newton iterations + 2) -- and dealt out in snake order: SIMD k gets ranks
    k, 2S-1-k, 2S+k, 4S-1-k.  Expensive worlds (fallen robots, many contacts) stay expensive for ~1 s, so
    calling this every few control steps is enough.  The permutation lives in a fixed buffer the captured
    graph reads, so it can be refreshed between replays."""
    n = self.sim.num_envs
    dev = self.key_qpos.device
    if n % simd_stride or n // simd_stride != 4:
      return  # the dispatch-order argument above needs exactly four workgroups per SIMD
    if self.world_order is None:
      self.world_order = torch.arange(n, dtype=torch.int32, device=dev)
      k = torch.arange(simd_stride, device=dev)
      S = simd_stride
      ranks = torch.stack([k, 2 * S - 1 - k, 2 * S + k, 4 * S - 1 - k])  # [t][k] = rank served by workgroup k + S t
      slot_of_rank = torch.empty(n, dtype=torch.long, device=dev)
      slot_of_rank[ranks.reshape(-1)] = torch.arange(n, device=dev)
      self._slot_of_rank = slot_of_rank
    d = self.sim.data
    cost = d.nefc.view(-1).float() * (d.solver_niter.view(-1).float() + 2.0)
    idx = torch.argsort(cost, descending=True)
    self.world_order[self._slot_of_rank] = idx.to(torch.int32)

  def random_action(self, out: torch.Tensor | None = None) -> torch.Tensor:
    """U(-1, 1) per actuator (reference scripts/play.py:159-172 "random" agent: ``2 rand - 1``), one launch;
    ``out=self.action_buffer`` writes the captured graph's input in place."""
    if out is None:
      out = torch.empty((self.sim.num_envs, self.m.nu), device=self.key_qpos.device)
    return out.uniform_(-1.0, 1.0, generator=self.gen)

  @property
  def action_buffer(self) -> torch.Tensor | None:
    """The static action input of the captured control-step graph (None before capture_graph())."""
    return self._action_buf if self._graph is not None else None

  def observation_rows(self) -> torch.Tensor:
    """A policy-observation-sized row per env (99 floats for G1) for the gather path."""
    d = self.sim.data
    parts = [d.qvel, d.qpos[:, 7:] if self.has_free else d.qpos, d.ctrl, d.sensordata]
    if self.has_free:
      parts.append(d.qpos[:, 3:7])
    if self._obs_buf is None:
      self._obs_buf = torch.cat(parts, dim=1)
    else:
      torch.cat(parts, dim=1, out=self._obs_buf)  # no allocation in the steady state
    return self._obs_buf


def g1_action_scale(model: Model) -> np.ndarray:
  from . import robots

  names = [model.names["joint"][j].split("/")[-1] for j in model.actuator_trnid[:, 0]]
  return robots.action_scale(robots.g1_actuators(), names).astype(np.float32)


def go1_action_scale(model: Model) -> np.ndarray:
  from . import robots

  names = [model.names["joint"][j].split("/")[-1] for j in model.actuator_trnid[:, 0]]
  return robots.action_scale(robots.go1_actuators(), names).astype(np.float32)

