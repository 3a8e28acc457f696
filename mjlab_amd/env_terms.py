"""Host side of the ENVIRONMENT TERMS of include/mjlab_amd.h: the reference's event and command terms that write mjData
(reference envs/mdp/events.py:42-143, tasks/velocity/mdp/velocity_command.py:64-102, managers/command_manager.py:44-66,
managers/event_manager.py:116-138) in mask-based form, one HIP launch per term on torch's current stream -- so that they can sit
inside the hipGraph of the whole control step (mjlab_amd/graphed_env.py).  Every function takes the torch tensors the reference's
term would read or write and the block of uniforms U (one row per world); nothing is returned, mjData / the term's buffers are
updated in place.  No fallback: without the HIP library these raise (mjlab_amd.native.NativeLibraryError)."""

from __future__ import annotations

import ctypes

import torch

from . import native

_vp = ctypes.c_void_p


class VelocityCommand(ctypes.Structure):  # mjlab_velocity_command_t
  _fields_ = [("nworld", ctypes.c_int), ("ldu", ctypes.c_int), ("ld_heading", ctypes.c_int), ("heading_command", ctypes.c_int),
              ("mask", _vp), ("U", _vp), ("ranges", _vp), ("heading_w", _vp), ("time_left", _vp), ("vel_command_b", _vp),
              ("heading_target", _vp), ("is_heading_env", _vp), ("is_standing_env", _vp), ("command_counter", _vp),
              ("dt", ctypes.c_float), ("resampling_lo", ctypes.c_float), ("resampling_hi", ctypes.c_float),
              ("rel_heading_envs", ctypes.c_float), ("rel_standing_envs", ctypes.c_float), ("heading_control_stiffness", ctypes.c_float),
              ("error_vel_xy", _vp), ("error_vel_yaw", _vp), ("root_link_lin_vel_b", _vp), ("root_link_ang_vel_b", _vp),
              ("ld_lin_vel", ctypes.c_int), ("ld_ang_vel", ctypes.c_int), ("inv_max_command_step", ctypes.c_float)]  # fmt: skip


def _stream(t: torch.Tensor) -> int:
  return torch.cuda.current_stream(t.device).cuda_stream


def _f32(t: torch.Tensor, what: str, inner_contiguous: bool = True) -> torch.Tensor:
  if t.dtype != torch.float32 or not t.is_cuda:
    raise TypeError(f"{what}: expected a float32 device tensor, got {t.dtype} on {t.device}")
  if inner_contiguous and t.dim() > 1:
    exp = 1
    for sz, st in zip(reversed(t.shape[1:]), reversed(t.stride()[1:]), strict=True):
      if sz > 1 and st != exp:
        raise ValueError(f"{what}: the trailing dimensions must be contiguous (shape {tuple(t.shape)}, strides {t.stride()})")
      exp *= sz
  return t


def _dense(t: torch.Tensor, what: str, dtype: torch.dtype) -> torch.Tensor:
  if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
    raise TypeError(f"{what}: expected a contiguous {dtype} device tensor, got {t.dtype}, strides {t.stride()} on {t.device}")
  return t


def _ld(t: torch.Tensor) -> int:
  return int(t.stride(0)) if t.shape[0] > 1 else 0


def reset_root_state_uniform(qpos: torch.Tensor, qvel: torch.Tensor, q_adr: int, v_adr: int, mask: torch.Tensor, default_root_state: torch.Tensor,
                             env_origins: torch.Tensor, U: torch.Tensor, pose_range: torch.Tensor, velocity_range: torch.Tensor) -> None:
  """qpos / qvel: mjData arrays (nworld, nq) / (nworld, nv); pose_range / velocity_range: device (2, 6) [lo; hi]; U: (nworld, >= 12)."""
  n = qpos.shape[0]
  native.check(native.lib().mjlab_event_reset_root_state_uniform(
    _dense(qpos, "qpos", torch.float32).data_ptr(), qpos.shape[1], q_adr, _dense(qvel, "qvel", torch.float32).data_ptr(), qvel.shape[1], v_adr, n,
    _dense(mask, "mask", torch.bool).data_ptr(), _f32(default_root_state, "default_root_state").data_ptr(), _ld(default_root_state),
    _dense(env_origins, "env_origins", torch.float32).data_ptr(), _f32(U, "U").data_ptr(), U.stride(0),
    _dense(pose_range, "pose_range", torch.float32).data_ptr(), _dense(velocity_range, "velocity_range", torch.float32).data_ptr(), _stream(qpos)),
    "mjlab_event_reset_root_state_uniform")  # fmt: skip


def reset_joints_by_scale(qpos: torch.Tensor, qvel: torch.Tensor, mask: torch.Tensor, joint_ids: torch.Tensor | None, q_adr: torch.Tensor, v_adr: torch.Tensor,
                          default_joint_pos: torch.Tensor, default_joint_vel: torch.Tensor, soft_joint_pos_limits: torch.Tensor, U: torch.Tensor,
                          ranges: torch.Tensor) -> None:
  """joint_ids / q_adr / v_adr: int32 device tensors of the nj selected joints (joint_ids None = all, in order); ranges: device
  [pos_lo, pos_hi, vel_lo, vel_hi]; U: (nworld, >= 2 nj)."""
  n, nj = qpos.shape[0], q_adr.numel()
  native.check(native.lib().mjlab_event_reset_joints_by_scale(
    _dense(qpos, "qpos", torch.float32).data_ptr(), qpos.shape[1], _dense(qvel, "qvel", torch.float32).data_ptr(), qvel.shape[1], n,
    _dense(mask, "mask", torch.bool).data_ptr(), nj, None if joint_ids is None else _dense(joint_ids, "joint_ids", torch.int32).data_ptr(),
    _dense(q_adr, "q_adr", torch.int32).data_ptr(), _dense(v_adr, "v_adr", torch.int32).data_ptr(),
    _f32(default_joint_pos, "default_joint_pos").data_ptr(), _ld(default_joint_pos), _f32(default_joint_vel, "default_joint_vel").data_ptr(), _ld(default_joint_vel),
    _f32(soft_joint_pos_limits, "soft_joint_pos_limits").data_ptr(), _ld(soft_joint_pos_limits), _f32(U, "U").data_ptr(), U.stride(0),
    _dense(ranges, "ranges", torch.float32).data_ptr(), _stream(qpos)),
    "mjlab_event_reset_joints_by_scale")  # fmt: skip


def push_by_setting_velocity(qvel: torch.Tensor, v_adr: int, time_left: torch.Tensor, dt: float, interval_range: torch.Tensor, root_link_vel_w: torch.Tensor,
                             root_link_quat_w: torch.Tensor, U: torch.Tensor, velocity_range: torch.Tensor) -> None:
  n = qvel.shape[0]
  native.check(native.lib().mjlab_event_push_by_setting_velocity(
    _dense(qvel, "qvel", torch.float32).data_ptr(), qvel.shape[1], v_adr, n, _dense(time_left, "time_left", torch.float32).data_ptr(), dt,
    _dense(interval_range, "interval_range", torch.float32).data_ptr(), _f32(root_link_vel_w, "root_link_vel_w").data_ptr(), root_link_vel_w.stride(0),
    _f32(root_link_quat_w, "root_link_quat_w").data_ptr(), root_link_quat_w.stride(0), _f32(U, "U").data_ptr(), U.stride(0),
    _dense(velocity_range, "velocity_range", torch.float32).data_ptr(), _stream(qvel)),
    "mjlab_event_push_by_setting_velocity")  # fmt: skip


def command_uniform_velocity(term, mask: torch.Tensor | None, U: torch.Tensor, ranges: torch.Tensor, dt: float, metrics: bool = False) -> None:
  """`term`: the reference's UniformVelocityCommand (its buffers are updated in place); mask None = compute(dt), else reset()
  for the worlds of the mask; ranges: device (4, 2) rows lin_vel_x, lin_vel_y, ang_vel_z, heading.  ``metrics`` (compute() only): the launch
  also runs ``_update_metrics`` (tasks/velocity/mdp/velocity_command.py:50-62) first, in place on ``term.metrics`` -- the caller then skips it."""
  cfg = term.cfg
  if cfg.init_velocity_prob > 0.0:
    raise NotImplementedError("command_uniform_velocity: the init-velocity branch is not covered by the fused term")
  c = VelocityCommand()
  c.nworld, c.ldu, c.heading_command = term.num_envs, U.stride(0), int(bool(cfg.heading_command))
  c.mask = None if mask is None else _dense(mask, "mask", torch.bool).data_ptr()
  c.U, c.ranges = _f32(U, "U").data_ptr(), _dense(ranges, "ranges", torch.float32).data_ptr()
  if mask is None and cfg.heading_command:
    h = _f32(term.robot.data.heading_w, "heading_w", inner_contiguous=False)
    c.heading_w, c.ld_heading = h.data_ptr(), int(h.stride(0))
  c.time_left = _dense(term.time_left, "time_left", torch.float32).data_ptr()
  c.vel_command_b = _dense(term.vel_command_b, "vel_command_b", torch.float32).data_ptr()
  c.heading_target = _dense(term.heading_target, "heading_target", torch.float32).data_ptr()
  c.is_heading_env = _dense(term.is_heading_env, "is_heading_env", torch.bool).data_ptr()
  c.is_standing_env = _dense(term.is_standing_env, "is_standing_env", torch.bool).data_ptr()
  c.command_counter = _dense(term.command_counter, "command_counter", torch.long).data_ptr()
  c.dt = dt
  c.resampling_lo, c.resampling_hi = cfg.resampling_time_range
  c.rel_heading_envs, c.rel_standing_envs = cfg.rel_heading_envs, cfg.rel_standing_envs
  c.heading_control_stiffness = cfg.heading_control_stiffness
  if metrics:
    if mask is not None:
      raise ValueError("command_uniform_velocity: the metrics belong to compute()")
    lv, av = _f32(term.robot.data.root_link_lin_vel_b, "root_link_lin_vel_b"), _f32(term.robot.data.root_link_ang_vel_b, "root_link_ang_vel_b")
    c.error_vel_xy = _dense(term.metrics["error_vel_xy"], "error_vel_xy", torch.float32).data_ptr()
    c.error_vel_yaw = _dense(term.metrics["error_vel_yaw"], "error_vel_yaw", torch.float32).data_ptr()
    c.root_link_lin_vel_b, c.root_link_ang_vel_b, c.ld_lin_vel, c.ld_ang_vel = lv.data_ptr(), av.data_ptr(), _ld(lv), _ld(av)
    # the reference divides by the Python float max_command_step; torch turns a division by a host scalar into a multiplication by 1 / scalar in float32
    import numpy as np

    c.inv_max_command_step = float(np.float32(1.0) / np.float32(cfg.resampling_time_range[1] / term._env.step_dt))
  native.check(native.lib().mjlab_command_uniform_velocity(ctypes.byref(c), _stream(U)), "mjlab_command_uniform_velocity")


class MotionTables(ctypes.Structure):  # mjlab_motion_tables_t
  _fields_ = [("joint_pos", _vp), ("joint_vel", _vp), ("body_pos_w", _vp), ("body_quat_w", _vp), ("body_lin_vel_w", _vp), ("body_ang_vel_w", _vp),
              ("body_indexes", _vp), ("nframe", ctypes.c_int), ("nj", ctypes.c_int), ("nbody_m", ctypes.c_int), ("nb", ctypes.c_int)]  # fmt: skip


def motion_tables(term) -> tuple:
  """(MotionTables, keep-alive) for the reference's MotionCommand `term`: the whole tables its MotionLoader holds
  (tasks/tracking/mdp/commands.py:28-50) and the tracked bodies' indices as int32."""
  mo = term.motion
  idx = mo._body_indexes.to(torch.int32).contiguous()
  t = MotionTables()
  t.joint_pos, t.joint_vel = _dense(mo.joint_pos, "joint_pos", torch.float32).data_ptr(), _dense(mo.joint_vel, "joint_vel", torch.float32).data_ptr()
  t.body_pos_w = _dense(mo._body_pos_w, "body_pos_w", torch.float32).data_ptr()
  t.body_quat_w = _dense(mo._body_quat_w, "body_quat_w", torch.float32).data_ptr()
  t.body_lin_vel_w = _dense(mo._body_lin_vel_w, "body_lin_vel_w", torch.float32).data_ptr()
  t.body_ang_vel_w = _dense(mo._body_ang_vel_w, "body_ang_vel_w", torch.float32).data_ptr()
  t.body_indexes = idx.data_ptr()
  t.nframe, t.nj, t.nbody_m, t.nb = mo.joint_pos.shape[0], mo.joint_pos.shape[1], mo._body_pos_w.shape[1], idx.numel()
  return t, (idx,)


def command_motion_write(tab: MotionTables, qpos: torch.Tensor, qvel: torch.Tensor, q_adr: int, v_adr: int, joint_q_adr: torch.Tensor, joint_v_adr: torch.Tensor,
                         mask: torch.Tensor, time_steps: torch.Tensor, env_origins: torch.Tensor, soft_joint_pos_limits: torch.Tensor, U: torch.Tensor,
                         pose_range: torch.Tensor, velocity_range: torch.Tensor, joint_range: tuple) -> None:
  """MotionCommand._resample_command's state write for the worlds of `mask` (U: (nworld, >= 12 + nj): pose, velocity, joint draws)."""
  native.check(native.lib().mjlab_command_motion_write(
    ctypes.byref(tab), _dense(qpos, "qpos", torch.float32).data_ptr(), qpos.shape[1], q_adr, _dense(qvel, "qvel", torch.float32).data_ptr(), qvel.shape[1], v_adr,
    _dense(joint_q_adr, "joint_q_adr", torch.int32).data_ptr(), _dense(joint_v_adr, "joint_v_adr", torch.int32).data_ptr(), qpos.shape[0],
    _dense(mask, "mask", torch.bool).data_ptr(), _dense(time_steps, "time_steps", torch.long).data_ptr(), _dense(env_origins, "env_origins", torch.float32).data_ptr(),
    _f32(soft_joint_pos_limits, "soft_joint_pos_limits").data_ptr(), _ld(soft_joint_pos_limits), _f32(U, "U").data_ptr(), U.stride(0),
    _dense(pose_range, "pose_range", torch.float32).data_ptr(), _dense(velocity_range, "velocity_range", torch.float32).data_ptr(),
    float(joint_range[0]), float(joint_range[1]), _stream(qpos)), "mjlab_command_motion_write")  # fmt: skip


def command_motion_relative(tab: MotionTables, time_steps: torch.Tensor, env_origins: torch.Tensor, xpos: torch.Tensor, xquat: torch.Tensor, anchor_body_id: int,
                            anchor_index: int, body_pos_relative_w: torch.Tensor, body_quat_relative_w: torch.Tensor, exact: int = 8 + 2 + 4 + 16 * 2 + 64 * 2) -> None:
  """``exact`` (include/mjlab_amd.h): 8 = the reference helpers' unfused rounding, + 2 / 4 = yaw_quat / quat_apply as NNC-fused, + 16 s1 + 64 s2 = the two
  quat_mul calls (0 unfused, 1 / 2 = fused for 2-D / 3-D operands); 0 = plain."""
  native.check(native.lib().mjlab_command_motion_relative(
    ctypes.byref(tab), xpos.shape[0], _dense(time_steps, "time_steps", torch.long).data_ptr(), _dense(env_origins, "env_origins", torch.float32).data_ptr(),
    _dense(xpos, "xpos", torch.float32).data_ptr(), _dense(xquat, "xquat", torch.float32).data_ptr(), xpos.shape[1], anchor_body_id, anchor_index,
    _dense(body_pos_relative_w, "body_pos_relative_w", torch.float32).data_ptr(), _dense(body_quat_relative_w, "body_quat_relative_w", torch.float32).data_ptr(),
    int(exact), _stream(xpos)), "mjlab_command_motion_relative")  # fmt: skip


class CopyEntry(ctypes.Structure):  # mjlab_copy_entry_t
  _fields_ = [("dst", _vp), ("src", _vp), ("nbytes", ctypes.c_ulonglong)]


def copy_batch(pairs: list) -> None:
  """``dst.copy_(src)`` for every (dst, src) of `pairs` in ONE launch per 32 pairs (``mjlab_copy_batch``); pairs the launch cannot
  express (another dtype, a strided tensor, a host tensor) keep ``copy_``."""
  todo = []
  for dst, src in pairs:
    if dst.is_cuda and src.is_cuda and dst.dtype == src.dtype and dst.shape == src.shape and dst.is_contiguous() and src.is_contiguous() and dst.numel() > 0:
      todo.append((dst, src))
    else:
      dst.copy_(src)
  if todo:
    arr = (CopyEntry * len(todo))()
    for k, (dst, src) in enumerate(todo):
      arr[k].dst, arr[k].src, arr[k].nbytes = dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size()
    native.check(native.lib().mjlab_copy_batch(arr, len(todo), _stream(todo[0][0])), "mjlab_copy_batch")


class MotionFrame:
  """MotionCommand's gathered properties from ONE launch per phase (``mjlab_command_motion_frame``): persistent output buffers (graph
  safe) under the reference's property names (tasks/tracking/mdp/commands.py:128-215).  ``anchor_*`` are the anchor's rows of the body
  arrays -- the same table entries (+ the same origin) the reference gathers separately."""

  NAMES = ("joint_pos", "joint_vel", "body_pos_w", "body_quat_w", "body_lin_vel_w", "body_ang_vel_w", "anchor_pos_w", "anchor_quat_w", "anchor_lin_vel_w",
           "anchor_ang_vel_w", "robot_body_pos_w", "robot_body_quat_w", "robot_body_lin_vel_w", "robot_body_ang_vel_w")

  def __init__(self, term, tab: MotionTables) -> None:
    dev, n, nb, nj = term.time_steps.device, term.num_envs, int(tab.nb), int(tab.nj)
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)  # noqa: E731
    self.term, self.tab = term, tab
    self.out = {"joint_pos": z(n, nj), "joint_vel": z(n, nj), "body_pos_w": z(n, nb, 3), "body_quat_w": z(n, nb, 4), "body_lin_vel_w": z(n, nb, 3),
                "body_ang_vel_w": z(n, nb, 3), "robot_body_pos_w": z(n, nb, 3), "robot_body_quat_w": z(n, nb, 4), "robot_body_lin_vel_w": z(n, nb, 3),
                "robot_body_ang_vel_w": z(n, nb, 3)}
    self.track = term.body_indexes.to(torch.int32).contiguous()
    a = int(term.motion_anchor_body_index)
    self.views = dict(self.out)
    for k in ("pos_w", "quat_w", "lin_vel_w", "ang_vel_w"):
      self.views["anchor_" + k] = self.out["body_" + k][:, a]

  def update(self, env_origins: torch.Tensor) -> dict:
    """Launch for the term's current ``time_steps`` and the robot's current ``EntityData``; returns name -> tensor."""
    t, o = self.term, self.out
    pose, vel = _f32(t.robot.data.body_link_pose_w, "body_link_pose_w"), _f32(t.robot.data.body_link_vel_w, "body_link_vel_w")
    if not (pose.is_contiguous() and vel.is_contiguous()):
      pose, vel = pose.contiguous(), vel.contiguous()
    native.check(native.lib().mjlab_command_motion_frame(
      ctypes.byref(self.tab), pose.shape[0], _dense(t.time_steps, "time_steps", torch.long).data_ptr(), _dense(env_origins, "env_origins", torch.float32).data_ptr(),
      pose.data_ptr(), vel.data_ptr(), pose.shape[1], self.track.data_ptr(), o["joint_pos"].data_ptr(), o["joint_vel"].data_ptr(), o["body_pos_w"].data_ptr(),
      o["body_quat_w"].data_ptr(), o["body_lin_vel_w"].data_ptr(), o["body_ang_vel_w"].data_ptr(), o["robot_body_pos_w"].data_ptr(),
      o["robot_body_quat_w"].data_ptr(), o["robot_body_lin_vel_w"].data_ptr(), o["robot_body_ang_vel_w"].data_ptr(), _stream(pose)), "mjlab_command_motion_frame")  # fmt: skip
    return self.views


class MotionSampleArgs(ctypes.Structure):  # mjlab_motion_sample_t
  _fields_ = [(k, _vp) for k in ("mask", "terminated", "time_steps", "U", "cdf", "entropy", "top1_prob", "top1_bin", "hist_out", "any_failed_out", "m_entropy",
                                 "m_top1_prob", "m_top1_bin")] + [("time_step_total", ctypes.c_longlong)] \
    + [(k, ctypes.c_int) for k in ("nworld", "ldu", "bin_count", "hist_always")] + [("time_left", _vp), ("command_counter", _vp)] \
    + [("resampling_width", ctypes.c_float), ("resampling_lo", ctypes.c_float)]


MOTION_SAMPLE_MAX_BINS = 4096  # include/mjlab_amd.h MJLAB_MOTION_SAMPLE_MAX_BINS


def command_motion_sample(term, mask: torch.Tensor, terminated: torch.Tensor, U: torch.Tensor, cdf: torch.Tensor, entropy: torch.Tensor, top1_prob: torch.Tensor,
                          top1_bin: torch.Tensor, hist_out: torch.Tensor, any_failed_out: torch.Tensor | None, resampling_time_range: tuple | None = None) -> None:
  """MotionCommand._adaptive_sampling's per-world part for the worlds of `mask` (``mjlab_command_motion_sample``): the failed worlds' bins
  counted into `hist_out` (when some world failed; always, with the 0 / 1 flag in `any_failed_out`, for a sharded caller), new phases by
  inverse CDF into ``term.time_steps``, the sampling metrics filled from the three device scalars when some world is masked.
  ``resampling_time_range`` (lo, hi): also CommandTerm._resample's timer ``time_left = U[:, 0] * (hi - lo) + lo`` and ``command_counter += 1``
  for the masked worlds (managers/command_manager.py:62-66)."""
  a = MotionSampleArgs()
  a.mask, a.terminated = _dense(mask, "mask", torch.bool).data_ptr(), _dense(terminated, "terminated", torch.bool).data_ptr()
  a.time_steps = _dense(term.time_steps, "time_steps", torch.long).data_ptr()
  a.U, a.ldu = _f32(U, "U").data_ptr(), int(U.stride(0))
  a.cdf = _dense(cdf, "cdf", torch.float32).data_ptr()
  for key, t in (("entropy", entropy), ("top1_prob", top1_prob), ("top1_bin", top1_bin)):
    if t.dtype != torch.float32 or not t.is_cuda or t.numel() != 1:
      raise TypeError(f"{key}: expected a float32 device scalar")
    setattr(a, key, t.data_ptr())
  a.hist_out = _dense(hist_out, "hist_out", torch.float32).data_ptr()
  a.any_failed_out = 0 if any_failed_out is None else any_failed_out.data_ptr()
  a.m_entropy, a.m_top1_prob, a.m_top1_bin = (_dense(term.metrics[k], k, torch.float32).data_ptr() for k in ("sampling_entropy", "sampling_top1_prob", "sampling_top1_bin"))
  a.time_step_total, a.nworld, a.bin_count, a.hist_always = int(term.motion.time_step_total), mask.shape[0], int(term.bin_count), int(any_failed_out is not None)
  if cdf.numel() != a.bin_count or hist_out.numel() != a.bin_count:
    raise ValueError("cdf / hist_out: bin_count entries expected")
  if resampling_time_range is not None:
    lo, hi = resampling_time_range
    a.time_left, a.command_counter = _dense(term.time_left, "time_left", torch.float32).data_ptr(), _dense(term.command_counter, "command_counter", torch.long).data_ptr()
    a.resampling_width, a.resampling_lo = float(hi - lo), float(lo)
  native.check(native.lib().mjlab_command_motion_sample(ctypes.byref(a), _stream(mask)), "mjlab_command_motion_sample")


class MotionSamplerArgs(ctypes.Structure):  # mjlab_motion_sampler_t
  _fields_ = [(k, _vp) for k in ("bin_failed_count", "current_bin_failed", "kernel", "cdf", "entropy", "top1_prob", "top1_bin")] \
    + [(k, ctypes.c_float) for k in ("alpha", "one_minus_alpha", "uniform_term")] + [(k, ctypes.c_int) for k in ("bin_count", "kernel_size", "do_update", "do_dist")]


class MotionSampler:
  """The adaptive sampler's global part for one MotionCommand (``mjlab_command_motion_sampler``): ``update()`` = the end of
  ``_update_command`` (bin_failed_count takes the step's histogram in: the reference's bits), ``distribution()`` = the sampling distribution
  of ``_adaptive_sampling`` into persistent buffers -- ``cdf`` and the three logged scalars that ``command_motion_sample`` reads."""

  def __init__(self, term) -> None:
    dev, nb = term.bin_failed_count.device, int(term.bin_count)
    self.term = term
    self.cdf = torch.zeros(nb, dtype=torch.float32, device=dev)
    self.scalars = torch.zeros(3, dtype=torch.float32, device=dev)
    self.entropy, self.top1_prob, self.top1_bin = self.scalars[0], self.scalars[1], self.scalars[2]
    self.kernel = _dense(term.kernel.to(torch.float32).contiguous(), "kernel", torch.float32)

  def _launch(self, update: bool, dist: bool) -> None:
    t, a = self.term, MotionSamplerArgs()
    a.bin_failed_count = _dense(t.bin_failed_count, "bin_failed_count", torch.float32).data_ptr()
    a.current_bin_failed = _dense(t._current_bin_failed, "_current_bin_failed", torch.float32).data_ptr()
    a.kernel, a.cdf = self.kernel.data_ptr(), self.cdf.data_ptr()
    a.entropy, a.top1_prob, a.top1_bin = self.entropy.data_ptr(), self.top1_prob.data_ptr(), self.top1_bin.data_ptr()
    alpha = float(t.cfg.adaptive_alpha)
    a.alpha, a.one_minus_alpha, a.uniform_term = alpha, 1 - alpha, float(t.cfg.adaptive_uniform_ratio) / float(t.bin_count)
    a.bin_count, a.kernel_size, a.do_update, a.do_dist = int(t.bin_count), int(self.kernel.numel()), int(update), int(dist)
    native.check(native.lib().mjlab_command_motion_sampler(ctypes.byref(a), _stream(self.cdf)), "mjlab_command_motion_sampler")

  def update(self) -> None:
    self._launch(True, False)

  def distribution(self) -> tuple:
    self._launch(False, True)
    return self.cdf, self.entropy, self.top1_prob, self.top1_bin


class MotionMetricsArgs(ctypes.Structure):  # mjlab_motion_metrics_t
  _fields_ = [(k, _vp) for k in ("body_pos_w", "body_quat_w", "body_lin_vel_w", "body_ang_vel_w", "robot_body_pos_w", "robot_body_quat_w", "robot_body_lin_vel_w",
                                 "robot_body_ang_vel_w", "body_pos_relative_w", "body_quat_relative_w", "joint_pos", "joint_vel", "robot_joint_pos", "robot_joint_vel", "out")] \
    + [(k, ctypes.c_int) for k in ("ld_robot_joint_pos", "ld_robot_joint_vel", "nworld", "nb", "nj", "anchor_index")]


class MotionMetrics:
  """MotionCommand._update_metrics (reference tasks/tracking/mdp/commands.py:221-254) as ONE launch (``mjlab_command_motion_metrics``) into a
  persistent (10, num_envs) buffer whose rows ARE the entries of ``term.metrics`` (bound once: the reference rebinds ten new tensors per
  call).  Logging quantities (CommandTerm.reset averages them into ``extras["log"]``): a few ulp from the reference's torch reductions."""

  KEYS = ("error_anchor_pos", "error_anchor_rot", "error_anchor_lin_vel", "error_anchor_ang_vel", "error_body_pos", "error_body_rot", "error_body_lin_vel",
          "error_body_ang_vel", "error_joint_pos", "error_joint_vel")

  def __init__(self, term) -> None:
    self.term = term
    self.buf = torch.zeros((len(self.KEYS), term.num_envs), dtype=torch.float32, device=term.time_steps.device)
    self.rows = [self.buf[k] for k in range(len(self.KEYS))]
    for row, key in zip(self.rows, self.KEYS, strict=True):
      if key in term.metrics:  # what the reference's last update left (an environment that resets before the next update logs it)
        row.copy_(term.metrics[key])
      term.metrics[key] = row

  def adopt(self) -> None:
    """Host side, before a replay: an entry somebody rebound since the last update (the reference's own ``_update_metrics`` in an eager
    step) hands its values to the row and is bound back -- the captured launches and the reset bookkeeping address the rows."""
    m = self.term.metrics
    for row, key in zip(self.rows, self.KEYS, strict=True):
      cur = m.get(key)
      if cur is not row:
        if cur is not None and cur.shape == row.shape:
          row.copy_(cur)
        m[key] = row

  def update(self) -> None:
    t = self.term
    a = MotionMetricsArgs()
    for key in ("body_pos_w", "body_quat_w", "body_lin_vel_w", "body_ang_vel_w", "robot_body_pos_w", "robot_body_quat_w", "robot_body_lin_vel_w", "robot_body_ang_vel_w",
                "body_pos_relative_w", "body_quat_relative_w", "joint_pos", "joint_vel"):
      setattr(a, key, _dense(getattr(t, key), key, torch.float32).data_ptr())
    rjp, rjv = _f32(t.robot_joint_pos, "robot_joint_pos"), _f32(t.robot_joint_vel, "robot_joint_vel")
    a.robot_joint_pos, a.robot_joint_vel, a.ld_robot_joint_pos, a.ld_robot_joint_vel = rjp.data_ptr(), rjv.data_ptr(), _ld(rjp), _ld(rjv)
    a.out = self.buf.data_ptr()
    a.nworld, a.nb, a.nj, a.anchor_index = t.num_envs, len(t.cfg.body_names), rjp.shape[1], int(t.motion_anchor_body_index)
    for row, key in zip(self.rows, self.KEYS, strict=True):  # (an eager reference step in between rebinds the entries: the rows stay the entries)
      t.metrics[key] = row
    native.check(native.lib().mjlab_command_motion_metrics(ctypes.byref(a), _stream(self.buf)), "mjlab_command_motion_metrics")


def reward_accumulate(values: torch.Tensor, weights: torch.Tensor, columns: torch.Tensor, dt: float, reward_buf: torch.Tensor, sum_ptrs: torch.Tensor,
                      step_reward: torch.Tensor) -> None:
  """The mjlab_reward_accumulate launch on plain tensors: `values` (k, n) raw term outputs, `weights` (k) float32, `columns` (k) int32,
  `sum_ptrs` (k) int64 device addresses of the episode-sum vectors (n float32 each)."""
  native.check(native.lib().mjlab_reward_accumulate(
    _dense(values, "values", torch.float32).data_ptr(), weights.data_ptr(), columns.data_ptr(), int(values.shape[0]), values.shape[1], float(dt),
    _dense(reward_buf, "reward_buf", torch.float32).data_ptr(), sum_ptrs.data_ptr(), _dense(step_reward, "step_reward", torch.float32).data_ptr(),
    step_reward.shape[1], _stream(values)), "mjlab_reward_accumulate")  # fmt: skip


class RewardAccumulator:
  """RewardManager.compute's accumulation loop (reference managers/reward_manager.py:77-89) as one launch for a given reward
  manager: the device tables (weights, step_reward columns, the episode-sum buffers' addresses) are built once."""

  def __init__(self, manager) -> None:
    self.manager = manager
    self.active = [(i, name, cfg) for i, (name, cfg) in enumerate(zip(manager._term_names, manager._term_cfgs, strict=True)) if cfg.weight != 0.0]
    self.idle = [i for i, cfg in enumerate(manager._term_cfgs) if cfg.weight == 0.0]
    dev = manager._reward_buf.device
    self._host_weights = [float(cfg.weight) for _, _, cfg in self.active]
    self.weights = torch.tensor(self._host_weights, dtype=torch.float32, device=dev)
    self.columns = torch.tensor([i for i, _, _ in self.active], dtype=torch.int32, device=dev)
    self.sums = [_dense(manager._episode_sums[name], "episode_sums", torch.float32) for _, name, _ in self.active]
    self.sum_ptrs = torch.tensor([t.data_ptr() for t in self.sums], dtype=torch.int64, device=dev)

  def refresh_weights(self) -> None:
    """The reference reads ``term_cfg.weight`` on every ``compute()`` (managers/reward_manager.py:82-86); here the weights sit in a
    device table a captured graph reads.  Called on the host once per control step (GraphedRlEnv.step, outside the graph): a weight
    that changed since the table was built -- a user's curriculum -- is uploaded (the next replay uses it); a term that crossed
    between zero and non-zero changes WHICH terms are evaluated, which a captured graph cannot follow: that raises."""
    cfgs = self.manager._term_cfgs
    if [i for i, cfg in enumerate(cfgs) if cfg.weight == 0.0] != self.idle:
      raise RuntimeError("RewardAccumulator: a reward term's weight crossed between zero and non-zero after the accumulator was built; "
                         "build a new GraphedRlEnv (the set of evaluated terms is part of the captured step)")
    now = [float(cfg.weight) for _, _, cfg in self.active]
    if now != self._host_weights:
      self._host_weights = now
      self.weights.copy_(torch.tensor(now, dtype=torch.float32), non_blocking=False)

  def compute(self, dt: float) -> torch.Tensor:
    m = self.manager
    for i in self.idle:
      m._step_reward[:, i] = 0.0
    if not self.active:
      m._reward_buf[:] = 0.0
      return m._reward_buf
    values = torch.stack([cfg.func(m._env, **cfg.params) for _, _, cfg in self.active], dim=0)  # (k, n): one launch for the k raw outputs
    if any(m._episode_sums[name] is not t for (_, name, _), t in zip(self.active, self.sums, strict=True)):
      raise RuntimeError("RewardAccumulator: an episode-sum buffer of the reward manager was replaced")
    reward_accumulate(values, self.weights, self.columns, dt, m._reward_buf, self.sum_ptrs, m._step_reward)
    return m._reward_buf


class _FillEntry(ctypes.Structure):  # mjlab_fill_entry_t
  _fields_ = [("ptr", _vp), ("pattern", ctypes.c_longlong), ("row_stride_bytes", ctypes.c_int), ("row_bytes", ctypes.c_int), ("elem_bytes", ctypes.c_int), ("from_device", ctypes.c_int)]


class _SumEntry(ctypes.Structure):  # mjlab_sum_entry_t
  _fields_ = [("ptr", _vp), ("is_bool", ctypes.c_int), ("pad_", ctypes.c_int)]


def _upload(entries: list, device) -> torch.Tensor:
  raw = b"".join(bytes(e) for e in entries)
  return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


class MaskedFill:
  """One launch for a list of masked row fills (``buf[mask] = value`` / ``buf[mask, a:b] = value`` of the managers' reset()).  Built
  once from (tensor, value) pairs -- views with a contiguous row are fine; the tensors are kept alive and must not be replaced."""

  def __init__(self, items: list) -> None:
    import struct

    entries, self.keep = [], []
    for t, value in items:
      if not t.is_cuda or (t.dim() > 1 and not t[0].is_contiguous()):
        raise TypeError(f"MaskedFill: a device tensor with contiguous rows is needed (shape {tuple(t.shape)}, strides {t.stride()})")
      eb = t.element_size()
      if eb not in (1, 4, 8):
        raise TypeError(f"MaskedFill: element size {eb}")
      from_device = 0
      if isinstance(value, torch.Tensor):  # a value read when the launch runs: an integer device scalar at least as wide as the elements
        if t.dtype.is_floating_point or t.dtype == torch.bool or value.dtype.is_floating_point or value.numel() != 1 or value.element_size() < eb or value.device != t.device:
          raise TypeError(f"MaskedFill: a device-valued fill needs integer elements and an integer scalar at least as wide ({t.dtype} <- {value.dtype})")
        pattern, from_device = value.data_ptr(), 1
        self.keep.append(value)
      elif t.dtype.is_floating_point:
        pattern = struct.unpack("<q", struct.pack("<d", float(value)))[0] if eb == 8 else struct.unpack("<i", struct.pack("<f", float(value)))[0]
      else:
        pattern = int(value)
      row = int(t[0].numel()) if t.dim() > 1 else 1
      entries.append(_FillEntry(t.data_ptr(), pattern, int(t.stride(0)) * eb, row * eb, eb, from_device))
      self.keep.append(t)
    self.n = items[0][0].shape[0]
    self.table = _upload(entries, items[0][0].device)
    self.count = len(entries)

  def __call__(self, mask: torch.Tensor) -> None:
    native.check(native.lib().mjlab_masked_fill_rows(self.table.data_ptr(), self.count, _dense(mask, "mask", torch.bool).data_ptr(), self.n, _stream(mask)),
                 "mjlab_masked_fill_rows")


class MaskedSums:
  """One launch for the masked sums the managers' reset() logs: ``out[i] = vectors[i][mask].sum()`` (bools counted), ``out[-1] =
  mask.sum()``.  The vectors (float32 or bool, shape (n,)) are kept alive and must not be replaced."""

  def __init__(self, vectors: list) -> None:
    self.keep = [_dense(v, "vector", v.dtype) for v in vectors]
    for v in vectors:
      if v.dtype not in (torch.float32, torch.bool) or v.dim() != 1:
        raise TypeError(f"MaskedSums: float32 or bool vectors, got {v.dtype} {tuple(v.shape)}")
    dev = vectors[0].device
    self.n, self.k = vectors[0].shape[0], len(vectors)
    self.table = _upload([_SumEntry(v.data_ptr(), int(v.dtype == torch.bool), 0) for v in vectors], dev)
    self.out = torch.zeros(self.k + 1, dtype=torch.float32, device=dev)

  def __call__(self, mask: torch.Tensor) -> torch.Tensor:
    native.check(native.lib().mjlab_masked_sums(self.table.data_ptr(), self.k, _dense(mask, "mask", torch.bool).data_ptr(), self.n, self.out.data_ptr(), _stream(mask)),
                 "mjlab_masked_sums")
    return self.out
