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
              ("rel_heading_envs", ctypes.c_float), ("rel_standing_envs", ctypes.c_float), ("heading_control_stiffness", ctypes.c_float)]  # fmt: skip


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


def command_uniform_velocity(term, mask: torch.Tensor | None, U: torch.Tensor, ranges: torch.Tensor, dt: float) -> None:
  """`term`: the reference's UniformVelocityCommand (its buffers are updated in place); mask None = compute(dt), else reset()
  for the worlds of the mask; ranges: device (4, 2) rows lin_vel_x, lin_vel_y, ang_vel_z, heading."""
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
  native.check(native.lib().mjlab_command_uniform_velocity(ctypes.byref(c), _stream(U)), "mjlab_command_uniform_velocity")
