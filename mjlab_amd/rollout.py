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


class _MotionReset(ctypes.Structure):
  """mjlab_motion_reset_t (include/mjlab_amd.h)."""

  _fields_ = [
    ("joint_pos", ctypes.c_void_p), ("joint_vel", ctypes.c_void_p), ("root_pos", ctypes.c_void_p), ("root_quat", ctypes.c_void_p),
    ("root_lin_vel", ctypes.c_void_p), ("root_ang_vel", ctypes.c_void_p), ("soft_limits", ctypes.c_void_p), ("rnd", ctypes.c_void_p),
    ("time_steps", ctypes.c_void_p), ("nframe", ctypes.c_int), ("bins", ctypes.c_int),
    ("pose_lo", ctypes.c_float * 6), ("pose_hi", ctypes.c_float * 6), ("vel_lo", ctypes.c_float * 6), ("vel_hi", ctypes.c_float * 6),
    ("joint_lo", ctypes.c_float), ("joint_hi", ctypes.c_float), ("dz", ctypes.c_float), ("dup", ctypes.c_float),
  ]  # fmt: skip


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
    ("reset_qpos", ctypes.c_void_p), ("reset_qvel", ctypes.c_void_p), ("term_ref", ctypes.c_void_p), ("term_dz", ctypes.c_float), ("term_dup", ctypes.c_float),
    ("motion", ctypes.c_void_p),
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


# The tracking task's events that touch the physics state or model (reference src/mjlab/tasks/tracking/tracking_env_cfg.py:29-36,
# 58-76,154-199,256-283,306 and config/g1/flat_env_cfg.py): resets to a random phase of the motion (MotionCommand._resample_command,
# mdp/commands.py:299-363) with the cfg's pose / velocity / joint noise, a 6-component root-velocity kick every U(1, 3) s, startup
# randomisation of foot friction, torso com and joint zero offsets, termination when the anchor leaves the motion frame
# (|z error| > 0.25 m, |projected-gravity z error| > 0.8), 10 s episodes.
TRACKING_TASK_EVENTS = {
  "g1": {
    "friction_range": (0.3, 1.2), "friction_geoms": r"(left|right)_foot[1-7]_collision$",
    "push": {"interval_s": (1.0, 3.0), "velocity": {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "z": (-0.2, 0.2), "roll": (-0.52, 0.52), "pitch": (-0.52, 0.52), "yaw": (-0.78, 0.78)}},
    "motion_reset": {
      "pose_range": {"x": (-0.05, 0.05), "y": (-0.05, 0.05), "z": (-0.01, 0.01), "roll": (-0.1, 0.1), "pitch": (-0.1, 0.1), "yaw": (-0.2, 0.2)},
      "velocity_range": {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "z": (-0.2, 0.2), "roll": (-0.52, 0.52), "pitch": (-0.52, 0.52), "yaw": (-0.78, 0.78)},
      "joint_position_range": (-0.1, 0.1), "soft_joint_pos_limit_factor": 0.9, "anchor_dz": 0.25, "anchor_dup": 0.8,
    },
    "com_body": "torso_link", "com_ranges": ((-0.025, 0.025), (-0.05, 0.05), (-0.05, 0.05)), "qpos0_range": (-0.01, 0.01),
    "episode_length_s": 10.0,
  },
}  # fmt: skip


def synthetic_motion(model: Model, nframe: int = 500, fps: float = 50.0, amplitude: float = 0.2, freq_hz: float = 0.5, root_z: float = 0.76,
                     seed: int = 0) -> dict[str, np.ndarray]:
  """A stand-in for the tracking task's ``motion.npz`` (none is in the reference tree: SURVEY.md section 8d(4)): 10 s at 50 fps,
  every joint sweeping +-0.2 rad at 0.5 Hz around the keyframe with its own phase, the pelvis held at z = 0.76 m, upright.
  Keys follow the loader (reference tasks/tracking/mdp/commands.py:30-50) for the quantities a reset reads: ``joint_pos``,
  ``joint_vel`` (T, nq - 7) and the anchor body's ``body_pos_w`` / ``body_quat_w`` / ``body_lin_vel_w`` / ``body_ang_vel_w`` (T, 1, .)
  -- the anchor here is the floating base itself (the reference's G1 config anchors on the torso link, one waist chain away)."""
  rng = np.random.default_rng(seed)
  t = np.arange(nframe, dtype=np.float64) / fps
  key = np.asarray(model.key_qpos[0] if model.nkey else model.qpos0, dtype=np.float64)
  nj = model.nq - 7
  phase = rng.uniform(0.0, 2.0 * np.pi, nj)
  arg = 2.0 * np.pi * freq_hz * t[:, None] + phase[None, :]
  jp = key[7:][None, :] + amplitude * np.sin(arg)
  jv = amplitude * 2.0 * np.pi * freq_hz * np.cos(arg)
  hinge = [j for j in range(model.njnt) if model.jnt_type[j] != JNT_FREE]
  lo, hi = np.asarray(model.jnt_range)[hinge, 0], np.asarray(model.jnt_range)[hinge, 1]
  limited = np.asarray(model.jnt_limited)[hinge].astype(bool)
  jp = np.where(limited[None, :], np.clip(jp, lo[None, :] + 0.02, hi[None, :] - 0.02), jp)
  z = np.zeros((nframe, 1, 3))
  pos = z.copy()
  pos[:, 0, 2] = root_z
  quat = np.zeros((nframe, 1, 4))
  quat[:, 0, 0] = 1.0
  return {"joint_pos": jp.astype(np.float32), "joint_vel": jv.astype(np.float32), "body_pos_w": pos.astype(np.float32),
          "body_quat_w": quat.astype(np.float32), "body_lin_vel_w": z.astype(np.float32), "body_ang_vel_w": z.astype(np.float32), "fps": np.float32(fps)}


def write_motion_npz(path: str, model: Model, device: str = "cuda:0", motion: dict | None = None) -> tuple:
  """A complete ``motion.npz`` for the tracking task's ``MotionLoader`` (reference tasks/tracking/mdp/commands.py:30-65; keys and shapes
  of scripts/csv_to_npz.py:298-309): the joint trajectory and root pose of `motion` (default: ``synthetic_motion``) with the pose and
  velocity of EVERY robot body, from this package's forward kinematics -- one world per frame through ``Simulation.forward()`` (what
  the reference's converter does with one ``mj_forward`` per frame).  Returns the shape of ``body_pos_w``."""
  from .sim import Simulation, SimulationCfg

  mo = synthetic_motion(model) if motion is None else motion
  nframe = mo["joint_pos"].shape[0]
  sim = Simulation(nframe, SimulationCfg(njmax=250, use_graph=False), model, device)
  d = sim.data
  dev = d.qpos.device
  d.qpos[:, 0:3] = torch.from_numpy(mo["body_pos_w"][:, 0]).to(dev)
  d.qpos[:, 3:7] = torch.from_numpy(mo["body_quat_w"][:, 0]).to(dev)
  d.qpos[:, 7:] = torch.from_numpy(mo["joint_pos"]).to(dev)
  d.qvel[:] = 0.0
  d.qvel[:, 6:] = torch.from_numpy(mo["joint_vel"]).to(dev)
  sim.forward()
  root = int(model.jnt_bodyid[0])
  pos, quat, cv = d.xpos[:, root:].double(), d.xquat[:, root:].double(), d.cvel[:, root:].double()  # the robot's bodies (world and terrain come first)
  sub = d.subtree_com[:, root].double()[:, None, :]
  lin = cv[..., 3:] - torch.cross(cv[..., :3], (sub - pos).expand_as(pos), dim=-1)  # body-origin velocity from the com-based spatial velocity (entity/data.py:20-31)
  f32 = lambda t: t.float().cpu().numpy()  # noqa: E731
  np.savez(path, fps=np.array([float(mo.get("fps", 50.0))]), joint_pos=mo["joint_pos"], joint_vel=mo["joint_vel"], body_pos_w=f32(pos), body_quat_w=f32(quat),
           body_lin_vel_w=f32(lin), body_ang_vel_w=f32(cv[..., :3]))
  return tuple(pos.shape)


def _quat_from_euler_xyz(r: torch.Tensor, p: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
  """(w, x, y, z) of roll-pitch-yaw, the convention of the reference's quat_from_euler_xyz (isaaclab utils/math.py)."""
  cr, sr, cp, sp, cy, sy = torch.cos(r * 0.5), torch.sin(r * 0.5), torch.cos(p * 0.5), torch.sin(p * 0.5), torch.cos(y * 0.5), torch.sin(y * 0.5)
  return torch.stack([cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp], dim=-1)


def _quat_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
  aw, ax, ay, az = a.unbind(-1)
  bw, bx, by, bz = b.unbind(-1)
  return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                      aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], dim=-1)


def _quat_apply_inverse(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
  xyz = q[..., 1:]
  t = torch.cross(xyz, v, dim=-1) * 2.0
  return v - q[..., 0:1] * t + torch.cross(xyz, t, dim=-1)


def motion_reset_tables(m: Model, motion: dict, motion_reset: dict | None, decimation: int, n: int, dev) -> dict:
  """Device tables and parameters of the tracking task's reset (reference tasks/tracking/mdp/commands.py:30-65 MotionLoader,
  :107 the adaptive sampler's one-second bins, tracking_env_cfg.py:56-76 the cfg's noise ranges): what ``motion_reset_rows`` and the
  control kernel (``mjlab_motion_reset_t``) read."""
  cfg = dict(TRACKING_TASK_EVENTS["g1"]["motion_reset"], **(motion_reset or {}))
  tab = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32, device=dev) for k, v in motion.items() if k != "fps"}
  nframe = int(tab["joint_pos"].shape[0])
  hinge = [j for j in range(m.njnt) if m.jnt_type[j] != JNT_FREE]
  rng_ = np.asarray(m.jnt_range, dtype=np.float64)[hinge]
  mean, half = 0.5 * (rng_[:, 0] + rng_[:, 1]), 0.5 * (rng_[:, 1] - rng_[:, 0]) * cfg["soft_joint_pos_limit_factor"]
  limited = np.asarray(m.jnt_limited)[hinge].astype(bool)
  soft_lo = torch.tensor(np.where(limited, mean - half, -np.inf), dtype=torch.float32, device=dev)
  soft_hi = torch.tensor(np.where(limited, mean + half, np.inf), dtype=torch.float32, device=dev)
  six = ("x", "y", "z", "roll", "pitch", "yaw")
  pr = torch.tensor([cfg["pose_range"].get(k, (0.0, 0.0)) for k in six], dtype=torch.float32, device=dev)
  vr = torch.tensor([cfg["velocity_range"].get(k, (0.0, 0.0)) for k in six], dtype=torch.float32, device=dev)
  # bins of one second like the reference's adaptive sampler (commands.py:107); nothing has failed yet: uniform over the bins
  return {"tab": tab, "nframe": nframe, "bins": int(nframe // (1.0 / (m.opt.timestep * decimation))) + 1, "soft_lo": soft_lo, "soft_hi": soft_hi,
          "pose_range": pr, "velocity_range": vr, "joint_range": cfg["joint_position_range"], "dz": float(cfg["anchor_dz"]), "dup": float(cfg["anchor_dup"]),
          "time_steps": torch.zeros((n,), dtype=torch.int32, device=dev), "rnd": torch.zeros((n, 14 + m.nq - 7), device=dev),
          "reset_qpos": torch.zeros((n, m.nq), device=dev), "reset_qvel": torch.zeros((n, m.nv), device=dev), "term_ref": torch.zeros((n, 2), device=dev)}


def motion_reset_rows(mo: dict, env_origins: torch.Tensor | None) -> torch.Tensor:
  """Refresh, for EVERY world, the state its reset would write now (reference MotionCommand._resample_command,
  tasks/tracking/mdp/commands.py:299-363: a motion frame of a freshly sampled phase + the cfg's noise, through
  write_joint_state_to_sim / write_root_state_to_sim) and the termination reference of its current phase; returns the
  sampled phases.  Plain device ops, no host sync: which worlds take their row is decided by the reset itself.  The control
  kernel computes the same rows per resetting world (bit for bit: tests/test_gpu_fullsize.py); this function is checked element
  by element against the reference's own ``_resample_command`` fed the same uniforms (tests/test_reference_env.py)."""
  tab, T = mo["tab"], mo["nframe"]
  u = mo["rnd"]  # refreshed by the caller: the same uniforms the control kernel reads (mjlab_motion_reset_t.rnd)
  bins = torch.clamp((u[:, 0] * mo["bins"]).long(), max=mo["bins"] - 1)
  t_new = ((bins.float() + u[:, 1]) / mo["bins"] * (T - 1)).long()
  pose = mo["pose_range"][:, 0] + (mo["pose_range"][:, 1] - mo["pose_range"][:, 0]) * u[:, 2:8]
  vel = mo["velocity_range"][:, 0] + (mo["velocity_range"][:, 1] - mo["velocity_range"][:, 0]) * u[:, 8:14]
  root_pos = tab["body_pos_w"][t_new, 0] + pose[:, 0:3]
  if env_origins is not None:
    root_pos = root_pos + env_origins
  root_ori = _quat_mul(_quat_from_euler_xyz(pose[:, 3], pose[:, 4], pose[:, 5]), tab["body_quat_w"][t_new, 0])
  lin = tab["body_lin_vel_w"][t_new, 0] + vel[:, 0:3]
  ang = _quat_apply_inverse(root_ori, tab["body_ang_vel_w"][t_new, 0] + vel[:, 3:6])
  lo_j, hi_j = mo["joint_range"]
  jp = torch.minimum(torch.maximum(tab["joint_pos"][t_new] + lo_j + (hi_j - lo_j) * u[:, 14:], mo["soft_lo"]), mo["soft_hi"])
  torch.cat([root_pos, root_ori, jp], dim=1, out=mo["reset_qpos"])
  torch.cat([lin, ang, tab["joint_vel"][t_new]], dim=1, out=mo["reset_qvel"])
  t_cur = torch.clamp(mo["time_steps"].long(), max=T - 1)
  q = tab["body_quat_w"][t_cur, 0]
  z = tab["body_pos_w"][t_cur, 0, 2]
  if env_origins is not None:
    z = z + env_origins[:, 2]
  torch.stack([z, 1.0 - 2.0 * (q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2])], dim=1, out=mo["term_ref"])
  return t_new


class PhysicsRollout:
  def __init__(self, sim: Simulation, action_scale: np.ndarray | float = 0.25, decimation: int = 4,
               episode_length_s: float = 20.0, min_height: float = 0.3, seed: int = 42, key: int = 0,
               masked_forward: bool = False, fused_reset: bool = True, min_up_z: float | None = None,
               max_init_terrain_level: int | None = 5, friction_range: tuple[float, float] | None = None,
               friction_geoms: str | None = None, push: dict | None = None, bad_orientation_deg: float | None = None,
               substeps_per_call: int = 1, control_kernel: bool = False, motion: dict | None = None, motion_reset: dict | None = None,
               com_body: str | None = None, com_ranges: tuple | None = None, qpos0_range: tuple | None = None) -> None:
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
    # startup events of the tracking task: torso com offset (randomize_field "add" on body_ipos), joint zero offsets (qpos0)
    if com_body is not None:
      b = next(i for i, name in enumerate(m.names["body"]) if name and name.split("/")[-1] == com_body)
      sim.expand_model_fields(["body_ipos"])
      r = torch.rand((n, 3), device=dev, generator=self.gen)
      lo_ = torch.tensor([c[0] for c in com_ranges], device=dev)
      hi_ = torch.tensor([c[1] for c in com_ranges], device=dev)
      sim.model.body_ipos[:, b] += lo_ + (hi_ - lo_) * r
      sim.create_graph()
    if qpos0_range is not None:
      sim.expand_model_fields(["qpos0"])
      sim.model.qpos0[:, 7:] += qpos0_range[0] + (qpos0_range[1] - qpos0_range[0]) * torch.rand((n, m.nq - 7), device=dev, generator=self.gen)
      sim.create_graph()
    # resets to a phase of a motion (the tracking task): tables on the device, a phase counter per world
    self.motion = None
    if motion is not None:
      if not self.has_free:
        raise ValueError("motion resets need a floating base")
      self.motion = motion_reset_tables(m, motion, motion_reset, decimation, n, dev)
      tab, nframe, soft_lo, soft_hi, pr, vr = (self.motion[k] for k in ("tab", "nframe", "soft_lo", "soft_hi", "pose_range", "velocity_range"))
      cfg = dict(TRACKING_TASK_EVENTS["g1"]["motion_reset"], **(motion_reset or {}))
      if not (control_kernel or not fused_reset):
        raise ValueError("motion resets run in the control kernel or in the torch reset chain (fused_reset=False)")
      # the control kernel does all of this per resetting world itself (mjlab_motion_reset_t): tables and parameters, once
      mo = self.motion
      mo["anchor"] = {k: tab[k][:, 0].contiguous() for k in ("body_pos_w", "body_quat_w", "body_lin_vel_w", "body_ang_vel_w")}
      mo["soft"] = torch.stack([soft_lo, soft_hi], dim=1).contiguous()
      st = _MotionReset()
      st.joint_pos, st.joint_vel = tab["joint_pos"].contiguous().data_ptr(), tab["joint_vel"].contiguous().data_ptr()
      st.root_pos, st.root_quat = mo["anchor"]["body_pos_w"].data_ptr(), mo["anchor"]["body_quat_w"].data_ptr()
      st.root_lin_vel, st.root_ang_vel = mo["anchor"]["body_lin_vel_w"].data_ptr(), mo["anchor"]["body_ang_vel_w"].data_ptr()
      st.soft_limits, st.rnd, st.time_steps = mo["soft"].data_ptr(), mo["rnd"].data_ptr(), mo["time_steps"].data_ptr()
      st.nframe, st.bins = nframe, mo["bins"]
      for k in range(6):
        st.pose_lo[k], st.pose_hi[k], st.vel_lo[k], st.vel_hi[k] = float(pr[k, 0]), float(pr[k, 1]), float(vr[k, 0]), float(vr[k, 1])
      st.joint_lo, st.joint_hi, st.dz, st.dup = float(cfg["joint_position_range"][0]), float(cfg["joint_position_range"][1]), mo["dz"], mo["dup"]
      mo["struct_dev"] = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).to(dev)
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

  def _motion_rows(self) -> torch.Tensor:
    return motion_reset_rows(self.motion, self.env_origins)

  def reset_all(self) -> None:
    d = self.sim.data
    n = self.sim.num_envs
    if self.motion is not None:
      torch.rand(self.motion["rnd"].shape, device=self.key_qpos.device, generator=self.gen, out=self.motion["rnd"])
      t_new = self._motion_rows()
      d.qpos[:] = self.motion["reset_qpos"]
      d.qvel[:] = self.motion["reset_qvel"]
      self.motion["time_steps"].copy_(t_new.to(torch.int32))
      d.ctrl[:] = self.default_joint
      d.qacc_warmstart[:] = 0.0
      self.sim.forward()
      return
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
    t_new = None
    if self.motion is not None:
      mo = self.motion
      torch.rand(mo["rnd"].shape, device=dev, generator=self.gen, out=mo["rnd"])
      if not (self.control_kernel and self.fused_reset):  # the torch chain; the control kernel does the same per resetting world
        t_new = self._motion_rows()
        # a motion that has ended resamples like a reset (commands.py:365-369 _update_command): those worlds time out this step
        ended = mo["time_steps"] + 1 >= mo["nframe"]
        self.episode_length.copy_(torch.where(ended, torch.full_like(self.episode_length, self.max_len - 1), self.episode_length))
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
      if self.motion is not None:
        c.motion = self.motion["struct_dev"].data_ptr()
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
      if self.motion is not None:
        mo = self.motion
        up_z = 1.0 - 2.0 * (d.qpos[:, 4] ** 2 + d.qpos[:, 5] ** 2)
        fell = fell | ((d.qpos[:, 2] - mo["term_ref"][:, 0]).abs() > mo["dz"]) | ((up_z - mo["term_ref"][:, 1]).abs() > mo["dup"])
      bad = ~torch.isfinite(d.qpos).all(dim=1)
      reset = fell | bad | (self.episode_length >= self.max_len)
      fresh = self._reset_qpos_from(rnd3) if self.motion is None else self.motion["reset_qpos"]
      fresh_v = torch.zeros_like(d.qvel) if self.motion is None else self.motion["reset_qvel"]
      rm = reset.unsqueeze(1)
      d.qpos[:] = torch.where(rm, fresh, torch.nan_to_num(d.qpos))
      d.qvel[:] = torch.where(rm, fresh_v, torch.nan_to_num(d.qvel))
      if self.motion is not None:
        self.motion["time_steps"].copy_(torch.where(reset, t_new, self.motion["time_steps"].long() + 1).to(torch.int32))
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

