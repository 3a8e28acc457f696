"""The ENVIRONMENT TERMS of include/mjlab_amd.h (mjlab_amd/env_terms.py: one HIP launch per event / command term, mask based)
against the formulas of the reference's terms written out in torch fp64 here -- reference envs/mdp/events.py:42-143,
tasks/velocity/mdp/velocity_command.py:64-102, managers/command_manager.py:44-66, managers/event_manager.py:116-138 and the
helpers third_party/isaaclab/isaaclab/utils/math.py:96 / :269 / :521 / :645 / :1354.  Needs no reference tree (the comparison
with the reference's own functions inside the captured environment step: tests/test_gpu_reference_env.py).

Tolerance: 2e-6 absolute on quantities of magnitude <= ~4 (fp32 kernels against fp64 formulas); rows outside the mask, and
decisions (booleans, counters, timers that did not run out), bit for bit."""

import math
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-6
N = 777  # (not a multiple of the block size)


def _dev():
  return torch.device("cuda:0")


def quat_from_euler_xyz(r, p, y):
  cy, sy, cr, sr, cp, sp = torch.cos(y / 2), torch.sin(y / 2), torch.cos(r / 2), torch.sin(r / 2), torch.cos(p / 2), torch.sin(p / 2)
  return torch.stack([cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp], -1)


def quat_mul(a, b):
  w1, x1, y1, z1 = a.unbind(-1)
  w2, x2, y2, z2 = b.unbind(-1)
  return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                      w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)  # fmt: skip


def quat_apply_inverse(q, v):
  xyz = q[..., 1:]
  t = torch.cross(xyz, v, dim=-1) * 2
  return v - q[..., 0:1] * t + torch.cross(xyz, t, dim=-1)


def wrap_to_pi(a):
  w = (a + math.pi) % (2 * math.pi)
  return torch.where((w == 0) & (a > 0), torch.full_like(a, math.pi), w - math.pi)


def _unit_quats(n, g):
  q = torch.randn((n, 4), generator=g, dtype=torch.float64)
  return q / q.norm(dim=-1, keepdim=True)


def _setup(seed, nq=36, nv=35):
  g = torch.Generator().manual_seed(seed)
  qpos = torch.randn((N, nq), generator=g)
  qvel = torch.randn((N, nv), generator=g)
  mask = torch.rand(N, generator=g) < 0.3
  return g, qpos, qvel, mask


@pytest.mark.parametrize("shared_root", [True, False])
def test_reset_root_state_uniform(shared_root):
  from mjlab_amd import env_terms

  dev = _dev()
  g, qpos, qvel, mask = _setup(1)
  root = torch.cat([torch.randn((N, 3), generator=g, dtype=torch.float64), _unit_quats(N, g), torch.randn((N, 6), generator=g, dtype=torch.float64) * 0.3], -1)
  if shared_root:
    root = root[:1].expand(N, 13)
  org = torch.randn((N, 3), generator=g, dtype=torch.float64) * 5
  U = torch.rand((N, 20), generator=g)  # a wider block: the term reads columns 3..15 of its row
  pose = torch.tensor([[-0.5, -0.5, 0.0, -0.2, -0.1, -3.14], [0.5, 0.5, 0.1, 0.2, 0.1, 3.14]])
  vel = torch.tensor([[-0.5, -0.4, -0.3, -0.2, -0.1, -0.6], [0.5, 0.4, 0.3, 0.2, 0.1, 0.6]])
  q_adr, v_adr = 0, 0
  dq, dv, du = qpos.to(dev), qvel.to(dev), U.to(dev)
  root32 = root.float().to(dev) if not shared_root else root[:1].float().to(dev).expand(N, 13)
  env_terms.reset_root_state_uniform(dq, dv, q_adr, v_adr, mask.to(dev), root32, org.float().to(dev), du[:, 3:15], pose.to(dev), vel.to(dev))
  torch.cuda.synchronize()
  # reference formulas on the fp32-rounded inputs, in fp64
  r64, o64, u = root32.cpu().double(), org.float().double(), U[:, 3:15].double()
  rs = u[:, :6] * (pose[1] - pose[0]).double() + pose[0].double()
  pos = r64[:, 0:3] + rs[:, 0:3] + o64
  ori = quat_mul(r64[:, 3:7], quat_from_euler_xyz(rs[:, 3], rs[:, 4], rs[:, 5]))
  v = r64[:, 7:13] + u[:, 6:] * (vel[1] - vel[0]).double() + vel[0].double()
  v = torch.cat([v[:, :3], quat_apply_inverse(ori, v[:, 3:])], -1)
  want_q, want_v = qpos.clone().double(), qvel.clone().double()
  want_q[mask, 0:7] = torch.cat([pos, ori], -1)[mask]
  want_v[mask, 0:6] = v[mask]
  got_q, got_v = dq.cpu(), dv.cpu()
  assert torch.equal(got_q[~mask], qpos[~mask]) and torch.equal(got_v[~mask], qvel[~mask])
  assert torch.equal(got_q[:, 7:], qpos[:, 7:]) and torch.equal(got_v[:, 6:], qvel[:, 6:])
  assert (got_q.double() - want_q).abs().max() <= 5 * TOL, (got_q.double() - want_q).abs().max()  # (positions up to ~20 m)
  assert (got_v.double() - want_v).abs().max() <= TOL
  assert mask.sum() > 100


@pytest.mark.parametrize("subset", [False, True])
def test_reset_joints_by_scale(subset):
  from mjlab_amd import env_terms

  dev = _dev()
  g, qpos, qvel, mask = _setup(2)
  nj_all = 29
  ids = torch.tensor([0, 3, 4, 11, 28]) if subset else None
  sel = ids if subset else torch.arange(nj_all)
  nj = len(sel)
  jpos = (torch.randn((1, nj_all), generator=g)).expand(N, nj_all)  # the reference's defaults: one row shared by all envs
  jvel = torch.randn((N, nj_all), generator=g)
  lim = torch.stack([jpos[0] - torch.rand(nj_all, generator=g), jpos[0] + torch.rand(nj_all, generator=g)], -1)[None].expand(N, nj_all, 2)
  U = torch.rand((N, 2 * nj + 3), generator=g)
  ranges = torch.tensor([0.5, 1.5, -1.0, 1.0])
  qa, va = 7 + sel, 6 + sel
  dq, dv = qpos.to(dev), qvel.to(dev)
  env_terms.reset_joints_by_scale(dq, dv, mask.to(dev), None if ids is None else ids.to(dev, torch.int32), qa.to(dev, torch.int32), va.to(dev, torch.int32),
                                  jpos[:1].to(dev).expand(N, nj_all), jvel.to(dev), lim[:1].to(dev).expand(N, nj_all, 2), U.to(dev)[:, 1 : 1 + 2 * nj], ranges.to(dev))
  torch.cuda.synchronize()
  u = U[:, 1 : 1 + 2 * nj].double()
  p = jpos[:, sel].double() * (u[:, :nj] * (ranges[1] - ranges[0]).double() + ranges[0].double())
  p = torch.minimum(torch.maximum(p, lim[:, sel, 0].double()), lim[:, sel, 1].double())
  v = jvel[:, sel].double() * (u[:, nj:] * (ranges[3] - ranges[2]).double() + ranges[2].double())
  want_q, want_v = qpos.clone().double(), qvel.clone().double()
  rows = mask.nonzero().flatten()
  want_q[rows[:, None], qa[None, :]] = p[mask]
  want_v[rows[:, None], va[None, :]] = v[mask]
  got_q, got_v = dq.cpu(), dv.cpu()
  assert torch.equal(got_q[~mask], qpos[~mask]) and torch.equal(got_v[~mask], qvel[~mask])
  others = torch.ones(36, dtype=torch.bool)
  others[qa] = False
  assert torch.equal(got_q[:, others], qpos[:, others])
  assert (got_q.double() - want_q).abs().max() <= TOL and (got_v.double() - want_v).abs().max() <= TOL
  clipped = ((p == lim[:, sel, 0].double()) | (p == lim[:, sel, 1].double()))[mask].sum()
  assert clipped > 0  # the soft limits acted somewhere


def test_push_by_setting_velocity():
  from mjlab_amd import env_terms

  dev = _dev()
  g, _, qvel, _ = _setup(3)
  time_left = torch.rand(N, generator=g) * 0.1  # dt 0.02: about a fifth of the worlds trigger
  dt = 0.02
  interval = torch.tensor([1.0, 3.0])
  vel_w = torch.randn((N, 6), generator=g)
  xquat = _unit_quats(N * 3, g).float().view(N, 3, 4)  # root_link_quat_w is a strided view of xquat in the reference
  U = torch.rand((N, 9), generator=g)
  rng = torch.tensor([[-0.5, -0.5, 0.0, -0.1, -0.2, -0.3], [0.5, 0.5, 0.2, 0.1, 0.2, 0.3]])
  dv, dt_left = qvel.to(dev), time_left.to(dev)
  env_terms.push_by_setting_velocity(dv, 0, dt_left, dt, interval.to(dev), vel_w.to(dev), xquat.to(dev)[:, 1], U.to(dev)[:, 2:9], rng.to(dev))
  torch.cuda.synchronize()
  t32 = time_left - torch.tensor(dt, dtype=torch.float32)
  trig = t32 < 1e-6
  u = U[:, 2:9].double()
  v = vel_w.double() + u[:, :6] * (rng[1] - rng[0]).double() + rng[0].double()
  v = torch.cat([v[:, :3], quat_apply_inverse(xquat[:, 1].double(), v[:, 3:])], -1)
  want_v = qvel.clone().double()
  want_v[trig, 0:6] = v[trig]
  want_t = torch.where(trig, u[:, 6] * 2.0 + 1.0, t32.double())
  got_v, got_t = dv.cpu(), dt_left.cpu()
  assert 50 < trig.sum() < N - 50
  assert torch.equal(got_v[~trig], qvel[~trig]) and torch.equal(got_t[~trig], t32[~trig])
  assert (got_v.double() - want_v).abs().max() <= TOL and (got_t.double() - want_t).abs().max() <= TOL


def _command_term(g, dev, heading_command=True):
  cfg = types.SimpleNamespace(init_velocity_prob=0.0, heading_command=heading_command, resampling_time_range=(3.0, 8.0), rel_heading_envs=0.7,
                              rel_standing_envs=0.2, heading_control_stiffness=0.5)
  heading_w = (torch.rand((N, 2), generator=g) * 8 - 4).to(dev)
  t = types.SimpleNamespace(cfg=cfg, num_envs=N, time_left=(torch.rand(N, generator=g) * 0.1).to(dev), vel_command_b=torch.randn((N, 3), generator=g).to(dev),
                            heading_target=(torch.rand(N, generator=g) * 6.28 - 3.14).to(dev), is_heading_env=(torch.rand(N, generator=g) < 0.5).to(dev),
                            is_standing_env=(torch.rand(N, generator=g) < 0.3).to(dev), command_counter=torch.randint(0, 5, (N,), generator=g).to(dev),
                            robot=types.SimpleNamespace(data=types.SimpleNamespace(heading_w=heading_w[:, 1])))
  return t


def _snapshot(t):
  return {k: getattr(t, k).cpu().clone() for k in ("time_left", "vel_command_b", "heading_target", "is_heading_env", "is_standing_env", "command_counter")}


def _resampled(s0, m, u, ranges, cfg):
  """The reference's _resample + _resample_command for the rows of m, fp64."""
  out = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in s0.items()}
  lo, hi = ranges[:, 0].double(), ranges[:, 1].double()
  out["time_left"][m] = (u[:, 0] * (cfg.resampling_time_range[1] - cfg.resampling_time_range[0]) + cfg.resampling_time_range[0])[m]
  out["vel_command_b"][m] = (u[:, 1:4] * (hi[:3] - lo[:3]) + lo[:3])[m]
  if cfg.heading_command:
    out["heading_target"][m] = (u[:, 4] * (hi[3] - lo[3]) + lo[3])[m]
    out["is_heading_env"][m] = (u[:, 5].float() <= cfg.rel_heading_envs)[m]
  out["is_standing_env"][m] = (u[:, 6].float() <= torch.tensor(cfg.rel_standing_envs, dtype=torch.float32))[m]
  out["command_counter"][m] += 1
  return out


def _check(got, want, m, s0):
  for k in ("is_heading_env", "is_standing_env", "command_counter"):
    assert torch.equal(got[k], want[k]), k
  for k in ("time_left", "vel_command_b", "heading_target"):
    assert (got[k].double() - want[k]).abs().max() <= TOL, (k, (got[k].double() - want[k]).abs().max())


@pytest.mark.parametrize("heading_command", [True, False])
def test_command_uniform_velocity_reset(heading_command):
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(4)
  t = _command_term(g, dev, heading_command)
  s0 = _snapshot(t)
  mask = torch.rand(N, generator=g) < 0.4
  U = torch.rand((N, 8), generator=g)
  ranges = torch.tensor([[-1.0, 1.0], [-0.5, 0.5], [-0.7, 0.7], [-3.14, 3.14]])
  env_terms.command_uniform_velocity(t, mask.to(dev), U.to(dev), ranges.to(dev), 0.02)
  torch.cuda.synchronize()
  got = _snapshot(t)
  want = _resampled(s0, mask, U.double(), ranges, t.cfg)
  _check(got, want, mask, s0)
  for k, v in s0.items():  # rows outside the mask: untouched, bit for bit
    assert torch.equal(got[k][~mask], v[~mask]), k


@pytest.mark.parametrize("heading_command", [True, False])
def test_command_uniform_velocity_compute(heading_command):
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(5)
  t = _command_term(g, dev, heading_command)
  s0 = _snapshot(t)
  U = torch.rand((N, 8), generator=g)
  ranges = torch.tensor([[-1.0, 1.0], [-0.5, 0.5], [-0.7, 0.7], [-3.14, 3.14]])
  dt = 0.02
  env_terms.command_uniform_velocity(t, None, U.to(dev), ranges.to(dev), dt)
  torch.cuda.synchronize()
  got = _snapshot(t)
  t32 = s0["time_left"] - torch.tensor(dt, dtype=torch.float32)
  m = t32 <= 0
  assert 50 < m.sum() < N - 50
  s1 = dict(s0)
  s1["time_left"] = t32
  want = _resampled(s1, m, U.double(), ranges, t.cfg)
  # _update_command (velocity_command.py:89-102) on every row
  if heading_command:
    hw = t.robot.data.heading_w.cpu().double()
    err = wrap_to_pi(want["heading_target"].double() - hw)
    yaw = torch.clamp(0.5 * err, ranges[2, 0].double(), ranges[2, 1].double())
    want["vel_command_b"][:, 2] = torch.where(want["is_heading_env"], yaw, want["vel_command_b"][:, 2])
  want["vel_command_b"][want["is_standing_env"]] = 0.0
  _check(got, want, m, s0)
  assert torch.equal(got["time_left"][~m], t32[~m])


def test_command_uniform_velocity_compute_with_its_metrics():
  """``metrics=True``: ``_update_metrics`` (reference tasks/velocity/mdp/velocity_command.py:50-62) at the head of the compute launch, on the
  command as it stood BEFORE the resample / update -- against the reference's lines in torch on the same device (1 ulp: the norm's reduction),
  the velocities given as row views of a wider buffer; everything else as without the metrics."""
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(6)
  ta, tb = _command_term(g, dev), None
  g = torch.Generator().manual_seed(6)
  tb = _command_term(g, dev)
  vel = torch.randn((N, 9), generator=g).to(dev)
  m0 = {"error_vel_xy": torch.rand(N, generator=g).to(dev), "error_vel_yaw": torch.rand(N, generator=g).to(dev)}
  for t in (ta, tb):
    t.metrics = {k: v.clone() for k, v in m0.items()}
    t.robot.data.root_link_lin_vel_b, t.robot.data.root_link_ang_vel_b = vel[:, 1:4], vel[:, 5:8]
    t._env = types.SimpleNamespace(step_dt=0.02)
  U = torch.rand((N, 8), generator=g).to(dev)
  ranges = torch.tensor([[-1.0, 1.0], [-0.5, 0.5], [-0.7, 0.7], [-3.14, 3.14]]).to(dev)
  cmd0 = ta.vel_command_b.clone()
  env_terms.command_uniform_velocity(ta, None, U, ranges, 0.02, metrics=True)
  env_terms.command_uniform_velocity(tb, None, U, ranges, 0.02)
  torch.cuda.synchronize()
  for k, v in _snapshot(ta).items():
    assert torch.equal(v, _snapshot(tb)[k]), k
  assert torch.equal(tb.metrics["error_vel_xy"], m0["error_vel_xy"])  # (without the switch the launch leaves them alone)
  max_command_step = ta.cfg.resampling_time_range[1] / 0.02
  want_xy = m0["error_vel_xy"] + torch.norm(cmd0[:, :2] - vel[:, 1:3], dim=-1) / max_command_step
  want_yaw = m0["error_vel_yaw"] + torch.abs(cmd0[:, 2] - vel[:, 7]) / max_command_step
  assert float((ta.metrics["error_vel_xy"] - want_xy).abs().max()) <= 1.2e-7 * float(want_xy.abs().max())
  assert torch.equal(ta.metrics["error_vel_yaw"], want_yaw)


def test_terms_refuse_what_they_cannot_address():
  from mjlab_amd import env_terms

  dev = _dev()
  g, qpos, qvel, mask = _setup(6)
  with pytest.raises(TypeError):
    env_terms.reset_root_state_uniform(qpos.to(dev).double(), qvel.to(dev), 0, 0, mask.to(dev), torch.zeros((N, 13), device=dev), torch.zeros((N, 3), device=dev),
                                       torch.zeros((N, 12), device=dev), torch.zeros((2, 6), device=dev), torch.zeros((2, 6), device=dev))
  with pytest.raises(RuntimeError, match="bad sizes"):
    env_terms.reset_root_state_uniform(qpos.to(dev), qvel.to(dev), 33, 0, mask.to(dev), torch.zeros((N, 13), device=dev), torch.zeros((N, 3), device=dev),
                                       torch.zeros((N, 12), device=dev), torch.zeros((2, 6), device=dev), torch.zeros((2, 6), device=dev))


def test_motion_file_from_the_hip_forward_kinematics_equals_the_oracle(tmp_path):
  """mjlab_amd.rollout.write_motion_npz (one world per frame through Simulation.forward) against the same file built from the CPU
  oracle's kinematics (tests/_motion_fixture.py): keys, shapes, and every body's pose / velocity to fp32 rounding."""
  import sys
  from pathlib import Path

  import numpy as np

  sys.path.insert(0, str(Path(__file__).resolve().parent))
  from _motion_fixture import write_full_motion

  from mjlab_amd import robots
  from mjlab_amd.rollout import write_motion_npz

  model = robots.load_model("g1_tracking_flat")
  shape = write_motion_npz(str(tmp_path / "hip.npz"), model, "cuda:0")
  assert tuple(write_full_motion(str(tmp_path / "oracle.npz"))) == shape
  a, b = np.load(tmp_path / "hip.npz"), np.load(tmp_path / "oracle.npz")
  assert sorted(a.files) == sorted(b.files)
  for k in a.files:
    assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
    assert np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() <= 5e-6, (k, np.abs(a[k] - b[k]).max())


# ---------------------------------------------------------------------------------------------------- MotionCommand (tracking task)
def quat_apply(q, v):
  xyz = q[..., 1:]
  t = torch.cross(xyz, v, dim=-1) * 2
  return v + q[..., 0:1] * t + torch.cross(xyz, t, dim=-1)


def _motion_term(g, dev, nframe=40, nbody_m=9, nj=29):
  """Synthetic motion tables shaped like the reference's MotionLoader holds them (tasks/tracking/mdp/commands.py:28-50)."""
  idx = torch.tensor([0, 3, 4, 7, 8])  # tracked bodies on the tables' body axis; the first is the floating base
  mo = types.SimpleNamespace(joint_pos=torch.randn((nframe, nj), generator=g).to(dev), joint_vel=torch.randn((nframe, nj), generator=g).to(dev),
                             _body_pos_w=torch.randn((nframe, nbody_m, 3), generator=g).to(dev), _body_quat_w=_unit_quats(nframe * nbody_m, g).float().view(nframe, nbody_m, 4).to(dev),
                             _body_lin_vel_w=torch.randn((nframe, nbody_m, 3), generator=g).to(dev), _body_ang_vel_w=torch.randn((nframe, nbody_m, 3), generator=g).to(dev),
                             _body_indexes=idx.to(dev))
  return types.SimpleNamespace(motion=mo), idx


def test_command_motion_write():
  from mjlab_amd import env_terms

  dev = _dev()
  g, qpos, qvel, mask = _setup(7)
  term, idx = _motion_term(g, dev)
  mo, nj = term.motion, 29
  tab, keep = env_terms.motion_tables(term)
  ts = torch.randint(0, 40, (N,), generator=g)
  org = torch.randn((N, 3), generator=g) * 3
  lim = torch.stack([-0.8 - torch.rand(nj, generator=g), 0.8 + torch.rand(nj, generator=g)], -1)[None].expand(N, nj, 2)
  U = torch.rand((N, 15 + nj), generator=g)
  pose = torch.tensor([[-0.05, -0.05, -0.01, -0.1, -0.1, -0.2], [0.05, 0.05, 0.01, 0.1, 0.1, 0.2]])
  vel = torch.tensor([[-0.5, -0.5, -0.2, -0.52, -0.52, -0.78], [0.5, 0.5, 0.2, 0.52, 0.52, 0.78]])
  jr = (-0.1, 0.1)
  qa, va = 7 + torch.arange(nj), 6 + torch.arange(nj)
  dq, dv = qpos.to(dev), qvel.to(dev)
  env_terms.command_motion_write(tab, dq, dv, 0, 0, qa.to(dev, torch.int32), va.to(dev, torch.int32), mask.to(dev), ts.to(dev), org.to(dev),
                                 lim[:1].to(dev).expand(N, nj, 2), U.to(dev)[:, 3:], pose.to(dev), vel.to(dev), jr)
  torch.cuda.synchronize()
  # MotionCommand._resample_command (:305-363) in fp64 on the same inputs
  u = U[:, 3:].double()
  b0 = int(idx[0])
  rs = u[:, :6] * (pose[1] - pose[0]).double() + pose[0].double()
  root_pos = mo._body_pos_w.cpu().double()[ts, b0] + org.double() + rs[:, :3]
  root_ori = quat_mul(quat_from_euler_xyz(rs[:, 3], rs[:, 4], rs[:, 5]), mo._body_quat_w.cpu().double()[ts, b0])
  rv = u[:, 6:12] * (vel[1] - vel[0]).double() + vel[0].double()
  lin = mo._body_lin_vel_w.cpu().double()[ts, b0] + rv[:, :3]
  ang = quat_apply_inverse(root_ori, mo._body_ang_vel_w.cpu().double()[ts, b0] + rv[:, 3:])
  jp = mo.joint_pos.cpu().double()[ts] + (u[:, 12:] * (jr[1] - jr[0]) + jr[0])
  jp = torch.minimum(torch.maximum(jp, lim[..., 0].double()), lim[..., 1].double())
  want_q, want_v = qpos.clone().double(), qvel.clone().double()
  want_q[mask] = torch.cat([root_pos, root_ori, jp], -1)[mask]
  want_v[mask] = torch.cat([lin, ang, mo.joint_vel.cpu().double()[ts]], -1)[mask]
  got_q, got_v = dq.cpu(), dv.cpu()
  assert torch.equal(got_q[~mask], qpos[~mask]) and torch.equal(got_v[~mask], qvel[~mask])
  assert (got_q.double() - want_q).abs().max() <= 3 * TOL and (got_v.double() - want_v).abs().max() <= 3 * TOL
  assert ((jp == lim[..., 0].double()) | (jp == lim[..., 1].double()))[mask].sum() > 0  # the soft limits acted somewhere
  del keep


def test_command_motion_relative():
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(8)
  term, idx = _motion_term(g, dev)
  mo, nb, nbody = term.motion, len(idx), 12
  tab, keep = env_terms.motion_tables(term)
  ts = torch.randint(0, 40, (N,), generator=g)
  org = torch.randn((N, 3), generator=g) * 3
  xpos = torch.randn((N, nbody, 3), generator=g)
  xquat = _unit_quats(N * nbody, g).float().view(N, nbody, 4)
  anchor_gid, anchor_index = 5, 2
  out_p, out_q = torch.zeros((N, nb, 3), device=dev), torch.zeros((N, nb, 4), device=dev)
  env_terms.command_motion_relative(tab, ts.to(dev), org.to(dev), xpos.to(dev), xquat.to(dev), anchor_gid, anchor_index, out_p, out_q)
  torch.cuda.synchronize()
  # MotionCommand._update_command (:371-392) in fp64
  bp = mo._body_pos_w.cpu().double()[ts][:, idx] + org.double()[:, None, :]
  bq = mo._body_quat_w.cpu().double()[ts][:, idx]
  apos, aquat = bp[:, anchor_index], bq[:, anchor_index]
  rp, rq = xpos.double()[:, anchor_gid], xquat.double()[:, anchor_gid]
  inv = torch.cat([aquat[:, :1], -aquat[:, 1:]], -1) / aquat.pow(2).sum(-1, keepdim=True).clamp(min=1e-9)
  d = quat_mul(rq, inv)
  yaw = torch.atan2(2 * (d[:, 0] * d[:, 3] + d[:, 1] * d[:, 2]), 1 - 2 * (d[:, 2] ** 2 + d[:, 3] ** 2))
  dq = torch.stack([torch.cos(yaw / 2), torch.zeros_like(yaw), torch.zeros_like(yaw), torch.sin(yaw / 2)], -1)
  dq = dq / dq.norm(dim=-1, keepdim=True)
  dqr = dq[:, None, :].expand(N, nb, 4)
  want_q = quat_mul(dqr, bq)
  delta = rp.clone()
  delta[:, 2] = apos[:, 2]
  want_p = delta[:, None, :] + quat_apply(dqr, bp - apos[:, None, :])
  assert (out_q.cpu().double() - want_q).abs().max() <= TOL, (out_q.cpu().double() - want_q).abs().max()
  assert (out_p.cpu().double() - want_p).abs().max() <= 5 * TOL, (out_p.cpu().double() - want_p).abs().max()
  del keep


def test_command_motion_frame_equals_the_reference_gathers_bit_for_bit():
  """mjlab_command_motion_frame (env_terms.MotionFrame) against the expressions of MotionCommand's properties (tasks/tracking/mdp/
  commands.py:128-215) in torch on the same device: copies and one addition -- exact equality."""
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(21)
  term, idx = _motion_term(g, dev)
  mo, nb, nbody_e = term.motion, len(idx), 30
  tab, keep = env_terms.motion_tables(term)
  ts = torch.randint(0, 40, (N,), generator=g).to(dev)
  org = (torch.randn((N, 3), generator=g) * 3).to(dev)
  pose = torch.randn((N, nbody_e, 7), generator=g).to(dev)
  vel = torch.randn((N, nbody_e, 6), generator=g).to(dev)
  body_indexes = torch.tensor([0, 5, 11, 17, 29]).to(dev)
  term.time_steps, term.num_envs, term.body_indexes, term.motion_anchor_body_index = ts, N, body_indexes, 2
  term.robot = types.SimpleNamespace(data=types.SimpleNamespace(body_link_pose_w=pose, body_link_vel_w=vel))
  got = env_terms.MotionFrame(term, tab).update(org)
  torch.cuda.synchronize()
  body = lambda table: table[:, idx.to(dev)]  # MotionLoader.body_*_w (:52-65)  # noqa: E731
  want = {
    "joint_pos": mo.joint_pos[ts], "joint_vel": mo.joint_vel[ts],
    "body_pos_w": body(mo._body_pos_w)[ts] + org[:, None, :], "body_quat_w": body(mo._body_quat_w)[ts],
    "body_lin_vel_w": body(mo._body_lin_vel_w)[ts], "body_ang_vel_w": body(mo._body_ang_vel_w)[ts],
    "anchor_pos_w": body(mo._body_pos_w)[ts, 2] + org, "anchor_quat_w": body(mo._body_quat_w)[ts, 2],
    "anchor_lin_vel_w": body(mo._body_lin_vel_w)[ts, 2], "anchor_ang_vel_w": body(mo._body_ang_vel_w)[ts, 2],
    "robot_body_pos_w": pose[..., :3][:, body_indexes], "robot_body_quat_w": pose[..., 3:7][:, body_indexes],
    "robot_body_lin_vel_w": vel[..., :3][:, body_indexes], "robot_body_ang_vel_w": vel[..., 3:6][:, body_indexes],
  }
  assert set(got) == set(want) == set(env_terms.MotionFrame.NAMES)
  for k, v in want.items():
    assert got[k].shape == v.shape and torch.equal(got[k], v), k
  del keep


def test_command_motion_metrics_equal_the_reference_formulas():
  """mjlab_command_motion_metrics (env_terms.MotionMetrics) against MotionCommand._update_metrics (reference tasks/tracking/mdp/commands.py:
  221-254; quat_error_magnitude = math.py:682-693 over quat_box_minus :584-598 and axis_angle_from_quat :472-500) restated in float64 on
  the same inputs: 1e-6 relative (logging quantities -- a few float32 ulp from the reference's torch reductions), small angles (the Taylor
  branch), identical quaternions and a robot joint array that is a row view included."""
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(77)
  nb, nj, a = 14, 29, 3
  r = lambda *shape: torch.randn(shape, generator=g)  # noqa: E731
  uq = lambda *shape: torch.nn.functional.normalize(r(*shape, 4), dim=-1)  # noqa: E731
  t = types.SimpleNamespace(num_envs=N, metrics={"error_anchor_pos": torch.full((N,), 3.0).to(dev), "sampling_entropy": torch.zeros(N).to(dev)},
                            time_steps=torch.zeros(N, dtype=torch.long, device=dev), cfg=types.SimpleNamespace(body_names=["b"] * nb), motion_anchor_body_index=a)
  f = {"body_pos_w": r(N, nb, 3), "body_quat_w": uq(N, nb), "body_lin_vel_w": r(N, nb, 3), "body_ang_vel_w": r(N, nb, 3), "robot_body_pos_w": r(N, nb, 3),
       "robot_body_quat_w": uq(N, nb), "robot_body_lin_vel_w": r(N, nb, 3), "robot_body_ang_vel_w": r(N, nb, 3), "body_pos_relative_w": r(N, nb, 3),
       "body_quat_relative_w": uq(N, nb), "joint_pos": r(N, nj), "joint_vel": r(N, nj)}
  f["robot_body_quat_w"][0] = f["body_quat_relative_w"][0]  # zero error: the Taylor branch of axis_angle_from_quat
  f["robot_body_quat_w"][1] = torch.nn.functional.normalize(f["body_quat_relative_w"][1] + 1e-7 * r(nb, 4), dim=-1)
  f["robot_body_quat_w"][2] = -f["body_quat_relative_w"][2]  # the same rotation, the other sign
  state = r(N, 2 * nj + 5)
  for k, v in f.items():
    setattr(t, k, v.to(dev))
  sd = state.to(dev)
  t.robot_joint_pos, t.robot_joint_vel = sd[:, 2:2 + nj], sd[:, 2 + nj:2 + 2 * nj]  # row views, as EntityData hands them out
  mm = env_terms.MotionMetrics(t)
  assert float(t.metrics["error_anchor_pos"][0]) == 3.0 and set(env_terms.MotionMetrics.KEYS) <= set(t.metrics) and "sampling_entropy" in t.metrics
  rows = [t.metrics[k] for k in env_terms.MotionMetrics.KEYS]
  mm.update()
  torch.cuda.synchronize()
  assert all(t.metrics[k] is row for k, row in zip(env_terms.MotionMetrics.KEYS, rows, strict=True))  # the entries are bound once

  d = {k: v.double() for k, v in f.items()}
  rjp, rjv = state[:, 2:2 + nj].double(), state[:, 2 + nj:2 + 2 * nj].double()

  def qmul(p, q):
    w1, x1, y1, z1 = p.unbind(-1)
    w2, x2, y2, z2 = q.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)

  def qerr(q1, q2):
    q = qmul(q1, q2 * torch.tensor([1.0, -1.0, -1.0, -1.0], dtype=torch.float64))
    q = q * (1.0 - 2.0 * (q[..., 0:1] < 0.0))
    mag = q[..., 1:].norm(dim=-1)
    half = torch.atan2(mag, q[..., 0])
    ang = 2.0 * half
    k = torch.where(ang.abs() > 1e-6, torch.sin(half) / ang, 0.5 - ang * ang / 48)
    return (q[..., 1:4] / k.unsqueeze(-1)).norm(dim=-1)

  want = {
    "error_anchor_pos": (d["body_pos_w"][:, a] - d["robot_body_pos_w"][:, a]).norm(dim=-1), "error_anchor_rot": qerr(d["body_quat_w"][:, a], d["robot_body_quat_w"][:, a]),
    "error_anchor_lin_vel": (d["body_lin_vel_w"][:, a] - d["robot_body_lin_vel_w"][:, a]).norm(dim=-1),
    "error_anchor_ang_vel": (d["body_ang_vel_w"][:, a] - d["robot_body_ang_vel_w"][:, a]).norm(dim=-1),
    "error_body_pos": (d["body_pos_relative_w"] - d["robot_body_pos_w"]).norm(dim=-1).mean(dim=-1), "error_body_rot": qerr(d["body_quat_relative_w"], d["robot_body_quat_w"]).mean(dim=-1),
    "error_body_lin_vel": (d["body_lin_vel_w"] - d["robot_body_lin_vel_w"]).norm(dim=-1).mean(dim=-1),
    "error_body_ang_vel": (d["body_ang_vel_w"] - d["robot_body_ang_vel_w"]).norm(dim=-1).mean(dim=-1),
    "error_joint_pos": (d["joint_pos"] - rjp).norm(dim=-1), "error_joint_vel": (d["joint_vel"] - rjv).norm(dim=-1),
  }
  for k, v in want.items():
    got = t.metrics[k].double().cpu()
    # identical / sign-flipped / 1e-7-perturbed quaternions: the float32 product carries ~1e-7 of rounding, which IS the angle there (the
    # reference's float32 chain has the same floor) -- absolute 1e-6 rad for those rows, 1e-6 relative elsewhere
    assert bool(((got - v).abs() <= 1e-6 * v.abs() + 1e-6).all()), (k, float((got - v).abs().max()))
  assert float(want["error_body_rot"][3:].min()) > 0.5  # (random orientations: the ordinary branch is what the bulk exercises)


@pytest.mark.parametrize("sharded", [False, True])
@pytest.mark.parametrize("case", ["some", "nobody", "masked_but_nobody_failed"])
def test_command_motion_sample_equals_the_torch_restatement_bit_for_bit(sharded, case):
  """mjlab_command_motion_sample against the expressions of GraphedRlEnv._resample_MotionCommand's torch path (the mask-based form of
  MotionCommand._adaptive_sampling, reference tasks/tracking/mdp/commands.py:256-297) on the same device and the same cdf: the failure
  histogram, the new phases and the sampling metrics -- exact equality; an empty mask leaves everything as it was."""
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(5)
  n, nbin, total = 4100, 23, 1117
  p = torch.rand(nbin, generator=g) + 0.05
  p = (p / p.sum()).to(dev)
  cdf = torch.cumsum(p, 0)
  H, pmax, top = (-(p * (p + 1e-12).log()).sum() / math.log(nbin)), p.max(), (p.argmax().float() / nbin)
  mask = (torch.rand(n, generator=g) < 0.3).to(dev) if case != "nobody" else torch.zeros(n, dtype=torch.bool, device=dev)
  terminated = (torch.rand(n, generator=g) < 0.5).to(dev) if case == "some" else torch.zeros(n, dtype=torch.bool, device=dev)
  ts0 = torch.randint(0, total, (n,), generator=g).to(dev)
  U = torch.rand((n, 40), generator=g).to(dev)[:, 3:30]  # a column slice of the step's block of uniforms
  U[5, 1] = 0.0
  U[6, 1] = 0.99999994  # beyond the last cdf entry where the running sum falls short of 1: clamped to the last bin
  U[7, 1] = float(cdf[3])  # exactly on an edge: searchsorted's "first index with cdf >= u"
  metrics0 = {k: torch.rand(n, generator=g).to(dev) for k in ("sampling_entropy", "sampling_top1_prob", "sampling_top1_bin")}
  cur0 = torch.rand(nbin, generator=g).to(dev)
  tl0, cc0 = (torch.rand(n, generator=g) * 3).to(dev), torch.randint(0, 9, (n,), generator=g).to(dev)
  term = types.SimpleNamespace(time_steps=ts0.clone(), metrics={k: v.clone() for k, v in metrics0.items()}, bin_count=nbin,
                               motion=types.SimpleNamespace(time_step_total=total), _current_bin_failed=cur0.clone(), time_left=tl0.clone(), command_counter=cc0.clone())
  row = torch.full((nbin + 1,), -1.0, device=dev)
  timer = (0.37, 4.1) if sharded else None  # CommandTerm._resample's timer and counter ride along (the reset-phase call) or not (the update-phase call)
  env_terms.command_motion_sample(term, mask, terminated, U, cdf, H, pmax, top, row[:nbin] if sharded else term._current_bin_failed, row[nbin:] if sharded else None, timer)
  torch.cuda.synchronize()
  if timer is None:
    assert torch.equal(term.time_left, tl0) and torch.equal(term.command_counter, cc0)
  else:  # managers/command_manager.py:62-66 as GraphedRlEnv._command_resample writes it
    assert torch.equal(term.time_left, torch.where(mask, U[:, 0] * (timer[1] - timer[0]) + timer[0], tl0))
    assert torch.equal(term.command_counter, cc0 + mask.to(cc0.dtype))
  # the torch path (mjlab_amd/graphed_env.py, _resample_MotionCommand)
  failed = terminated & mask
  bins = torch.clamp((ts0 * nbin) // max(total, 1), 0, nbin - 1)
  counts = torch.zeros(nbin, device=dev).scatter_add_(0, bins, failed.to(torch.float32))
  sampled = torch.searchsorted(cdf, U[:, 1].contiguous()).clamp_(max=nbin - 1)
  t_new = ((sampled + U[:, 2]) / nbin * (total - 1)).long()
  assert torch.equal(term.time_steps, torch.where(mask, t_new, ts0))
  if sharded:
    assert torch.equal(row[:nbin], counts) and float(row[nbin]) == float(failed.any()) and torch.equal(term._current_bin_failed, cur0)
  else:
    assert torch.equal(term._current_bin_failed, counts if bool(failed.any()) else cur0)
  for k, val in (("sampling_entropy", H), ("sampling_top1_prob", pmax), ("sampling_top1_bin", top)):
    assert torch.equal(term.metrics[k], val.expand(n) if bool(mask.any()) else metrics0[k]), k
  if case == "some":
    assert int(failed.sum()) > 100 and int(counts.sum()) == int(failed.sum()) and int(sampled[6]) == nbin - 1 and int(sampled[7]) == 3


@pytest.mark.parametrize("ksize", [1, 3])
def test_command_motion_sampler_update_and_distribution(ksize):
  """mjlab_command_motion_sampler (env_terms.MotionSampler) against the reference's lines (tasks/tracking/mdp/commands.py:267-281, 291-294,
  394-398) in torch on the same device: the statistics' update bit for bit (elementwise), the distribution -- cdf, entropy, top bin -- to
  float rounding (another summation order)."""
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(31)
  nbin, alpha, ratio, lam = 37, 0.001, 0.1, 0.8
  kernel = torch.tensor([lam**i for i in range(ksize)])
  kernel = (kernel / kernel.sum()).to(dev)
  bfc0, cur0 = (torch.rand(nbin, generator=g) * 0.05).to(dev), torch.randint(0, 5, (nbin,), generator=g).float().to(dev)
  term = types.SimpleNamespace(bin_failed_count=bfc0.clone(), _current_bin_failed=cur0.clone(), bin_count=nbin, kernel=kernel,
                               cfg=types.SimpleNamespace(adaptive_alpha=alpha, adaptive_uniform_ratio=ratio, adaptive_kernel_size=ksize))
  ms = env_terms.MotionSampler(term)
  ms.update()
  torch.cuda.synchronize()
  assert torch.equal(term.bin_failed_count, alpha * cur0 + (1 - alpha) * bfc0) and not bool(term._current_bin_failed.any())
  cdf, H, pmax, top = ms.distribution()
  torch.cuda.synchronize()
  p = term.bin_failed_count + ratio / float(nbin)
  p = torch.nn.functional.pad(p.unsqueeze(0).unsqueeze(0), (0, ksize - 1), mode="replicate")
  p = torch.nn.functional.conv1d(p, kernel.view(1, 1, -1)).view(-1)
  p = p / p.sum()
  Hw = -(p * (p + 1e-12).log()).sum() / math.log(nbin)
  pm, im = p.max(dim=0)
  assert float((cdf - torch.cumsum(p, 0)).abs().max()) <= 5e-7 and abs(float(cdf[-1]) - 1.0) <= 1e-6
  assert abs(float(H) - float(Hw)) <= 2e-6 and abs(float(pmax) - float(pm)) <= 1e-7 * float(pm) + 1e-9 and float(top) == float(im.float() / nbin)


def test_log_book_publishes_the_same_numbers_through_the_one_launch():
  """env_core.LogBook with the count at hand on the GPU (``mjlab_log_finish``) against the same book's torch lines: every kind of entry
  (episode reward sums / command metrics / termination counts / curriculum state), steps with and without resets -- bit for bit, and the
  entries handed out stay the same tensors."""
  from mjlab_amd import env_core

  dev = _dev()
  g = torch.Generator().manual_seed(12)
  kinds = ["sum_len"] * 5 + ["sum"] * 4 + ["count"] * 3 + ["state"]
  books = [env_core.LogBook(20.0, dev), env_core.LogBook(20.0, dev)]
  src = torch.zeros(len(kinds) + 1, device=dev)  # (persistent: what the masked sums' output is)
  handed = None
  for step in range(8):
    nreset = 0 if step in (2, 3, 6) else int(torch.randint(1, 40, (1,), generator=g))
    mask = torch.zeros(N, dtype=torch.bool, device=dev)
    mask[torch.randperm(N, generator=g)[:nreset].to(dev)] = True
    src.copy_(torch.cat([torch.randn(len(kinds), generator=g) * 30, torch.tensor([float(nreset)])]).to(dev))
    log = {f"k{i}": (src[i], kd) for i, kd in enumerate(kinds)}
    a = books[0].publish(dict(log), mask, src[-1])  # the launch (after the first publication, which allocates through the torch lines)
    b = books[1].publish(dict(log), mask)  # the torch lines
    torch.cuda.synchronize()
    assert list(a) == list(b)
    for k in a:
      assert torch.equal(a[k], b[k]), (step, k, float(a[k]), float(b[k]))
    if handed is None:
      handed = {k: v.data_ptr() for k, v in a.items()}
    assert {k: v.data_ptr() for k, v in a.items()} == handed
  assert books[0]._ptrs is not None and books[1]._ptrs is None  # (the launch was what ran on the first book)


def test_copy_batch_copies_every_pair():
  """mjlab_copy_batch: 40 pairs (two launches) of float, int64 and bool tensors, odd byte counts included; strided / mixed-dtype pairs
  keep copy_."""
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(3)
  pairs = []
  for k in range(40):
    shape = (N, 1 + k % 7) if k % 3 else (N,)
    if k % 5 == 0:
      src = torch.rand(shape, generator=g) > 0.5
    elif k % 5 == 1:
      src = torch.randint(-9, 9, shape, generator=g)
    else:
      src = torch.randn(shape, generator=g)
    pairs.append((torch.zeros_like(src).to(dev), src.to(dev)))
  odd = (torch.zeros(7, dtype=torch.bool, device=dev), torch.tensor([1, 0, 1, 1, 0, 0, 1], dtype=torch.bool, device=dev))
  strided = (torch.zeros((N, 2), device=dev), torch.randn((N, 4), generator=g).to(dev)[:, ::2])
  cast = (torch.zeros(N, device=dev), torch.arange(N, device=dev))
  env_terms.copy_batch(pairs + [odd, strided, cast])
  torch.cuda.synchronize()
  for dst, src in pairs + [odd, strided]:
    assert torch.equal(dst, src)
  assert torch.equal(cast[0], cast[1].float())


def test_reward_accumulate_equals_the_managers_loop_bit_for_bit():
  """mjlab_reward_accumulate against the reference loop's torch operations (managers/reward_manager.py:77-89) on the same raw term
  values: reward, episode sums and per-step term values, bitwise."""
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(11)
  names = ["a", "b", "zero", "c", "d"]
  weights = [1.0, -0.1, 0.0, 2.5, -1.0e-3]
  raws = [torch.randn(N, generator=g).to(dev) * s for s in (1.0, 30.0, 1.0, 1e-3, 500.0)]
  dt = 0.02

  def manager():
    cfgs = [types.SimpleNamespace(weight=w, params={}, func=(lambda env, _r=r: _r)) for w, r in zip(weights, raws, strict=True)]
    return types.SimpleNamespace(_term_names=list(names), _term_cfgs=cfgs, _env=None, _reward_buf=torch.full((N,), 7.0, device=dev),
                                 _episode_sums={n: torch.randn(N, generator=g).to(dev) for n in names}, _step_reward=torch.full((N, len(names)), 3.0, device=dev))

  m1, m2 = manager(), manager()
  for n in names:
    m2._episode_sums[n].copy_(m1._episode_sums[n])
  out = env_terms.RewardAccumulator(m1).compute(dt)
  # the reference's loop, operation by operation
  m2._reward_buf[:] = 0.0
  for i, (n, cfg) in enumerate(zip(m2._term_names, m2._term_cfgs, strict=True)):
    if cfg.weight == 0.0:
      m2._step_reward[:, i] = 0.0
      continue
    value = cfg.func(None) * cfg.weight * dt
    m2._reward_buf += value
    m2._episode_sums[n] += value
    m2._step_reward[:, i] = value / dt
  torch.cuda.synchronize()
  assert out is m1._reward_buf and torch.equal(m1._reward_buf, m2._reward_buf)
  assert torch.equal(m1._step_reward, m2._step_reward)
  for n in names:
    assert torch.equal(m1._episode_sums[n], m2._episode_sums[n]), n


@pytest.mark.parametrize("fill", [False, True])
def test_masks_that_select_nobody_or_everybody(fill):
  """All-false mask: nothing is written; all-true mask: every row is written (reset events and the velocity command)."""
  from mjlab_amd import env_terms

  dev = _dev()
  g, qpos, qvel, _ = _setup(12)
  mask = torch.full((N,), fill, dtype=torch.bool, device=dev)
  dq, dv = qpos.to(dev), qvel.to(dev)
  root = torch.cat([torch.zeros(1, 3), torch.tensor([[1.0, 0, 0, 0]]), torch.zeros(1, 6)], -1).to(dev).expand(N, 13)
  U = torch.rand((N, 12 + 58), generator=g).to(dev)
  env_terms.reset_root_state_uniform(dq, dv, 0, 0, mask, root, torch.zeros((N, 3), device=dev), U[:, :12],
                                     torch.tensor([[-0.5] * 6, [0.5] * 6], device=dev), torch.tensor([[-0.1] * 6, [0.1] * 6], device=dev))
  qa, va = (7 + torch.arange(29)).to(dev, torch.int32), (6 + torch.arange(29)).to(dev, torch.int32)
  env_terms.reset_joints_by_scale(dq, dv, mask, None, qa, va, torch.ones((1, 29), device=dev).expand(N, 29), torch.ones((1, 29), device=dev).expand(N, 29),
                                  torch.tensor([-10.0, 10.0], device=dev).repeat(29).view(1, 29, 2).expand(N, 29, 2), U[:, 12:], torch.tensor([0.5, 1.5, -1.0, 1.0], device=dev))
  torch.cuda.synchronize()
  if not fill:
    assert torch.equal(dq.cpu(), qpos) and torch.equal(dv.cpu(), qvel)
  else:
    got = dq.cpu()
    assert bool((got[:, 0:3].abs() <= 0.5).all()) and bool(((got[:, 3:7].norm(dim=1) - 1).abs() < 1e-5).all())
    assert bool((got[:, 7:] >= 0.5).all()) and bool((got[:, 7:] <= 1.5).all()) and bool((dv.cpu()[:, 6:].abs() <= 1.0).all())
    assert not torch.equal(got, qpos)


def test_masked_fill_rows_and_masked_sums():
  """env_terms.MaskedFill / MaskedSums (the managers' reset() bookkeeping and logging as one launch each) against the torch operations
  they replace: fills bit for bit (float / int64 / bool buffers, whole rows and column ranges), sums to fp32 summation order."""
  from mjlab_amd import env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(13)
  mask = (torch.rand(N, generator=g) < 0.3).to(dev)
  a = torch.randn((N, 29), generator=g).to(dev)
  b = torch.randn((N, 32, 6), generator=g).to(dev)
  c = torch.randn((N, 35), generator=g).to(dev)
  cnt = torch.randint(0, 9, (N,), generator=g).to(dev)
  flag = torch.zeros(N, dtype=torch.bool, device=dev)
  v = torch.randn(N, generator=g).to(dev)
  want = [t.clone() for t in (a, b, c, cnt, flag, v)]
  fill = env_terms.MaskedFill([(a, 0.0), (b[:, 2:32], 0.0), (c[:, 0:6], 0.0), (cnt, 7), (flag, 1), (v, -1.5)])
  fill(mask)
  torch.cuda.synchronize()
  want[0][mask] = 0.0
  want[1][mask, 2:32] = 0.0
  want[2][mask, 0:6] = 0.0
  want[3][mask] = 7
  want[4][mask] = True
  want[5][mask] = -1.5
  for got, w in zip((a, b, c, cnt, flag, v), want, strict=True):
    assert torch.equal(got, w)
  x = [torch.randn(N, generator=g).to(dev) * s for s in (1.0, 100.0, 1e-3)]
  done = (torch.rand(N, generator=g) < 0.5).to(dev)
  sums = env_terms.MaskedSums([*x, done])
  out = sums(mask).cpu().double()
  ref = [float(t[mask].double().sum()) for t in x] + [float((done & mask).sum()), float(mask.sum())]
  for got, r in zip(out.tolist(), ref, strict=True):
    assert abs(got - r) <= 2e-6 * max(1.0, abs(r)) * 30, (got, r)
  assert out[-1] == float(mask.sum()) and out[-2] == float((done & mask).sum())  # counts are exact
  none = torch.zeros(N, dtype=torch.bool, device=dev)
  assert float(sums(none).abs().max()) == 0.0


def test_masked_fill_with_a_value_read_on_the_device_follows_it_through_a_captured_graph():
  """A fill whose value is a device scalar (the event manager's env-step count, managers/event_manager.py:139-148): int32 rows from an
  int64 counter (its low bytes) and int64 rows, eagerly and as a captured launch replayed after the counter moved; the torch twin
  (env_core.TorchMaskedFill) does the same; floating-point targets are refused."""
  from mjlab_amd import env_core, env_terms

  dev = _dev()
  g = torch.Generator().manual_seed(14)
  counter = torch.full((), 41, dtype=torch.long, device=dev)
  a = torch.randint(0, 9, (N,), generator=g, dtype=torch.int32).to(dev)
  b = torch.randint(0, 9, (N, 3), generator=g).to(dev)
  c = torch.randn(N, generator=g).to(dev)
  twin = [t.clone() for t in (a, b, c)]
  fill = env_terms.MaskedFill([(a, counter), (b, counter), (c, 2.5)])
  tfill = env_core.TorchMaskedFill([(twin[0], counter), (twin[1], counter), (twin[2], 2.5)])
  mask = torch.zeros(N, dtype=torch.bool, device=dev)

  def step(m):
    mask.copy_(m)
    counter.add_(1)
    fill(mask)

  stream = torch.cuda.Stream()
  stream.wait_stream(torch.cuda.current_stream())
  masks = [(torch.rand(N, generator=g) < 0.2).to(dev) for _ in range(5)]
  want = [t.clone() for t in (a, b, c)]
  with torch.cuda.stream(stream):
    step(masks[0])
  torch.cuda.current_stream().wait_stream(stream)
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph, stream=stream):
    counter.add_(1)
    fill(mask)
  # (the capture itself runs nothing: the counter still reads 42)
  want[0][masks[0]] = 42
  want[1][masks[0]] = 42
  want[2][masks[0]] = 2.5
  tfill(masks[0])
  for k, m in enumerate(masks[1:]):
    mask.copy_(m)
    graph.replay()
    torch.cuda.synchronize()
    assert int(counter) == 43 + k
    want[0][m] = 43 + k
    want[1][m] = 43 + k
    want[2][m] = 2.5
    tfill(m)
    for got, w, t in zip((a, b, c), want, twin, strict=True):
      assert torch.equal(got, w) and torch.equal(t, w), k
  with pytest.raises(TypeError):
    env_terms.MaskedFill([(c, counter)])
