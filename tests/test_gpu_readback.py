"""Fused entity read-back (mjlab_entity_readback) vs the reference's torch formulas, restated
from src/mjlab/entity/data.py:20-31,190-261,315-328,487-516 and
third_party/isaaclab/isaaclab/utils/math.py:521-662 (quat_mul, quat_apply, quat_apply_inverse)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _quat_mul(q1, q2):
  import torch

  w1, x1, y1, z1 = q1.unbind(-1)
  w2, x2, y2, z2 = q2.unbind(-1)
  return torch.stack([
    w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
    w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
    w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
    w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
  ], dim=-1)  # fmt: skip


def _quat_apply(q, v, sign=1.0):
  import torch

  xyz = q[..., 1:]
  t = torch.cross(xyz, v, dim=-1) * 2
  return v + sign * q[..., 0:1] * t + torch.cross(xyz, t, dim=-1)


def _vel_from_cvel(pos, sub, cvel):
  import torch

  return torch.cat([cvel[..., 3:6] - torch.cross(cvel[..., 0:3], sub - pos, dim=-1), cvel[..., 0:3]], dim=-1)


@pytest.mark.parametrize("name", ["g1_velocity_flat", "go1_velocity_flat"])
def test_readback_matches_reference_formulas(name):
  import torch

  from mjlab_amd import robots
  from mjlab_amd.entity_data import EntityReadback
  from mjlab_amd.rollout import PhysicsRollout
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model(name)
  sim = Simulation(512, SimulationCfg(njmax=300), model, "cuda:0")
  roll = PhysicsRollout(sim, action_scale=0.25, seed=4, min_height=0.3 if name.startswith("g1") else 0.15)
  for _ in range(10):
    roll.step(roll.random_action())
  ent = EntityReadback(sim)
  ent.update()
  torch.cuda.synchronize()
  d = sim.data
  ids = torch.from_numpy(ent.body_ids).long().cuda()
  root = int(ent.body_ids[0])
  sub = d.subtree_com[:, root].unsqueeze(1)
  pos, quat, ipos, cvel = d.xpos[:, ids], d.xquat[:, ids], d.xipos[:, ids], d.cvel[:, ids]
  iq = sim.model.body_iquat[:, ids]

  def close(a, b, tol=2e-6):
    scale = max(1.0, float(b.abs().max()))
    assert float((a - b).abs().max()) / scale < tol

  close(ent.body_link_pose_w, torch.cat([pos, quat], dim=-1), 0.0 + 1e-12)
  close(ent.body_link_vel_w, _vel_from_cvel(pos, sub, cvel))
  close(ent.body_com_pose_w, torch.cat([ipos, _quat_mul(quat, iq)], dim=-1))
  close(ent.body_com_vel_w, _vel_from_cvel(ipos, sub, cvel))
  rq = quat[:, 0]
  g = torch.tensor([0.0, 0.0, -1.0], device="cuda").expand(512, 3)
  f = torch.tensor([1.0, 0.0, 0.0], device="cuda").expand(512, 3)
  close(ent.projected_gravity_b, _quat_apply(rq, g, -1.0))
  fw = _quat_apply(rq, f, 1.0)
  close(ent.heading_w, torch.atan2(fw[:, 1], fw[:, 0]), 1e-5)
  lv = _vel_from_cvel(pos[:, 0], sub[:, 0], cvel[:, 0])
  cv = _vel_from_cvel(ipos[:, 0], sub[:, 0], cvel[:, 0])
  close(ent.root_link_lin_vel_b, _quat_apply(rq, lv[:, :3], -1.0))
  close(ent.root_link_ang_vel_b, _quat_apply(rq, lv[:, 3:], -1.0))
  close(ent.root_com_lin_vel_b, _quat_apply(rq, cv[:, :3], -1.0))
  close(ent.root_com_ang_vel_b, _quat_apply(rq, cv[:, 3:], -1.0))
  assert torch.equal(ent.root_link_pose_w, ent.body_link_pose_w[:, 0])
  jq = torch.from_numpy(np.asarray(model.jnt_qposadr)[ent.joint_ids]).long().cuda()
  jv = torch.from_numpy(np.asarray(model.jnt_dofadr)[ent.joint_ids]).long().cuda()
  assert torch.equal(ent.joint_pos, d.qpos[:, jq]) and torch.equal(ent.joint_vel, d.qvel[:, jv]) and torch.equal(ent.joint_acc, d.qacc[:, jv])
  assert ent.joint_pos.shape == (512, model.nu)
  # the root link's world velocity equals the free joint's qvel convention (linear world, angular body)
  close(ent.root_link_lin_vel_w if hasattr(ent, "root_link_lin_vel_w") else ent.root_link_vel_w[:, :3], d.qvel[:, :3], 2e-5)
  close(ent.root_link_ang_vel_b, d.qvel[:, 3:6], 2e-5)


def test_readback_matches_vectors_from_the_reference_code():
  """tests/golden/readback_reference.npz was computed by the reference's own torch functions
  (tools/make_readback_golden.py); the same mjData-shaped inputs are written into sim.data and
  the fused kernel must reproduce the outputs."""
  from pathlib import Path

  import torch

  from mjlab_amd import robots
  from mjlab_amd.entity_data import EntityReadback
  from mjlab_amd.sim import Simulation, SimulationCfg

  z = np.load(Path(__file__).parent / "golden" / "readback_reference.npz")
  model = robots.load_model("g1_velocity_flat")
  n = z["in_xpos"].shape[0]
  sim = Simulation(n, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
  for f in ("xpos", "xipos", "xquat", "cvel", "subtree_com"):
    getattr(sim.data, f)[:] = torch.from_numpy(z["in_" + f]).cuda()
  ent = EntityReadback(sim)
  ent.update()
  torch.cuda.synchronize()

  def close(a, name, tol=2e-6):
    b = torch.from_numpy(z[name]).cuda()
    assert float((a - b).abs().max()) / max(1.0, float(b.abs().max())) < tol, name

  close(ent.body_link_vel_w, "body_link_vel_w")
  close(ent.body_com_vel_w, "body_com_vel_w")
  close(ent.body_com_pose_w[..., 3:], "body_com_quat_w")
  for name in ("projected_gravity_b", "root_link_lin_vel_b", "root_link_ang_vel_b", "root_com_lin_vel_b", "root_com_ang_vel_b"):
    close(getattr(ent, name), name)
  close(ent.heading_w, "heading_w", 1e-5)


def test_readback_from_the_control_kernel_epilogue_equals_the_separate_launch():
  """SURVEY.md section 8f row 1 literally: the derived EntityData quantities "emitted by the step kernel's epilogue" --
  mjlab_control_t.readback_on refreshes them inside the one control-step launch, right after the forward() pass
  (before the interval push changes qvel); bit-identical to mjlab_entity_readback called after a forward()."""
  import torch

  from mjlab_amd import robots
  from mjlab_amd.entity_data import EntityReadback
  from mjlab_amd.rollout import VELOCITY_TASK_EVENTS, PhysicsRollout, g1_action_scale
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model("g1_velocity_flat")
  sim = Simulation(512, SimulationCfg(njmax=300), model, "cuda:0")
  ev = dict(VELOCITY_TASK_EVENTS["g1"])
  ev["push"] = None  # so that the state after the launch is the state the epilogue saw
  roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=5, substeps_per_call=4, control_kernel=True, **ev)
  fused, separate = EntityReadback(sim), EntityReadback(sim)
  roll.readback = fused
  for _ in range(8):
    roll.step(roll.random_action())
  separate.update()
  torch.cuda.synchronize()
  for name in ("body_link_pose_w", "body_link_vel_w", "body_com_pose_w", "body_com_vel_w", "_root", "joint_pos", "joint_vel", "joint_acc"):
    a, b = getattr(fused, name), getattr(separate, name)
    assert float(a.abs().max()) > 0 and torch.equal(a, b), name
