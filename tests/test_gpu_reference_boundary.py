"""Row N1 on the GPU: tests/golden/boundary_reference.npz holds what the reference's OWN
``EntityData`` (src/mjlab/entity/data.py:34-516) computed over this repository's Bridge objects on live
rollout states, and the per-world model tables its ``randomize_field``
(src/mjlab/envs/mdp/events.py:212-265) drew (tools/make_boundary_golden.py; the reference cannot travel
to the GPU box).  Here the same states go into a real ``Simulation``:

  * every EntityData property must come out of ``sim.data`` / ``mjlab_entity_readback`` (the fused
    kernel for the ones it covers, the reference's index expression on ``sim.data`` for the rest);
  * the friction / mass / com / damping tables, written through ``sim.model.<field>[...]`` like
    ``randomize_field`` writes them, must change the physics per world, in agreement with the oracle
    given the same tables.

Tolerance 2e-5 relative: the fixture's inputs are fp64 oracle states rounded to fp32, the derived
arrays on the device are fp32 end to end.
"""

from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Z = Path(__file__).parent / "golden" / "boundary_reference.npz"


def _sim(z, n):
  import torch

  from mjlab_amd import robots
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model("g1_velocity_flat")
  sim = Simulation(n, SimulationCfg(njmax=300), model, "cuda:0")
  for f in ("qpos", "qvel", "ctrl", "qacc_warmstart", "xfrc_applied", "qfrc_applied"):
    getattr(sim.data, f)[:] = torch.from_numpy(z["in_" + f]).cuda().view_as(getattr(sim.data, f))
  return sim, model


def _quat_from_matrix(m):
  """Rotation matrices (..., 3, 3) -> quaternions wxyz, positive w (the convention of the reference's
  third_party/isaaclab/isaaclab/utils/math.py quat_from_matrix up to sign)."""
  import torch

  m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
  # Shepperd: the largest of the four squared components is taken by its root, the others by division
  t = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], dim=-1)
  k = t.argmax(dim=-1, keepdim=True)
  r = t.gather(-1, k).sqrt() * 2  # 4 * |largest component|
  a, b, c = m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1]  # 4 w x, 4 w y, 4 w z
  d, e, f = m[..., 1, 0] + m[..., 0, 1], m[..., 0, 2] + m[..., 2, 0], m[..., 2, 1] + m[..., 1, 2]  # 4 x y, 4 x z, 4 y z
  r4 = r.squeeze(-1)
  cand = torch.stack([
    torch.stack([r4 * r4 / 4, a, b, c], dim=-1),
    torch.stack([a, r4 * r4 / 4, d, e], dim=-1),
    torch.stack([b, d, r4 * r4 / 4, f], dim=-1),
    torch.stack([c, e, f, r4 * r4 / 4], dim=-1),
  ], dim=-2)  # row k = 4 * q_k * q
  q = cand.gather(-2, k.unsqueeze(-1).expand(*k.shape[:-1], 1, 4)).squeeze(-2) / r
  return q * torch.where(q[..., :1] < 0, -1.0, 1.0)


def test_every_entitydata_property_of_the_reference_is_reproduced():
  import torch

  from mjlab_amd.entity_data import EntityReadback, entity_indexing

  z = np.load(Z)
  n = z["in_qpos"].shape[0]
  sim, model = _sim(z, n)
  sim.forward()
  ent = EntityReadback(sim)
  ent.update()
  torch.cuda.synchronize()
  d, ix = sim.data, entity_indexing(model, "cuda")
  L = lambda t: t.long()  # noqa: E731

  def vel(pos, body_ids):
    sub = d.subtree_com[:, int(ix["bodies"][0].id)].unsqueeze(1)
    cv = d.cvel[:, body_ids]
    return torch.cat([cv[..., 3:6] - torch.cross(cv[..., 0:3], sub - pos, dim=-1), cv[..., 0:3]], dim=-1)

  gb, sb = L(sim.model.geom_bodyid[L(ix["geom_ids"])]), L(sim.model.site_bodyid[L(ix["site_ids"])])
  gpos, spos = d.geom_xpos[:, L(ix["geom_ids"])], d.site_xpos[:, L(ix["site_ids"])]
  got = {
    # the fused read-back kernel
    "body_link_pose_w": ent.body_link_pose_w, "body_link_vel_w": ent.body_link_vel_w,
    "body_com_pose_w": ent.body_com_pose_w, "body_com_vel_w": ent.body_com_vel_w,
    "root_link_pose_w": ent.root_link_pose_w, "root_link_vel_w": ent.root_link_vel_w, "root_com_vel_w": ent.root_com_vel_w,
    "projected_gravity_b": ent.projected_gravity_b, "heading_w": ent.heading_w,
    "root_link_lin_vel_b": ent.root_link_lin_vel_b, "root_link_ang_vel_b": ent.root_link_ang_vel_b,
    "root_com_lin_vel_b": ent.root_com_lin_vel_b, "root_com_ang_vel_b": ent.root_com_ang_vel_b,
    "joint_pos": ent.joint_pos, "joint_vel": ent.joint_vel, "joint_acc": ent.joint_acc,
    # plain reads of sim.data with the entity's index tables
    "body_external_wrench": d.xfrc_applied[:, L(ix["body_ids"])],
    "actuator_force": d.actuator_force[:, L(ix["ctrl_ids"])],
    "generalized_force": d.qfrc_applied[:, L(ix["free_joint_v_adr"])],
    "geom_pose_w": torch.cat([gpos, _quat_from_matrix(d.geom_xmat[:, L(ix["geom_ids"])])], dim=-1),
    "geom_vel_w": vel(gpos, gb),
    "site_pose_w": torch.cat([spos, _quat_from_matrix(d.site_xmat[:, L(ix["site_ids"])])], dim=-1),
    "site_vel_w": vel(spos, sb),
  }  # fmt: skip
  derived = {
    "root_link_pos_w": got["root_link_pose_w"][:, :3], "root_link_quat_w": got["root_link_pose_w"][:, 3:],
    "root_link_lin_vel_w": got["root_link_vel_w"][:, :3], "root_link_ang_vel_w": got["root_link_vel_w"][:, 3:],
    "root_com_lin_vel_w": got["root_com_vel_w"][:, :3], "root_com_ang_vel_w": got["root_com_vel_w"][:, 3:],
    "body_link_pos_w": got["body_link_pose_w"][..., :3], "body_link_quat_w": got["body_link_pose_w"][..., 3:],
    "body_link_lin_vel_w": got["body_link_vel_w"][..., :3], "body_link_ang_vel_w": got["body_link_vel_w"][..., 3:],
    "body_com_pos_w": got["body_com_pose_w"][..., :3], "body_com_quat_w": got["body_com_pose_w"][..., 3:],
    "body_com_lin_vel_w": got["body_com_vel_w"][..., :3], "body_com_ang_vel_w": got["body_com_vel_w"][..., 3:],
    "body_external_force": got["body_external_wrench"][..., :3], "body_external_torque": got["body_external_wrench"][..., 3:],
    "geom_pos_w": got["geom_pose_w"][..., :3], "geom_quat_w": got["geom_pose_w"][..., 3:],
    "geom_lin_vel_w": got["geom_vel_w"][..., :3], "geom_ang_vel_w": got["geom_vel_w"][..., 3:],
    "site_pos_w": got["site_pose_w"][..., :3], "site_quat_w": got["site_pose_w"][..., 3:],
    "site_lin_vel_w": got["site_vel_w"][..., :3], "site_ang_vel_w": got["site_vel_w"][..., 3:],
  }  # fmt: skip
  got.update(derived)
  names = [k[3:] for k in z.files if k.startswith("ed_") and not k.startswith("ed_sensor_")]
  assert sorted(names) == sorted(got), set(names) ^ set(got)  # every property of the fixture is covered
  worst = {}
  for name in names:
    want = torch.from_numpy(z["ed_" + name]).cuda()
    have = got[name]
    assert have.shape == want.shape, (name, have.shape, want.shape)
    if name.endswith("quat_w") or name.endswith("pose_w"):  # q and -q are the same rotation
      qh, qw = have[..., -4:], want[..., -4:]
      sign = torch.sign((qh * qw).sum(-1, keepdim=True))
      have = torch.cat([have[..., :-4], qh * sign], dim=-1)
    if name == "heading_w":  # an angle: compare on the circle
      err = float(torch.atan2(torch.sin(have - want), torch.cos(have - want)).abs().max())
    else:
      err = float((have - want).abs().max()) / max(1.0, float(want.abs().max()))
    worst[name] = err
    tol = 1e-3 if name in ("joint_acc",) else 2e-5  # joint_acc = qacc: through the Newton solve
    assert err < tol, (name, err)
  for k in [k for k in z.files if k.startswith("ed_sensor_")]:
    adr = ix["sensor_adr"][k[len("ed_sensor_"):]]
    assert torch.equal(d.sensordata[:, L(adr)], torch.from_numpy(z[k]).cuda()), k
  print({k: f"{v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})


def test_tables_drawn_by_the_reference_randomize_field_change_the_physics_per_world():
  import torch

  from oracle.oracle import OracleSim

  z = np.load(Z)
  n = z["in_qpos"].shape[0]
  sim, model = _sim(z, n)
  # every world gets the SAME state, so any difference between worlds comes from the model tables
  for f in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(sim.data, f)[:] = getattr(sim.data, f)[3:4].clone()
  sim.data.xfrc_applied.zero_()
  sim.data.qfrc_applied.zero_()
  sim.data.qpos[:, 2] -= 0.004  # feet firmly on the ground
  ora = OracleSim(model, n, njmax=300, precision="f64")
  for f in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(ora, f)[:] = getattr(sim.data, f).cpu().numpy()
  sim.forward()
  torch.cuda.synchronize()
  base_qacc = sim.data.qacc.clone()
  assert float((base_qacc - base_qacc[0]).abs().max()) == 0.0  # identical worlds so far
  fields = ["geom_friction", "body_mass", "body_ipos", "dof_damping"]
  sim.expand_model_fields(fields)
  sim.create_graph()
  for f in fields:
    table = torch.from_numpy(z["dr_" + f]).cuda()
    ptr = getattr(sim.model, f).data_ptr()
    # the write pattern of randomize_field: model_field[env_grid, entity_grid] = values
    env_grid, ent_grid = torch.meshgrid(torch.arange(n, device="cuda"), torch.arange(table.shape[1], device="cuda"), indexing="ij")
    getattr(sim.model, f)[env_grid, ent_grid] = table
    assert getattr(sim.model, f).data_ptr() == ptr
    ora.expand_model_field(f)[:] = z["dr_" + f]
  assert int(sim.data.fold_valid.sum()) == 0  # handing out a per-world field dropped the forward() snapshot
  sim.forward()
  ora.forward(nthread=8)
  torch.cuda.synchronize()
  assert np.array_equal(sim.data.nefc.cpu().numpy().ravel(), ora.nefc.ravel())
  # friction as the contacts saw it: per world, per foot geom (max of foot and plane friction)
  ncon = int(sim.data.ncon.min())
  assert ncon > 0
  cf = sim.data.contact_friction.view(n, -1, 5)[:, :ncon, 0]
  assert torch.unique(cf).numel() > n  # per-world, per-geom values reached the contacts
  qacc = sim.data.qacc
  assert float((qacc - qacc[0]).abs().max()) > 1e-2  # the worlds now move differently
  assert float((qacc - base_qacc).abs().max()) > 1e-2

  def rel(a, b):
    a = a.cpu().numpy().astype(np.float64).reshape(n, -1)
    b = b.reshape(n, -1)
    return float((np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-6)).max())

  assert rel(sim.data.qM, ora.qM) < 2e-6
  assert rel(sim.data.qfrc_bias, ora.qfrc_bias) < 2e-5
  assert rel(sim.data.qacc, ora.qacc) < 2e-4
  sim.step()
  ora.step(1, nthread=8)
  assert rel(sim.data.qpos, ora.qpos) < 2e-6
  assert rel(sim.data.qvel, ora.qvel) < 2e-4
