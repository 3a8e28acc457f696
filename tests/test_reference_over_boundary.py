"""Row N1: the reference's OWN reader / writer code runs unmodified over this repository's boundary
objects (``mjlab_amd.sim_data.Bridge`` = the class of ``Simulation.data`` / ``Simulation.model``,
``entity_indexing`` = its index tables).  Needs /root/reference (skipped elsewhere); the vectors it
checks are the ones tests/test_gpu_reference_boundary.py replays on a real ``Simulation`` on the GPU.

  * ``EntityData`` (reference src/mjlab/entity/data.py:34-516): every property evaluates over the
    Bridges and equals the committed fixture; the writers land where mjData says they should and
    keep every storage pointer;
  * ``randomize_field`` (reference src/mjlab/envs/mdp/events.py:212-265): writes per-world values
    into ``sim.model.<field>[env_grid, ids]`` in place;
  * the Bridge cases of the reference's tests/test_sim_data.py:62-81 and the per-world cases of
    tests/test_domain_randomization.py:150,176 restated on Bridge.
"""

import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))


@pytest.fixture(scope="module")
def ref(reference_root):
  import make_boundary_golden as mbg

  EntityData, EntityIndexing, events = mbg.import_reference()
  from mjlab_amd import robots

  model = robots.load_model(mbg.SCENE)
  ora, inputs = mbg.oracle_states(model)
  mb, db = mbg.host_bridges(model, mbg.NWORLD, ora.dfield)
  ed = mbg.reference_entity_data(EntityData, EntityIndexing, model, mb, db, mbg.NWORLD)
  return mbg, model, mb, db, ed, events


def test_committed_vectors_are_what_the_reference_computes(ref):
  mbg = ref[0]
  fresh = mbg.compute()
  z = np.load(mbg.DST)
  assert sorted(z.files) == sorted(fresh)
  for k in z.files:
    assert np.array_equal(z[k], fresh[k]), k


def test_every_entitydata_property_evaluates_over_the_bridge(ref):
  mbg, model, mb, db, ed, _ = ref
  n = mbg.NWORLD
  nb = 30
  shapes = {"root_link_pose_w": (n, 7), "body_link_pose_w": (n, nb, 7), "body_com_vel_w": (n, nb, 6), "geom_pose_w": (n, 68, 7),
            "site_vel_w": (n, 6, 6), "joint_pos": (n, 29), "actuator_force": (n, 29), "projected_gravity_b": (n, 3), "heading_w": (n,)}  # fmt: skip
  for p in mbg.PROPERTIES:
    v = getattr(ed, p)
    assert torch.isfinite(v).all(), p
    if p in shapes:
      assert tuple(v.shape) == shapes[p], (p, v.shape)
  # conventions the engine must satisfy for these readers: the free joint's qvel is
  # [linear world, angular body]; the root link velocity derived from cvel / subtree_com agrees with it
  assert torch.allclose(ed.root_link_lin_vel_w, db.qvel[:, 0:3], atol=2e-5)
  assert torch.allclose(ed.root_link_ang_vel_b, db.qvel[:, 3:6], atol=2e-5)
  assert set(ed.sensor_data) == {"left_foot_ground_contact", "right_foot_ground_contact"}
  # the properties that raise upstream (whatever the engine) raise here in the same way
  for p in mbg.RAISING:
    with pytest.raises(Exception):  # NotImplementedError / torch.jit.Error wrapping the ValueError
      getattr(ed, p)


def test_entitydata_writers_write_through_the_bridge_in_place(ref):
  mbg, model, mb, db, ed, _ = ref
  n = mbg.NWORLD
  ptr = {f: getattr(db, f).data_ptr() for f in ("qpos", "qvel", "ctrl", "xfrc_applied", "qfrc_applied")}
  keep = {f: getattr(db, f).clone() for f in ptr}
  envs = torch.tensor([1, 5, 9])
  state = torch.arange(3 * 13, dtype=torch.float32).reshape(3, 13)
  ed.write_root_state(state, envs)
  assert torch.equal(db.qpos[envs, :7], state[:, :7]) and torch.equal(db.qvel[envs, :6], state[:, 7:])
  assert torch.equal(db.qpos[0], keep["qpos"][0])  # other worlds untouched
  jp, jv = torch.full((3, 29), 0.25), torch.full((3, 29), -0.5)
  ed.write_joint_state(jp, jv, None, envs)
  assert torch.equal(db.qpos[envs, 7:], jp) and torch.equal(db.qvel[envs, 6:], jv)
  ed.write_joint_position(torch.ones(n, 2), torch.tensor([3, 4]), None)
  assert torch.equal(db.qpos[:, 7 + 3], torch.ones(n)) and torch.equal(db.qpos[:, 7 + 4], torch.ones(n))
  ed.write_ctrl(torch.full((3, 29), 2.0), None, envs)
  assert torch.equal(db.ctrl[envs], torch.full((3, 29), 2.0))
  ed.write_external_wrench(torch.ones(n, 1, 3), 2 * torch.ones(n, 1, 3), [4], None)
  gb = int(ed.indexing.body_ids[4])
  assert torch.equal(db.xfrc_applied[:, gb, :3], torch.ones(n, 3)) and torch.equal(db.xfrc_applied[:, gb, 3:], 2 * torch.ones(n, 3))
  ed.clear_state(envs)
  assert float(db.xfrc_applied[envs].abs().sum()) == 0 and float(db.ctrl[envs].abs().sum()) == 0
  assert float(db.xfrc_applied[0].abs().sum()) > 0
  for f, p in ptr.items():
    assert getattr(db, f).data_ptr() == p, f  # graph-safe: the storage never moved
  for f, v in keep.items():
    getattr(db, f)[:] = v


def test_randomize_field_writes_per_world_values_in_place(ref):
  mbg, model, mb_shared, _, _, events = ref
  n = mbg.NWORLD
  mb, _ = mbg.host_bridges(model, n)
  base = torch.from_numpy(np.asarray(model.geom_friction, dtype=np.float32))
  # a shared (stride-0) field is one storage for all worlds: the reference expands before randomising
  # (envs/manager_based_env.py:124-129), and so does this test
  env = mbg.mock_env(model, mb, n)
  feet = mbg.foot_geom_local_ids(model)
  from types import SimpleNamespace

  cfg = SimpleNamespace(name="robot", joint_ids=slice(None), body_ids=slice(None), geom_ids=feet, site_ids=slice(None))
  t = mbg.expand_host_field(mb, "geom_friction")
  p = t.data_ptr()
  events.randomize_field(env, None, "geom_friction", ranges=(0.3, 1.2), asset_cfg=cfg)
  assert mb.geom_friction.data_ptr() == p
  gids = env.scene["robot"].indexing.geom_ids[feet].long()
  f0 = mb.geom_friction[:, gids, 0]
  assert float(f0.min()) >= 0.3 and float(f0.max()) <= 1.2
  assert torch.unique(f0).numel() == n * len(feet)  # a draw per world and geom
  others = torch.ones(model.ngeom, dtype=torch.bool)
  others[gids] = False
  assert torch.equal(mb.geom_friction[:, others], base[others].expand(n, -1, -1))  # nothing else moved
  assert torch.equal(mb.geom_friction[:, gids, 1:], base[gids, 1:].expand(n, -1, -1))  # default axis: 0 only
  # subsets of worlds (test_domain_randomization.py:150,176: body_mass / dof_damping on chosen envs)
  mbg.expand_host_field(mb, "body_mass")
  sel = torch.tensor([0, 2, 7])
  cfg_all = SimpleNamespace(name="robot", joint_ids=slice(None), body_ids=slice(None), geom_ids=slice(None), site_ids=slice(None))
  events.randomize_field(env, sel, "body_mass", ranges=(1.5, 2.0), operation="scale", asset_cfg=cfg_all)
  m0 = torch.from_numpy(np.asarray(model.body_mass, dtype=np.float32))
  ratio = mb.body_mass[:, 2:] / m0[2:]
  assert bool(((ratio[sel] >= 1.5) & (ratio[sel] <= 2.0)).all())
  rest = torch.ones(n, dtype=torch.bool)
  rest[sel] = False
  assert torch.equal(mb.body_mass[rest], m0.expand(int(rest.sum()), -1))
  with pytest.raises(ValueError, match="Unknown field"):
    events.randomize_field(env, None, "not_a_field", ranges=(0.0, 1.0))


def test_bridge_contract_of_the_reference_tests():
  """reference tests/test_sim_data.py:62-81 on Bridge: slice assignment keeps the address, attribute
  assignment raises with the reference's message, repeated access yields the same object."""
  from mjlab_amd.sim_data import Bridge

  b = Bridge("WarpBridge", {"arr": torch.tensor([[1.0, 2.0], [3.0, 4.0]])}, {"val": 1.0})
  p = b.arr.data_ptr()
  b.arr[:] = torch.zeros((2, 2))
  assert b.arr.data_ptr() == p and bool((b.arr == 0).all())
  with pytest.raises(AttributeError, match="Cannot set attribute 'arr' on WarpBridge"):
    b.arr = torch.zeros((2, 2))
  with pytest.raises(AttributeError, match="Use in-place operations instead"):
    b.val = 42.0
  assert b.arr is b.arr and b.val == 1.0
