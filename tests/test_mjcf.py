"""Model compiler: the constants the reference's own tests pin (tests/test_g1_constants.py:37-124,
tests/test_go1_constants.py:33-94) plus structural counts from SURVEY.md section 8."""

import numpy as np
import pytest

from mjlab_amd import mjcf, robots


@pytest.fixture(scope="module")
def g1():
  return robots.load_model("g1_velocity_flat")


@pytest.fixture(scope="module")
def go1():
  return robots.load_model("go1_velocity_flat")


def test_g1_sizes(g1):
  assert (g1.nq, g1.nv, g1.nu, g1.nbody, g1.ngeom, g1.npair) == (36, 35, 29, 32, 69, 502)
  assert g1.njnt == 30 and g1.nsensordata == 2
  assert g1.names["body"][:3] == ["world", "terrain", "robot/pelvis"]


def test_go1_sizes(go1):
  assert (go1.nq, go1.nv, go1.nu, go1.nbody, go1.ngeom, go1.npair) == (19, 18, 12, 15, 44, 30)
  assert go1.nsensordata == 4


def test_g1_actuator_params(g1):
  # gainprm[0] == stiffness, biasprm[1] == -stiffness, biasprm[2] == -damping, forcerange == +-effort
  cfgs = robots.g1_actuators()
  for i in range(g1.nu):
    name = g1.names["actuator"][i].split("/")[-1]
    cfg = [c for c in cfgs if mjcf.filter_exp(c.joint_names_expr, [name])][-1]
    a = g1.actuator(i)
    assert a.gainprm[0] == pytest.approx(cfg.stiffness)
    assert a.biasprm[1] == pytest.approx(-cfg.stiffness)
    assert a.biasprm[2] == pytest.approx(-cfg.damping)
    assert tuple(a.forcerange) == pytest.approx((-cfg.effort_limit, cfg.effort_limit))
    j = g1.actuator_trnid[i, 0]
    assert g1.dof_armature[g1.jnt_dofadr[j]] == pytest.approx(cfg.armature)
    assert tuple(g1.actuator_ctrlrange[i]) == pytest.approx(tuple(g1.jnt_range[j]))


def test_g1_keyframe(g1):
  q = g1.key("init_state").qpos
  assert q[:3] == pytest.approx([0, 0, 0.76])
  assert q[3:7] == pytest.approx([1, 0, 0, 0])
  names = [n.split("/")[-1] for n in g1.names["joint"][1:]]
  assert q[7 + names.index("left_knee_joint")] == pytest.approx(0.669)
  assert q[7 + names.index("right_shoulder_roll_joint")] == pytest.approx(-0.2)
  assert g1.key("init_state").ctrl == pytest.approx(q[7:])


def test_g1_foot_geoms(g1):
  feet = [i for i, n in enumerate(g1.names["geom"]) if "foot" in n and n.endswith("_collision")]
  assert len(feet) == 14
  for i in feet:
    g = g1.geom(i)
    assert g.condim[0] == 3 and g.priority[0] == 1 and g.friction[0] == pytest.approx(0.6)
  others = [i for i, n in enumerate(g1.names["geom"]) if n.endswith("_collision") and "foot" not in n]
  for i in others:
    assert g1.geom(i).condim[0] == 1 and g1.geom(i).priority[0] == 0


def test_go1_feet(go1):
  feet = [i for i, n in enumerate(go1.names["geom"]) if n.endswith("_foot_collision")]
  assert len(feet) == 4
  for i in feet:
    assert go1.geom_condim[i] == 3 and go1.geom_priority[i] == 1
    assert go1.geom_solimp[i][:3] == pytest.approx([0.9, 0.95, 0.023])
  assert go1.njnt == 13 and go1.nu == 12


def test_compile_constants_consistent(g1):
  # invweight0 is M^-1-derived, positive for moving bodies/dofs; static bodies get zero
  assert np.all(g1.dof_invweight0 > 0)
  assert np.all(g1.body_invweight0[:2] == 0)
  assert np.all(g1.body_invweight0[2:] > 0)
  kin = mjcf.kinematics_np(g1, g1.qpos0)
  M = mjcf.mass_matrix_np(g1, kin)
  assert np.allclose(M, M.T) and np.all(np.linalg.eigvalsh(M) > 0)
  assert g1.meaninertia == pytest.approx(np.mean(np.diag(M)))
  assert g1.body_subtreemass[0] == pytest.approx(g1.body_mass.sum())


def test_fromto_capsule():
  spec = mjcf.Spec.from_string(
    """<mujoco><worldbody><body name="b"><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/>
    <joint name="j" type="hinge" axis="0 0 1"/><geom name="g" type="capsule" size="0.1" fromto="0 0 0 0 0 -1"/></body></worldbody></mujoco>"""
  )
  m = spec.compile()
  assert m.geom_size[0][:2] == pytest.approx([0.1, 0.5])
  assert m.geom_pos[0] == pytest.approx([0, 0, -0.5])
  assert m.geom_rbound[0] == pytest.approx(0.6)
  R = mjcf.quat_to_mat(m.geom_quat[0])
  assert np.abs(R[:, 2]) == pytest.approx([0, 0, 1])


def test_default_classes_and_errors():
  xml = """<mujoco><default><default class="a"><geom size="0.2" type="sphere"/><default class="b"><geom size="0.3"/></default></default></default>
  <worldbody><body name="x" childclass="a"><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/><freejoint/>
  <geom name="g1"/><geom name="g2" class="b"/></body></worldbody></mujoco>"""
  m = mjcf.Spec.from_string(xml).compile()
  assert m.geom_size[0][0] == pytest.approx(0.2) and m.geom_size[1][0] == pytest.approx(0.3)
  with pytest.raises(NotImplementedError):
    mjcf.Spec.from_string('<mujoco><worldbody><body><joint type="ball"/></body></worldbody></mujoco>')


def test_assets_match_reference_compile(reference_root):
  """Committed .npz models are exactly what compiling the reference MJCF gives today."""
  for name in robots.SCENES:
    a, b = robots.load_model(name), robots.compile_scene(name)
    for k, v in b.__dict__.items():
      if isinstance(v, np.ndarray):
        assert np.array_equal(getattr(a, k), v), (name, k)
    assert a.names == b.names


def test_model_roundtrip(tmp_path, g1):
  p = tmp_path / "m.npz"
  g1.save(p)
  m2 = mjcf.Model.load(p)
  assert m2.opt.timestep == g1.opt.timestep and m2.opt.gravity == tuple(g1.opt.gravity)
  assert np.array_equal(m2.pair_geom, g1.pair_geom) and m2.names == g1.names
