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


def test_inertia_from_geoms_matches_numerical_integration():
  """Bodies without <inertial>: mass, centre of mass and inertia come from the geoms (MJCF
  inertiafromgeom).  Checked against a brute-force voxel integration of the same solids."""
  import numpy as np

  from mjlab_amd import mjcf

  xml = """
  <mujoco>
    <worldbody>
      <body name="b" pos="0 0 1">
        <freejoint/>
        <geom type="box" size="0.2 0.1 0.05" pos="0.1 0 0" density="500"/>
        <geom type="capsule" size="0.05 0.15" pos="-0.1 0.05 0.1" quat="0.9238795 0.3826834 0 0" density="800"/>
        <geom type="sphere" size="0.08" pos="0 -0.2 0" mass="0.3"/>
        <geom type="cylinder" size="0.06 0.1" pos="0 0.2 -0.1" density="1200" contype="0" conaffinity="0"/>
      </body>
    </worldbody>
  </mujoco>
  """
  spec = mjcf.Spec.from_string(xml)
  m = spec.compile()
  body = spec.body("b")
  # voxel integration in the body frame
  h = 0.005
  ax = np.arange(-0.45, 0.45, h) + h / 2
  X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
  P = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
  mass, first, second = 0.0, np.zeros(3), np.zeros((3, 3))
  for g in body.geoms:
    L = (P - g.pos) @ mjcf.quat_to_mat(mjcf.quat_normalize(g.quat))  # geom-frame coordinates
    s = g.size
    if g.type == mjcf.GEOM_BOX:
      inside = (np.abs(L) <= s).all(axis=1)
    elif g.type == mjcf.GEOM_SPHERE:
      inside = (L**2).sum(axis=1) <= s[0] ** 2
    elif g.type == mjcf.GEOM_CYLINDER:
      inside = (L[:, 0] ** 2 + L[:, 1] ** 2 <= s[0] ** 2) & (np.abs(L[:, 2]) <= s[1])
    else:  # capsule
      zc = np.clip(L[:, 2], -s[1], s[1])
      inside = L[:, 0] ** 2 + L[:, 1] ** 2 + (L[:, 2] - zc) ** 2 <= s[0] ** 2
    vol = inside.sum() * h**3
    rho = g.mass / vol if g.mass is not None else g.density
    pts = P[inside]
    mass += rho * vol
    first += rho * h**3 * pts.sum(axis=0)
    second += rho * h**3 * ((pts**2).sum() * np.eye(3) - pts.T @ pts)
  com = first / mass
  inertia = second - mass * (com @ com * np.eye(3) - np.outer(com, com))
  assert abs(m.body_mass[1] - mass) / mass < 5e-3
  np.testing.assert_allclose(m.body_ipos[1], com, atol=2e-3)
  R = mjcf.quat_to_mat(m.body_iquat[1])
  np.testing.assert_allclose(R @ np.diag(m.body_inertia[1]) @ R.T, inertia, rtol=0, atol=1e-2 * np.abs(inertia).max())
  assert (np.diff(m.body_inertia[1]) <= 0).all() and np.isclose(np.linalg.det(R), 1.0)
  # analytic single-geom cases
  one = mjcf.Spec.from_string('<mujoco><worldbody><body><freejoint/><geom type="box" size="0.1 0.1 0.1" mass="0.1"/></body></worldbody></mujoco>').compile()
  assert np.isclose(one.body_mass[1], 0.1) and np.allclose(one.body_inertia[1], 0.1 / 3 * 0.02)
