"""Box-terrain collision of the CPU oracle: sphere / capsule vs static boxes through the grid
broadphase.  Analytic cases, equivalence with the ground plane on a flat slab, broadphase vs
exhaustive search, and the rough-terrain G1 scene."""

import numpy as np
import pytest

from mjlab_amd import mjcf, robots, terrains
from mjlab_amd.mjcf import GEOM_BOX, Spec
from oracle.oracle import OracleSim

BODIES_XML = """
<mujoco model="probes">
  <compiler angle="radian"/>
  <option timestep="0.002"/>
  <worldbody>
    <body name="ball" pos="0 0 1">
      <inertial pos="0 0 0" mass="1" diaginertia="0.004 0.004 0.004"/>
      <freejoint name="ball_root"/>
      <geom name="ball_geom" type="sphere" size="0.1"/>
    </body>
    <body name="cap" pos="2 0 1" quat="0.707107 0 0.707107 0">
      <inertial pos="0 0 0" mass="1" diaginertia="0.02 0.02 0.002"/>
      <freejoint name="cap_root"/>
      <geom name="cap_geom" type="capsule" size="0.05 0.2"/>
    </body>
  </worldbody>
</mujoco>
"""


def probes_on(boxes, plane=False) -> mjcf.Model:
  """Free sphere (r 0.1) + free capsule (r 0.05, half length 0.2, axis along x) over static boxes."""
  spec = Spec.from_string(BODIES_XML)
  spec.option.integrator = mjcf.INT_IMPLICITFAST
  scene = Spec()
  scene.option = spec.option
  t = scene.add_body("terrain")
  if plane:
    scene.add_geom(t, "terrain", mjcf.GEOM_PLANE, (0, 0, 0.01))
  else:
    terrains.add_boxes(scene, t, np.asarray(boxes, dtype=np.float64))
  scene.attach(spec)
  return scene.compile()


SLAB = [[0, 0, -0.5, 20, 20, 0.5]]  # top face at z = 0


def contacts(o, w=0):
  n = int(o.ncon[w, 0])
  return dict(n=n, dist=o.contact_dist[w, :n], pos=o.contact_pos[w, :n], frame=o.contact_frame[w, :n], geom=o.contact_geom[w, :n])


def test_sphere_on_face_edge_corner_and_inside():
  m = probes_on([[0, 0, -0.5, 1, 1, 0.5]])
  o = OracleSim(m, nworld=5)
  o.qpos[:, 7:10] = [50, 0, 5]  # capsule out of the way
  o.qpos[:, 3] = o.qpos[:, 10] = 1
  r = 0.1
  o.qpos[0, :3] = [0.2, -0.3, 0.08]  # over the top face, 2 cm deep
  o.qpos[1, :3] = [1.05, 0.0, 0.05]  # beyond the +x edge, diagonal normal
  o.qpos[2, :3] = [1.04, 1.04, 0.04]  # beyond the (+x, +y, +z) corner
  o.qpos[3, :3] = [0.3, 0.2, -0.01]  # centre inside the box, nearest face = top
  o.qpos[4, :3] = [0.0, 0.0, 0.2]  # clear of the box
  o.forward()
  c = contacts(o, 0)
  assert c["n"] == 1 and np.isclose(c["dist"][0], -0.02)
  np.testing.assert_allclose(c["frame"][0, :3], [0, 0, -1], atol=1e-12)  # from the sphere into the box
  np.testing.assert_allclose(c["pos"][0], [0.2, -0.3, -0.01], atol=1e-12)  # midway between the surfaces
  assert tuple(c["geom"][0]) == (m.names["geom"].index("ball_geom"), m.names["geom"].index("terrain_0"))
  c = contacts(o, 1)
  d = np.array([0.05, 0.0, 0.05])
  assert c["n"] == 1 and np.isclose(c["dist"][0], np.linalg.norm(d) - r)
  np.testing.assert_allclose(c["frame"][0, :3], -d / np.linalg.norm(d), atol=1e-12)
  c = contacts(o, 2)
  d = np.array([0.04, 0.04, 0.04])
  assert c["n"] == 1 and np.isclose(c["dist"][0], np.linalg.norm(d) - r)
  np.testing.assert_allclose(c["frame"][0, :3], -d / np.linalg.norm(d), atol=1e-12)
  c = contacts(o, 3)
  assert c["n"] == 1 and np.isclose(c["dist"][0], -0.01 - r)
  np.testing.assert_allclose(c["frame"][0, :3], [0, 0, -1], atol=1e-12)
  assert contacts(o, 4)["n"] == 0


def test_capsule_on_face_matches_plane_capsule():
  """Over a face the capsule's contacts are plane_capsule's two end-point contacts."""
  mb, mp = probes_on(SLAB), probes_on(None, plane=True)
  rng = np.random.default_rng(0)
  for _ in range(20):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    z = rng.uniform(0.0, 0.3)
    res = []
    for m in (mb, mp):
      o = OracleSim(m, nworld=1)
      o.qpos[0, :3] = [50, 0, 5]
      o.qpos[0, 3] = 1
      o.qpos[0, 7:10] = [rng.uniform(-1, 1) * 0, 0, z]
      o.qpos[0, 10:14] = q
      o.forward()
      res.append(contacts(o))
    b, p = res
    assert b["n"] == p["n"]
    if b["n"]:
      np.testing.assert_allclose(b["dist"], p["dist"], atol=1e-12)
      np.testing.assert_allclose(b["pos"], p["pos"], atol=1e-12)
      np.testing.assert_allclose(b["frame"][:, :3], -p["frame"][:, :3], atol=1e-12)  # geom1 -> geom2 flips


def test_capsule_across_an_edge_and_a_ridge():
  # capsule along x, centred on the +x edge of a 1 m box: one end over the top face, one in the air
  m = probes_on([[0, 0, -0.5, 1, 1, 0.5]])
  o = OracleSim(m, nworld=2)
  o.qpos[:, :3] = [50, 0, 5]
  o.qpos[:, 3] = 1
  o.qpos[0, 7:10] = [1.0, 0.0, 0.04]  # axis spans x in [0.8, 1.2], 1 cm deep
  o.qpos[0, 10:14] = [0.707107, 0, 0.707107, 0]
  o.forward()
  c = contacts(o, 0)
  # the end over the face and the point where the axis leaves the face (x = 1), both 1 cm deep
  assert c["n"] == 2
  np.testing.assert_allclose(np.sort(c["pos"][:, 0]), [0.8, 1.0], atol=1e-5)
  np.testing.assert_allclose(c["dist"], -0.01, atol=1e-6)
  np.testing.assert_allclose(c["frame"][:, :3], [[0, 0, -1]] * 2, atol=1e-5)
  # thin ridge (2 cm wide) crossed at right angles by the capsule: one contact in the middle
  m = probes_on([[0, 0, -0.5, 0.01, 1, 0.5]])
  o = OracleSim(m, nworld=1)
  o.qpos[0, :3] = [50, 0, 5]
  o.qpos[0, 3] = 1
  o.qpos[0, 7:10] = [0.0, 0.0, 0.045]
  o.qpos[0, 10:14] = [0.707107, 0, 0.707107, 0]
  o.forward()
  c = contacts(o)
  assert c["n"] == 2  # both ends of the 2 cm support interval
  np.testing.assert_allclose(np.sort(c["pos"][:, 0]), [-0.01, 0.01], atol=1e-5)
  np.testing.assert_allclose(c["dist"], -0.005, atol=1e-6)


def test_sphere_and_capsule_come_to_rest_on_a_box():
  m = probes_on([[0, 0, -0.5, 5, 5, 0.5]])
  o = OracleSim(m, nworld=1)
  o.qpos[0, :3] = [0.3, 0.2, 0.15]
  o.qpos[0, 3] = 1
  o.qpos[0, 7:10] = [2.0, 0.1, 0.1]
  o.qpos[0, 10:14] = [0.707107, 0, 0.707107, 0]
  o.step(1500)
  assert abs(o.qpos[0, 2] - 0.1) < 2e-3 and abs(o.qpos[0, 9] - 0.05) < 2e-3
  assert np.abs(o.qvel).max() < 1e-3
  # contact forces carry the weight: total normal force = (m1 + m2) g
  o.forward()
  assert np.isclose(o.qfrc_constraint[0, 2] + o.qfrc_constraint[0, 8], 2 * 9.81, rtol=1e-3)


def dense_columns_cfg() -> terrains.TerrainGeneratorCfg:
  """12 cm columns: the capsule probe's bounding sphere (25 cm) reaches more than MJLAB_TCAND_MAX = 12 of them."""
  sub = terrains.BoxRandomGridTerrainCfg(grid_width=0.12, grid_height_range=(0.01, 0.03), platform_width=0.5)
  return terrains.TerrainGeneratorCfg(size=(4.0, 4.0), seed=3, num_rows=1, num_cols=2, sub_terrains={"grid": sub})


@pytest.mark.parametrize("kind", ["stairs", "dense_columns"])
def test_grid_broadphase_equals_exhaustive_search(monkeypatch, kind):
  """Same terrain compiled with 0.5 m cells and with one huge cell (every box a candidate of
  every geom) gives identical contacts -- also when a geom reaches more boxes than the candidate
  list holds (both keep the MJLAB_TCAND_MAX smallest ids, whatever the walk order)."""
  cfg = terrains.rough_terrains_cfg(seed=5, num_rows=3, num_cols=5) if kind == "stairs" else dense_columns_cfg()
  cfg.border_width = 2.0
  t = terrains.TerrainGenerator(cfg).generate()
  m_grid = probes_on(t.boxes)
  monkeypatch.setattr(mjcf, "TERRAIN_CELL", 1000.0)
  m_all = probes_on(t.boxes)
  assert m_grid.tgrid_nx > 10 and m_all.tgrid_nx == 1 and m_all.tgrid_ny == 1
  rng = np.random.default_rng(1)
  nw = 64
  og, oa = OracleSim(m_grid, nworld=nw), OracleSim(m_all, nworld=nw)
  for o in (og, oa):
    r = np.random.default_rng(2)
    # probes near the surface of randomly chosen boxes (on top, beside, across edges)
    b = t.boxes[r.integers(0, len(t.boxes), size=(nw, 2))]
    p = b[..., :3] + r.uniform(-1.05, 1.05, size=(nw, 2, 3)) * b[..., 3:]
    p[..., 2] = b[..., 2] + b[..., 5] + r.uniform(-0.02, 0.12, size=(nw, 2))
    o.qpos[:, 0:3], o.qpos[:, 7:10] = p[:, 0], p[:, 1]
    q = r.normal(size=(nw, 4))
    o.qpos[:, 3] = 1
    o.qpos[:, 10:14] = q / np.linalg.norm(q, axis=1, keepdims=True)
    o.forward()
  assert og.ncon.sum() > 40 and (og.ncon > 0).mean() > 0.5
  if kind == "dense_columns":
    # the candidate list really overflows: count the columns within reach of the capsule probes
    reach = 0.25  # bounding sphere: radius 0.05 + half length 0.2
    d = np.maximum(np.abs(og.qpos[:, None, 7:10] - t.boxes[None, :, :3]) - t.boxes[None, :, 3:], 0.0)
    assert ((d**2).sum(axis=2) <= reach**2).sum(axis=1).max() > 12
  np.testing.assert_array_equal(og.ncon, oa.ncon)
  for f in ("contact_dist", "contact_pos", "contact_frame", "contact_geom", "efc_J", "qacc"):
    np.testing.assert_array_equal(getattr(og, f), getattr(oa, f))
  del rng


def g1_on_slab() -> mjcf.Model:
  sensors = tuple(
    robots.ContactSensorCfg(name=f"{s}_foot_ground_contact", body1=f"{s}_ankle_roll_link", body2="terrain", num=1, data=("found",), reduce="netforce")
    for s in ("left", "right")
  )
  cfg = terrains.TerrainGeneratorCfg(size=(8.0, 8.0), seed=0, sub_terrains={"flat": terrains.BoxFlatTerrainCfg()})
  spec = robots.build_scene(robots.g1_spec(sensors), robots.G1_KNEES_BENT, cfg)
  for s in spec.sensors:
    if s.refname == "robot/terrain":
      s.refname = "terrain"
  robots._task_options(spec)
  return spec.compile()


@pytest.mark.skipif(not robots.REFERENCE_ROOT.exists(), reason="needs the reference robot MJCF")
def test_g1_on_a_flat_slab_equals_g1_on_the_plane():
  """The flat sub-terrain (a 1 m slab whose top is z = 0) must give the physics of the ground
  plane: same contact count, depths and points, same accelerations and sensor readings."""
  mb = g1_on_slab()
  mp = robots.load_model("g1_velocity_flat")
  assert mb.nterrain == 1 and mb.ntgeom == 33 and mb.npair == mp.npair - 33
  rng = np.random.default_rng(3)
  nw = 16
  qpos = np.tile(mp.key_qpos[0], (nw, 1))
  qpos[:, 2] -= rng.uniform(0.0, 0.03, nw)
  qpos[:, 7:] += rng.normal(scale=0.1, size=(nw, mp.nq - 7))
  qpos[:, 0:2] += rng.uniform(-2, 2, (nw, 2))  # the slab spans [-4, 4] x [-4, 4]
  qvel = rng.normal(scale=0.3, size=(nw, mp.nv))
  res = []
  for m in (mb, mp):
    o = OracleSim(m, nworld=nw, njmax=300)
    o.qpos[:], o.qvel[:], o.ctrl[:] = qpos, qvel, m.key_ctrl[0]
    o.forward()
    res.append(o)
  b, p = res
  np.testing.assert_array_equal(b.ncon, p.ncon)
  assert b.ncon.min() > 4
  np.testing.assert_array_equal(b.nefc, p.nefc)
  # the plane pairs come first in the flat scene, the terrain contacts last in the slab scene:
  # compare as sets of (depth, point)
  for w in range(nw):
    n = int(b.ncon[w, 0])
    kb = np.concatenate([b.contact_dist[w, :n, None], b.contact_pos[w, :n]], axis=1)
    kp = np.concatenate([p.contact_dist[w, :n, None], p.contact_pos[w, :n]], axis=1)
    np.testing.assert_allclose(kb[np.lexsort(kb.T)], kp[np.lexsort(kp.T)], atol=1e-10)
  np.testing.assert_allclose(b.qacc, p.qacc, rtol=1e-6, atol=1e-6)
  np.testing.assert_allclose(b.sensordata, p.sensordata, atol=1e-9)
  # ... and over a roll-out
  for o in res:
    for _ in range(100):
      o.step(1)
  np.testing.assert_allclose(b.qpos, p.qpos, atol=1e-5)


def test_moving_box_on_a_slab_equals_box_on_the_plane():
  """Corner contacts of a moving box on a terrain face are plane_box's contacts."""
  plane = robots.box_model()
  spec = Spec.from_string(robots.BOX_XML)
  spec.option.integrator = mjcf.INT_IMPLICITFAST
  spec.world.geoms.clear()  # drop the floor plane, stand the box on a slab instead
  terrains.add_boxes(spec, spec.add_body("terrain"), np.array([[0, 0, -0.5, 5, 5, 0.5]]))
  slab = spec.compile()
  assert slab.nterrain == 1 and slab.ntgeom == 1 and slab.npair == 0 and plane.npair == 1
  rng = np.random.default_rng(0)
  nw = 16
  qpos = np.tile(plane.qpos0, (nw, 1))
  qpos[:, 2] = rng.uniform(0.06, 0.16, nw)
  q = np.array([1.0, 0, 0, 0]) + rng.normal(scale=0.15, size=(nw, 4))
  qpos[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
  qvel = rng.normal(scale=0.3, size=(nw, 6))
  res = []
  for m in (slab, plane):
    o = OracleSim(m, nworld=nw)
    o.qpos[:], o.qvel[:] = qpos, qvel
    o.forward()
    res.append(o)
  b, p = res
  np.testing.assert_array_equal(b.ncon, p.ncon)
  assert b.ncon.max() == 4 and (b.ncon > 0).mean() > 0.6
  for w in range(nw):
    n = int(b.ncon[w, 0])
    np.testing.assert_allclose(b.contact_dist[w, :n], p.contact_dist[w, :n], atol=1e-12)
    np.testing.assert_allclose(b.contact_pos[w, :n], p.contact_pos[w, :n], atol=1e-12)
    np.testing.assert_allclose(b.contact_frame[w, :n, :3], -p.contact_frame[w, :n, :3], atol=1e-12)
  np.testing.assert_allclose(b.qacc, p.qacc, rtol=1e-7, atol=1e-7)
  for o in res:
    o.step(300)
  np.testing.assert_allclose(b.qpos, p.qpos, atol=1e-6)


def test_go1_rough_scene_stands():
  m = robots.load_model("go1_velocity_rough")
  assert m.nterrain == 3564 and m.npair == 0 and m.ntgeom == 30  # Go1 geoms only collide with the ground
  assert (m.geom_type[m.tgeom] == GEOM_BOX).sum() == 1  # the trunk
  nw = 10
  o = OracleSim(m, nworld=nw, njmax=300)
  o.reset(key=0)
  rng = np.random.default_rng(1)
  rows, cols = rng.integers(0, 10, nw), np.arange(nw) * 20 // nw
  o.qpos[:, :3] += m.terrain_origins[rows, cols]
  for _ in range(200):
    o.ctrl[:] = m.key_ctrl[0]
    o.step(1)
  assert np.isfinite(o.qpos).all()
  # four feet on the ground, trunk at standing height above the platform
  assert (o.sensordata.sum(axis=1) == 4).mean() > 0.8
  h = o.qpos[:, 2] - m.terrain_origins[rows, cols][:, 2]
  assert (h > 0.2).all() and (h < 0.4).all()
  # a robot dropped on its back rests on trunk corners / head, not inside the terrain
  o.reset(key=0)
  o.qpos[:, :3] += m.terrain_origins[rows, cols]
  o.qpos[:, 3:7] = [0, 1, 0, 0]
  o.step(300)
  h = o.qpos[:, 2] - m.terrain_origins[rows, cols][:, 2]
  assert np.isfinite(o.qpos).all() and (h > 0.03).all()
  tg = m.names["geom"].index("robot/trunk_collision")
  touching = [(o.contact_geom[w, : int(o.ncon[w, 0]), 0] == tg).any() for w in range(nw)]
  assert np.mean(touching) > 0.5


def test_g1_rough_scene_model():
  m = robots.load_model("g1_velocity_rough")
  assert m.nterrain == 3564 and m.ngeom == 3564 + 68 and m.nstaticgeom == 3564 and m.geom_lds0 == 3564
  assert m.ntgeom == 33 and m.npair == 469  # robot self pairs only: the terrain is reached through the grid
  assert m.terrain_origins.shape == (10, 20, 3)
  # grid covers every box footprint; items ascending inside a cell; cell0 is the lowest cell
  assert m.tgrid_start[-1] == m.ntitem == len(m.tgrid_item)
  for c in np.random.default_rng(0).integers(0, m.ntcellp1 - 1, 200):
    it = m.tgrid_item[m.tgrid_start[c] : m.tgrid_start[c + 1]]
    assert (np.diff(it) > 0).all()
    ix, iy = divmod(int(c), m.tgrid_ny)
    assert (m.tbox_cell0[it, 0] <= ix).all() and (m.tbox_cell0[it, 1] <= iy).all()
  np.testing.assert_array_equal(m.geom_type[m.tbox_geom], GEOM_BOX)


def test_g1_stands_and_walks_off_steps_on_rough_terrain():
  m = robots.load_model("g1_velocity_rough")
  nw = 12
  o = OracleSim(m, nworld=nw, njmax=300)
  o.reset(key=0)
  rng = np.random.default_rng(0)
  rows, cols = rng.integers(0, 10, nw), np.arange(nw) * 20 // nw
  o.qpos[:, :3] += m.terrain_origins[rows, cols]
  o.qpos[:, 0:2] += rng.uniform(-1.8, 1.8, (nw, 2))  # also off the central platform, onto the steps
  for _ in range(160):
    o.ctrl[:] = m.key_ctrl[0]
    o.step(1)
  assert np.isfinite(o.qpos).all() and np.isfinite(o.qvel).all()
  assert (o.ncon[:, 0] > 0).all()
  assert (o.nefc[:, 0] <= 300).all()
  # nobody fell through the terrain: pelvis above the local surface
  top = []
  for w in range(nw):
    p = o.qpos[w, :2]
    inside = np.all(np.abs(m.tbox_pos[:, :2] - p) <= m.tbox_size[:, :2], axis=1)
    top.append((m.tbox_pos[inside, 2] + m.tbox_size[inside, 2]).max())
  height = o.qpos[:, 2] - np.array(top)
  assert (height > 0.05).all() and (height > 0.5).mean() > 0.7  # a PD-only robot may sit down on a step edge
  # the foot sensors see the terrain body
  assert (o.sensordata.sum(axis=1) >= 1).mean() > 0.7


def test_capsule_box_deepest_contact_is_the_true_distance():
  """Property check against brute force: the smallest contact distance of capsule_box equals the
  true capsule-box distance (min over the axis of point-box distance, minus the radius) whenever
  the capsule does not pierce the box, and there is no contact when the capsule is clear."""
  m = probes_on([[0.0, 0.0, 0.0, 0.3, 0.2, 0.1]])  # one box at the origin, half sizes 0.3 x 0.2 x 0.1
  rng = np.random.default_rng(0)
  nw = 400
  o = OracleSim(m, nworld=nw)
  o.qpos[:, :3] = [50, 0, 5]
  o.qpos[:, 3] = 1
  d = rng.normal(size=(nw, 3))
  d /= np.linalg.norm(d, axis=1, keepdims=True)
  o.qpos[:, 7:10] = d * rng.uniform(0.1, 0.6, (nw, 1))
  q = rng.normal(size=(nw, 4))
  o.qpos[:, 10:14] = q / np.linalg.norm(q, axis=1, keepdims=True)
  o.forward()
  cap = m.names["geom"].index("cap_geom")
  half, r, s = 0.2, 0.05, np.array([0.3, 0.2, 0.1])
  axis = o.geom_xmat[:, cap].reshape(nw, 3, 3)[:, :, 2]
  centre = o.geom_xpos[:, cap]
  t = np.linspace(-1, 1, 4001)
  checked = clear = 0
  for w in range(nw):
    pts = centre[w] + np.outer(t * half, axis[w])
    dist = np.linalg.norm(np.maximum(np.abs(pts) - s, 0.0), axis=1)
    true = dist.min() - r
    n = int(o.ncon[w, 0])
    if dist.min() == 0.0:
      assert n >= 1  # axis pierces the box: the end points carry the contact
      continue
    if true > 1e-6:
      assert n == 0
      clear += 1
    elif true < -1e-6:
      assert n >= 1
      assert abs(o.contact_dist[w, :n].min() - true) < 2e-6, (w, o.contact_dist[w, :n], true)
      # every contact is a real one: its depth is the sphere-box distance at some axis point
      for k in range(n):
        assert (np.abs(dist - r - o.contact_dist[w, k]) < 1e-4).any()
      checked += 1
  assert checked > 40 and clear > 40


def _box_over(boxes) -> mjcf.Model:
  """The 0.1 m free box of robots.BOX_XML over static terrain boxes (no floor plane)."""
  spec = Spec.from_string(robots.BOX_XML)
  spec.option.integrator = mjcf.INT_IMPLICITFAST
  spec.world.geoms.clear()
  terrains.add_boxes(spec, spec.add_body("terrain"), np.asarray(boxes, dtype=np.float64))
  return spec.compile()


def test_box_face_on_a_stair_edge_and_on_a_stair_corner():
  """box_box beyond corners (the rule documented in oracle/mjoracle.c): a terrain EDGE running through the
  moving box gives two contacts along the edge, a terrain CORNER poking into a face gives one; normals
  point from the moving box into the terrain, depths are the distance to the face being entered."""
  # a step whose top face is z = 0 for x <= 0: its upper edge runs along y at x = 0
  m = _box_over([[-1.0, 0, -0.5, 1.0, 2.0, 0.5]])  # centre + half sizes
  o = OracleSim(m, nworld=3)
  pitch = np.deg2rad(25.0)
  quat = np.array([np.cos(pitch / 2), 0, np.sin(pitch / 2), 0])  # nose down towards +x: no corner reaches the tread
  # world 0: box centred over the edge, tilted, its bottom face 1 cm below the edge line
  c = np.cos(pitch)
  o.qpos[0, :3] = [0.0, 0.0, 0.1 * c - 0.01]
  o.qpos[0, 3:7] = quat
  # world 1: same, 5 cm higher: clear
  o.qpos[1, :3] = [0.0, 0.0, 0.1 * c + 0.04]
  o.qpos[1, 3:7] = quat
  # world 2: flat on the tread (x = -0.5): four corner contacts, the plane-box case
  o.qpos[2, :3] = [-0.5, 0.0, 0.095]
  o.qpos[2, 3:7] = [1, 0, 0, 0]
  o.forward()
  k = contacts(o, 0)
  assert k["n"] == 2 and int(o.ncon[1, 0]) == 0 and int(o.ncon[2, 0]) == 4
  # both contacts sit on the edge line (x = 0, z ~ 0), at y = -+0.05 (1/4 and 3/4 of the 0.2 m the edge runs inside the box)
  assert np.allclose(np.sort(k["pos"][:, 1]), [-0.05, 0.05], atol=1e-9)
  assert np.allclose(k["pos"][:, 0], k["pos"][0, 0]) and abs(k["pos"][0, 0]) < 0.01
  # normal = outward normal of the box's bottom face = the box's -z axis, pointing into the terrain (downwards)
  nz = np.array([np.sin(pitch), 0, np.cos(pitch)])
  assert np.allclose(k["frame"][:, :3], -nz, atol=1e-9)
  depth = 0.1 - (0.1 * c - 0.01) * c  # the edge point's distance above the bottom face, along the face normal
  assert np.allclose(k["dist"], -depth, atol=1e-12)
  assert (o.efc_force[0, : int(o.nefc[0, 0])] > 0).any() and o.qacc[0, 2] > -9.81  # the edge carries the box
  # a stair CORNER into the bottom face: small pillar under the box centre, box level, 8 mm into the pillar top
  m2 = _box_over([[0, 0, -0.5, 0.025, 0.025, 0.5]])
  o2 = OracleSim(m2, nworld=1)
  o2.qpos[0, :3] = [0.0, 0.0, 0.1 - 0.008]
  o2.forward()
  k2 = contacts(o2, 0)
  assert k2["n"] == 4  # the pillar's four top corners inside the bottom face (first 4 hits)
  assert np.allclose(k2["frame"][:, :3], [0, 0, -1], atol=1e-12) and np.allclose(k2["dist"], -0.008, atol=1e-12)
  assert np.allclose(np.abs(k2["pos"][:, :2]), 0.025, atol=1e-12)
  o2.step(400)
  assert abs(o2.qpos[0, 2] - 0.1) < 5e-3 and np.abs(o2.qvel).max() < 1e-2  # comes to rest on the pillar
