"""Box-terrain collision on the GPU vs the CPU oracle: sphere /
capsule vs static boxes through the grid broadphase, through the Simulation boundary / C ABI."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_parity import KIN, VEL, _np, _rel  # noqa: E402
from test_terrain_collision import probes_on  # noqa: E402

from mjlab_amd import robots, terrains  # noqa: E402
from oracle.oracle import OracleSim  # noqa: E402


def _sims(model, nworld, njmax=300, graph=False):
  from mjlab_amd.sim import Simulation, SimulationCfg

  return Simulation(nworld, SimulationCfg(njmax=njmax, use_graph=graph), model, "cuda:0"), OracleSim(model, nworld, njmax=njmax)


def _set(sim, ora, **fields):
  import torch

  for f, v in fields.items():
    getattr(sim.data, f)[:] = torch.from_numpy(np.asarray(v, dtype=np.float32)).cuda()
    getattr(ora, f)[:] = v


def _contacts_match(sim, ora, tol=2e-5, ftol=1e-4, ties=0):
  """Same contacts in the same order, distances within `tol`, positions and frames within tolerance -- except for TIES: the
  box rules pick the nearest face / the first candidates in order, and a point that is equidistant from two faces to 1e-7
  (same distance on both sides, different face) is decided by rounding.  `ties` = how many such contacts the caller accepts."""
  ncon = _np(sim.data.ncon).ravel()
  assert np.array_equal(ncon, ora.ncon.ravel())
  assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  gd, gp, gf, gg = (_np(getattr(sim.data, f)) for f in ("contact_dist", "contact_pos", "contact_frame", "contact_geom"))
  found, total = [], 0
  for w in range(sim.num_envs):
    n = int(ncon[w])
    total += n
    assert np.array_equal(gg[w, :n], ora.contact_geom[w, :n])  # same pairs in the same order
    assert np.abs(gd[w, :n] - ora.contact_dist[w, :n]).max(initial=0) < tol
    perr = np.abs(gp[w, :n] - ora.contact_pos[w, :n]).max(axis=1, initial=0) / max(1.0, np.abs(ora.contact_pos[w, :n]).max(initial=0))
    ferr = np.abs(gf[w, :n].reshape(n, 9) - ora.contact_frame[w, :n].reshape(n, 9)).max(axis=1, initial=0)
    for k in np.nonzero((perr >= tol) | (ferr >= ftol))[0]:
      found.append(f"world {w} contact {k} of {n} geoms {gg[w, k]}: pos gpu {gp[w, k]} oracle {ora.contact_pos[w, k]} dist gpu {gd[w, k]} oracle {ora.contact_dist[w, k]} "
                  f"normal gpu {gf[w, k].reshape(-1)[:3]} oracle {ora.contact_frame[w, k].reshape(-1)[:3]}")
  assert len(found) <= ties, "\n".join(found)


def _probe_states(t, nw, seed):
  r = np.random.default_rng(seed)
  b = t.boxes[r.integers(0, len(t.boxes), size=(nw, 2))]
  p = b[..., :3] + r.uniform(-1.05, 1.05, size=(nw, 2, 3)) * b[..., 3:]
  p[..., 2] = b[..., 2] + b[..., 5] + r.uniform(-0.02, 0.12, size=(nw, 2))
  qpos = np.zeros((nw, 14))
  qpos[:, 0:3], qpos[:, 7:10] = p[:, 0], p[:, 1]
  q = r.normal(size=(nw, 4))
  qpos[:, 3] = 1
  qpos[:, 10:14] = q / np.linalg.norm(q, axis=1, keepdims=True)
  return qpos


@pytest.mark.parametrize("kind", ["stairs", "random_grid", "dense_columns"])
def test_probes_on_random_terrain_forward_and_rollout(kind):
  if kind == "stairs":
    cfg = terrains.rough_terrains_cfg(seed=5, num_rows=3, num_cols=5)
  elif kind == "dense_columns":  # more boxes within reach than the candidate list holds
    from test_terrain_collision import dense_columns_cfg

    cfg = dense_columns_cfg()
  else:  # 0.45 m columns of random height: many small boxes per cell, contacts on column edges
    sub = terrains.BoxRandomGridTerrainCfg(grid_width=0.45, grid_height_range=(0.05, 0.2), platform_width=2.0)
    cfg = terrains.TerrainGeneratorCfg(size=(8.0, 8.0), seed=11, num_rows=2, num_cols=2, sub_terrains={"grid": sub})
  cfg.border_width = 2.0
  t = terrains.TerrainGenerator(cfg).generate()
  model = probes_on(t.boxes)
  nw = 256
  sim, ora = _sims(model, nw)
  _set(sim, ora, qpos=_probe_states(t, nw, 2), qvel=np.random.default_rng(3).normal(scale=0.2, size=(nw, 12)))
  sim.forward()
  ora.forward()
  assert ora.ncon.sum() > 150
  _contacts_match(sim, ora, ftol=1e-3)  # edge / corner normals of cm-sized offsets: 1e-4 .. 5e-4 in fp32 (5.01e-4 measured)
  # exact corner / edge configurations sit on decision boundaries of the Newton iteration count;
  # compare accelerations in the bulk and require every world to be close
  assert np.quantile(np.abs(_np(sim.data.qacc) - ora.qacc).max(axis=1) / np.maximum(1.0, np.abs(ora.qacc).max(axis=1)), 0.95) < 1e-3
  for _ in range(20):
    sim.step()
  ora.step(20)
  err = np.abs(_np(sim.data.qpos) - ora.qpos).max(axis=1)
  assert np.quantile(err, 0.95) < (5e-4 if kind == "dense_columns" else 1e-4) and np.isfinite(_np(sim.data.qpos)).all()


def _rough_states(model, nw, seed, spread=1.8):
  rng = np.random.default_rng(seed)
  qpos = np.tile(model.key_qpos[0], (nw, 1))
  rows, cols = rng.integers(0, 10, nw), np.arange(nw) * 20 // nw
  qpos[:, :3] += model.terrain_origins[rows, cols]
  qpos[:, 0:2] += rng.uniform(-spread, spread, (nw, 2))
  qpos[:, 2] -= rng.uniform(0.0, 0.04, nw)
  qpos[:, 7:] += rng.normal(scale=0.1, size=(nw, model.nq - 7))
  yaw = rng.uniform(-3.14, 3.14, nw)
  qpos[:, 3], qpos[:, 4:6], qpos[:, 6] = np.cos(yaw / 2), 0.0, np.sin(yaw / 2)
  return qpos, rng.normal(scale=0.3, size=(nw, model.nv))


def test_g1_rough_forward_all_fields():
  model = robots.load_model("g1_velocity_rough")
  nw = 64
  sim, ora = _sims(model, nw)
  qpos, qvel = _rough_states(model, nw, 0)
  _set(sim, ora, qpos=qpos, qvel=qvel, ctrl=np.tile(model.key_ctrl[0], (nw, 1)))
  sim.forward()
  ora.forward()
  assert (ora.ncon > 0).mean() > 0.8
  # frames: the foot capsules are 1 cm thin and up to ~100 m from the world origin, where fp32
  # coordinates resolve 7.6 um: the direction of a ~1 cm clamped-point -> centre vector at an edge
  # is good to ~1e-3 (same for any fp32 engine working in world coordinates)
  _contacts_match(sim, ora, tol=5e-5, ftol=3e-3)
  # quantities built from DIFFERENCES of world positions (offsets from the subtree CoM) inherit the
  # 7.6 um resolution of fp32 coordinates ~100 m from the origin: ~1e-5 relative instead of 1e-6
  far = ("cinert", "cdof", "qM")
  for f in KIN:
    assert _rel(_np(getattr(sim.data, f)), getattr(ora, f)) < (5e-5 if f in far else 2e-6), f
  for f in VEL:
    assert _rel(_np(getattr(sim.data, f)), getattr(ora, f)) < 1e-4, f
  nv = model.nv
  for w in range(nw):
    n = int(ora.nefc[w, 0])
    assert _rel(_np(sim.data.efc_J)[w].reshape(-1, nv)[:n], ora.efc_J[w].reshape(-1, nv)[:n]) < 3e-3  # rows inherit the edge-normal resolution (ftol above)
  qa, qo = _np(sim.data.qacc), ora.qacc
  assert np.quantile(np.abs(qa - qo).max(axis=1) / np.abs(qo).max(axis=1), 0.9) < 1e-3
  assert np.array_equal(_np(sim.data.sensordata), ora.sensordata.astype(np.float32))
  # terrain geoms keep the poses written at construction
  gx = _np(sim.data.geom_xpos)
  np.testing.assert_allclose(gx[:, : model.nterrain], np.broadcast_to(model.tbox_pos, (nw, model.nterrain, 3)), atol=1e-5)


def test_go1_rough_and_moving_box_match_oracle():
  """Go1 on the rough map (sphere feet, capsule legs, the trunk BOX by its corners) and a free box
  tumbling over stairs: same contacts in the same order as the oracle."""
  model = robots.load_model("go1_velocity_rough")
  nw = 64
  sim, ora = _sims(model, nw)
  rng = np.random.default_rng(8)
  qpos = np.tile(model.key_qpos[0], (nw, 1))
  rows, cols = rng.integers(0, 10, nw), np.arange(nw) * 20 // nw
  qpos[:, :3] += model.terrain_origins[rows, cols]
  qpos[:, 0:2] += rng.uniform(-1.8, 1.8, (nw, 2))
  qpos[:, 2] -= rng.uniform(0.0, 0.25, nw)  # down to the belly: trunk corners touch
  q = np.array([1.0, 0, 0, 0]) + rng.normal(scale=0.2, size=(nw, 4))
  qpos[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
  qpos[:, 7:] += rng.normal(scale=0.1, size=(nw, model.nq - 7))
  _set(sim, ora, qpos=qpos, qvel=rng.normal(scale=0.3, size=(nw, model.nv)), ctrl=np.tile(model.key_ctrl[0], (nw, 1)))
  sim.forward()
  ora.forward()
  trunk = model.names["geom"].index("robot/trunk_collision")
  assert sum((ora.contact_geom[w, : int(ora.ncon[w, 0]), 0] == trunk).any() for w in range(nw)) > 10
  _contacts_match(sim, ora, tol=5e-5, ftol=3e-3)
  assert np.array_equal(_np(sim.data.sensordata), ora.sensordata.astype(np.float32))
  qa, qo = _np(sim.data.qacc), ora.qacc
  assert np.quantile(np.abs(qa - qo).max(axis=1) / np.maximum(1.0, np.abs(qo).max(axis=1)), 0.9) < 1e-3
  for _ in range(5):
    sim.step()
  assert np.isfinite(_np(sim.data.qpos)).all()

  # free box over a small stairs map
  from mjlab_amd import mjcf

  spec = mjcf.Spec.from_string(robots.BOX_XML)
  spec.option.integrator = mjcf.INT_IMPLICITFAST
  spec.world.geoms.clear()
  cfg = terrains.rough_terrains_cfg(seed=5, num_rows=2, num_cols=5)
  cfg.border_width = 2.0
  t = terrains.TerrainGenerator(cfg).generate()
  terrains.add_boxes(spec, spec.add_body("terrain"), t.boxes)
  model = spec.compile()
  nw = 128
  sim, ora = _sims(model, nw)
  b = t.boxes[rng.integers(0, len(t.boxes), nw)]
  qpos = np.zeros((nw, 7))
  qpos[:, :3] = b[:, :3] + rng.uniform(-1.0, 1.0, (nw, 3)) * b[:, 3:]
  qpos[:, 2] = b[:, 2] + b[:, 5] + rng.uniform(0.05, 0.16, nw)
  q = rng.normal(size=(nw, 4))
  qpos[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
  _set(sim, ora, qpos=qpos, qvel=rng.normal(scale=0.2, size=(nw, 6)))
  sim.forward()
  ora.forward()
  assert (ora.ncon > 0).mean() > 0.5
  # one of the 456 contacts is a terrain-box corner inside the tumbling cube that is equidistant from two of its faces to 9e-8
  # (depth -0.0246012 on both sides): fp32 and fp64 leave through different faces
  _contacts_match(sim, ora, ties=1)


def test_g1_rough_per_world_friction_and_small_capacity():
  """Domain randomisation of the foot friction on the terrain scene (reference
  tasks/velocity/velocity_env_cfg.py:162-172 through expand_model_fields), and the capacity
  path: with room for only 16 contacts / 60 rows both sides keep the same first ones."""
  import torch

  model = robots.load_model("g1_velocity_rough")
  nw = 32
  feet = [i for i, n in enumerate(model.names["geom"]) if "foot" in n and n.endswith("_collision")]
  assert len(feet) == 14
  fr = np.random.default_rng(5).uniform(0.3, 1.2, (nw, len(feet)))
  for njmax in (300, 60):
    sim, ora = _sims(model, nw, njmax=njmax)
    qpos, qvel = _rough_states(model, nw, 6)
    _set(sim, ora, qpos=qpos, qvel=qvel, ctrl=np.tile(model.key_ctrl[0], (nw, 1)))
    sim.expand_model_fields(["geom_friction"])
    sim.model.geom_friction[:, feet, 0] = torch.from_numpy(fr.astype(np.float32)).cuda()
    ora.expand_model_field("geom_friction")[:, feet, 0] = fr
    sim.forward()
    ora.forward()
    _contacts_match(sim, ora, tol=5e-5, ftol=3e-3)
    gf = _np(sim.data.contact_friction)
    for w in range(nw):
      n = int(ora.ncon[w, 0])
      assert np.abs(gf[w, :n] - ora.contact_friction[w, :n]).max(initial=0) < 1e-6
    if njmax == 60:
      assert (ora.nefc[:, 0] <= 60).all() and (ora.nefc[:, 0] >= 56).mean() > 0.3  # capacity actually binds
    sim.step()
    assert np.isfinite(_np(sim.data.qpos)).all()


def test_g1_rough_rollout_tracks_oracle():
  model = robots.load_model("g1_velocity_rough")
  nw = 32
  sim, ora = _sims(model, nw, graph=True)
  qpos, qvel = _rough_states(model, nw, 1, spread=0.3)
  _set(sim, ora, qpos=qpos, qvel=qvel * 0.2, ctrl=np.tile(model.key_ctrl[0], (nw, 1)))
  for _ in range(10):
    sim.step()
  ora.step(10)
  err = np.abs(_np(sim.data.qpos) - ora.qpos).max(axis=1)
  assert np.quantile(err, 0.9) < 2e-4, err  # edge contacts ~100 m from the origin (see the forward test)
  assert np.isfinite(_np(sim.data.qvel)).all()


def test_g1_on_a_flat_slab_equals_g1_on_the_plane_gpu():
  """Flat sub-terrain vs ground plane on the device: same accelerations, same roll-out."""
  rough, flat = robots.load_model("g1_velocity_rough"), robots.load_model("g1_velocity_flat")
  nw = 32
  from mjlab_amd.sim import Simulation, SimulationCfg

  # columns 0..7 of the curriculum grid are flat slabs (top at z = 0)
  rng = np.random.default_rng(4)
  qpos = np.tile(flat.key_qpos[0], (nw, 1))
  qpos[:, 2] -= rng.uniform(0.0, 0.03, nw)
  qpos[:, 7:] += rng.normal(scale=0.1, size=(nw, flat.nq - 7))
  qvel = rng.normal(scale=0.3, size=(nw, flat.nv))
  origin = rough.terrain_origins[rng.integers(0, 10, nw), rng.integers(0, 8, nw)]
  assert np.all(origin[:, 2] == 0)
  import torch

  res = []
  for model, shift in ((rough, origin), (flat, 0 * origin)):
    sim = Simulation(nw, SimulationCfg(njmax=300), model, "cuda:0")
    q = qpos.copy()
    q[:, :3] += shift
    sim.data.qpos[:] = torch.from_numpy(q.astype(np.float32)).cuda()
    sim.data.qvel[:] = torch.from_numpy(qvel.astype(np.float32)).cuda()
    sim.data.ctrl[:] = torch.from_numpy(np.tile(flat.key_ctrl[0], (nw, 1)).astype(np.float32)).cuda()
    sim.forward()
    qacc = _np(sim.data.qacc).copy()
    ncon = _np(sim.data.ncon).copy()
    for _ in range(40):
      sim.step()
    q1 = _np(sim.data.qpos).copy()
    q1[:, :3] -= shift
    res.append((ncon, qacc, q1))
  (nb, ab, qb), (npl, ap, qp) = res
  assert np.array_equal(nb.ravel(), npl.ravel()) and (nb > 4).mean() > 0.7
  assert np.quantile(np.abs(ab - ap).max(axis=1) / np.abs(ap).max(axis=1), 0.9) < 2e-3
  # positions are offset by up to ~100 m on the rough map: fp32 resolution there is ~1e-5
  assert np.quantile(np.abs(qb - qp).max(axis=1), 0.9) < 2e-3


def test_rough_fullsize_rollout_is_stable():
  """4096 G1s over the whole 200-tile map, random actions, resets on: finite, above the terrain,
  contact capacity respected."""
  import torch

  from mjlab_amd.rollout import PhysicsRollout, g1_action_scale
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model("g1_velocity_rough")
  sim = Simulation(4096, SimulationCfg(njmax=300), model, "cuda:0")
  ro = PhysicsRollout(sim, g1_action_scale(model), seed=3)
  assert ro.env_origins is not None and ro.env_origins.shape == (4096, 3)
  nreset = 0
  for _ in range(60):
    nreset += int(ro.step(ro.random_action()).sum())
  torch.cuda.synchronize()
  qpos = _np(sim.data.qpos)
  assert np.isfinite(qpos).all() and np.isfinite(_np(sim.data.qvel)).all()
  assert (_np(sim.data.ncon) <= sim.nconmax).all() and (_np(sim.data.nefc) <= sim.njmax).all()
  assert (_np(sim.data.ncon) > 0).mean() > 0.6  # freshly reset robots start a few cm above the ground
  # heights above the own tile's spawn origin stay in a sane band (nobody fell through or flew off)
  rel_z = qpos[:, 2] - ro.env_origins[:, 2].cpu().numpy()
  assert (rel_z > -1.2).all() and (rel_z < 1.5).all()
  assert nreset < 4096  # not everybody falls within 1.2 s


def test_go1_trunk_resting_on_a_stair_edge_matches_oracle():
  """box_box beyond corner contacts: Go1 on its back across a stair edge -- the
  edge runs through the trunk box's face, no trunk corner touches anything; plus the free box of BOX_XML on a
  stair edge and on a pillar.  Same contacts in the same order as the oracle, the trunk is carried."""
  from mjlab_amd import mjcf

  spec = mjcf.Spec.from_string(robots.BOX_XML)
  spec.option.integrator = mjcf.INT_IMPLICITFAST
  spec.world.geoms.clear()
  terrains.add_boxes(spec, spec.add_body("terrain"), np.array([[-1.0, 0, -0.5, 1.0, 2.0, 0.5], [3.0, 0, -0.5, 0.025, 0.025, 0.5]]))
  model = spec.compile()
  sim, ora = _sims(model, 3)
  pitch = np.deg2rad(25.0)
  qpos = np.zeros((3, 7))
  qpos[0] = [0, 0, 0.1 * np.cos(pitch) - 0.01, np.cos(pitch / 2), 0, np.sin(pitch / 2), 0]  # tilted over the edge at x = 0
  qpos[1] = [-0.05, 0.03, 0.097, 1, 0, 0, 0]  # level, half over the edge: two corners + two edge points
  qpos[2] = [3.0, 0, 0.092, 1, 0, 0, 0]  # on the 5 cm pillar: the pillar's corners poke into the bottom face
  _set(sim, ora, qpos=qpos)
  sim.forward()
  ora.forward()
  assert ora.ncon.ravel().tolist() == [2, 4, 4]
  _contacts_match(sim, ora)
  assert _rel(_np(sim.data.qacc), ora.qacc) < 1e-4

  # Go1 of the rough scene, rolled onto its back and pitched 25 degrees, its trunk box laid across the upper
  # +x edge of randomly chosen terrain boxes: legs in the air, the edge runs through the trunk's (now lower) face
  model = robots.load_model("go1_velocity_rough")
  nw = 256
  sim, ora = _sims(model, nw)
  rng = np.random.default_rng(21)
  trunk = model.names["geom"].index("robot/trunk_collision")
  hz = float(model.geom_size[trunk][2])
  pos, size = np.asarray(model.tbox_pos), np.asarray(model.tbox_size)
  steps = np.flatnonzero((size[:, 0] > 0.1) & (size[:, 0] < 2.0) & (size[:, 1] > 0.3))
  b = steps[rng.integers(0, len(steps), nw)]
  qpos = np.tile(model.key_qpos[0], (nw, 1))
  qpos[:, 0] = pos[b, 0] + size[b, 0]
  qpos[:, 1] = pos[b, 1] + rng.uniform(-0.5, 0.5, nw) * size[b, 1]
  qpos[:, 2] = pos[b, 2] + size[b, 2] + hz * np.cos(pitch) - rng.uniform(0.002, 0.02, nw)
  roll = np.array([0.0, 1.0, 0.0, 0.0])  # 180 degrees about x
  pq = np.array([np.cos(pitch / 2), 0, np.sin(pitch / 2), 0])
  qpos[:, 3:7] = mjcf.quat_mul(pq, roll)
  _set(sim, ora, qpos=qpos, qvel=np.zeros((nw, model.nv)), ctrl=np.tile(model.key_ctrl[0], (nw, 1)))
  sim.forward()
  ora.forward()
  _contacts_match(sim, ora, tol=5e-5, ftol=3e-3)
  edge_worlds = 0
  for w in range(nw):
    n = int(ora.ncon[w, 0])
    on_trunk = ora.contact_geom[w, :n, 0] == trunk
    tilted = np.abs(ora.contact_frame[w, :n, 2]) < 0.99  # normal = a face normal of the pitched trunk, not the terrain's vertical
    edge_worlds += bool((on_trunk & tilted).any())
  assert edge_worlds > nw // 4, edge_worlds  # a stair edge into a trunk FACE does make contact now
  assert np.array_equal(_np(sim.data.sensordata), ora.sensordata.astype(np.float32))
