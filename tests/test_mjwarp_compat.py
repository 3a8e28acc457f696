"""The mujoco_warp-shaped facade (mjlab_amd/mjwarp_compat.py): put_model / put_data / step / forward /
expand_model_fields with the reference's call pattern (src/mjlab/sim/sim.py:107-139,182-195,
sim/randomization.py:20-55) give exactly what Simulation gives."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_facade_equals_simulation_and_accepts_mjmodel_like_input():
  import torch

  from mjlab_amd import mjwarp_compat as mjwarp
  from mjlab_amd import robots
  from mjlab_amd.sim import Simulation, SimulationCfg
  from test_from_mujoco import fake_mjmodel

  host = robots.load_model("g1_velocity_rough")
  nw = 32
  sim = Simulation(nw, SimulationCfg(njmax=300, use_graph=False, ls_parallel=True), host, "cuda:0")
  m = mjwarp.put_model(fake_mjmodel(host))
  m.opt.ls_parallel = True  # the reference sets it (sim/sim.py:111)
  d = mjwarp.put_data(fake_mjmodel(host), None, nworld=nw, nconmax=140_000, njmax=300)
  assert d.qpos.shape == (nw, host.nq) and d.xpos.shape == (nw, host.nbody, 3) and m.geom_friction.shape == (1, host.ngeom, 3)
  assert m.geom_bodyid.shape == (host.ngeom,)  # int topology fields carry no world dimension
  rng = np.random.default_rng(0)
  q = np.tile(host.key_qpos[0], (nw, 1))
  q[:, :3] += host.terrain_origins[rng.integers(0, 10, nw), rng.integers(0, 20, nw)]
  q[:, 2] -= rng.uniform(0, 0.03, nw)
  qt = torch.from_numpy(q.astype(np.float32)).cuda()
  ct = torch.from_numpy(np.tile(host.key_ctrl[0], (nw, 1)).astype(np.float32)).cuda()
  for data in (sim.data, d):
    data.qpos[:] = qt
    data.ctrl[:] = ct
    data.qacc_warmstart[:] = 0.0  # Simulation.__init__ ran a forward pass at qpos0, which left its warm start behind
  sim.forward()
  mjwarp.forward(m, d)
  for _ in range(5):
    sim.step()
    mjwarp.step(m, d)
  torch.cuda.synchronize()
  for f in ("qpos", "qvel", "qacc", "xpos", "geom_xpos", "sensordata", "ncon", "nefc"):
    assert torch.equal(getattr(sim.data, f), getattr(d, f)), f
  with pytest.raises(AttributeError):
    d.qpos = qt
  with pytest.raises(AttributeError):
    m.geom_friction = None


def test_facade_expand_model_fields():
  import torch

  from mjlab_amd import mjwarp_compat as mjwarp
  from mjlab_amd import robots

  host = robots.load_model("go1_velocity_flat")
  nw = 8
  m = mjwarp.put_model(host)
  d = mjwarp.put_data(host, None, nworld=nw, njmax=300)
  mjwarp.forward(m, d)
  before = m.geom_friction.data_ptr()
  mjwarp.expand_model_fields(m, nw, ["geom_friction"])
  assert m.geom_friction.shape == (nw, host.ngeom, 3) and m.geom_friction.data_ptr() != before and m.geom_friction.is_contiguous()
  feet = [i for i, n in enumerate(host.names["geom"]) if n.endswith("_foot_collision")]
  m.geom_friction[:, feet, 0] = torch.linspace(0.3, 1.2, nw, device="cuda")[:, None]
  d.qpos[:, 2] -= 0.05
  mjwarp.forward(m, d)
  torch.cuda.synchronize()
  want = torch.linspace(0.3, 1.2, nw, device="cuda")
  ncon, geom, mu = d.ncon.view(-1).cpu(), d.contact_geom.cpu(), d.contact_friction[:, :, 0].cpu()
  seen = 0
  for w in range(nw):
    for c in range(int(ncon[w])):
      if int(geom[w, c, 1]) in feet:  # plane-foot contacts take the foot's (higher priority) friction
        assert abs(float(mu[w, c]) - float(want[w])) < 1e-6
        seen += 1
  assert seen >= 4 * nw
  with pytest.raises(ValueError):
    mjwarp.expand_model_fields(m, nw, ["not_a_field"])
