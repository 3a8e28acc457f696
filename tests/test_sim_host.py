"""Host-side logic of the Simulation boundary that needs no GPU."""

import copy

import numpy as np
import pytest

from mjlab_amd import robots
from mjlab_amd.sim import Simulation, SimulationCfg, check_supported
from mjlab_amd.sim_data import Bridge


def test_supported_models_pass():
  for name in robots.SCENES:
    check_supported(robots.load_model(name))
  pgs = copy.deepcopy(robots.load_model("go1_velocity_flat"))
  pgs.opt.solver = 0  # mjSOL_PGS: the dual solver exists since round 3 (one kernel per stage: stage_pgs.h)
  check_supported(pgs)
  ell = copy.deepcopy(robots.load_model("g1_velocity_flat"))
  ell.opt.cone = 1  # mjCONE_ELLIPTIC: since round 5, Newton only, kernels of its own (stage_cone.h)
  check_supported(ell)
  for m in (robots.box_model(), robots.mixed_model(), robots.pendulum_model()):
    check_supported(m)


@pytest.mark.parametrize(
  "mutate, match",
  [
    (lambda m: setattr(m.opt, "solver", 3), "solver"),
    (lambda m: setattr(m.opt, "cone", 2), "cone"),
    (lambda m: (setattr(m.opt, "cone", 1), setattr(m.opt, "solver", 0)), "elliptic"),
    (lambda m: m.jnt_type.__setitem__(2, 1), "ball"),
    (lambda m: m.geom_condim.__setitem__(slice(None), 4), "condim"),
    (lambda m: m.sensor_intprm.__setitem__((0, 0), 3), "found"),
  ],
)
def test_unsupported_features_are_rejected(mutate, match):
  m = copy.deepcopy(robots.load_model("g1_velocity_flat"))
  mutate(m)
  with pytest.raises(NotImplementedError, match=match):
    check_supported(m)


def test_elliptic_cones_need_a_positive_impratio_and_accept_frictionless_geoms():
  """impratio <= 0 is refused (it used to be clamped silently); a zero sliding friction is no longer refused -- the
  collision stage and the restatement clamp contact friction at mjMINMU = 1e-5 like mj_contactParam (tests/test_oracle_elliptic.py)."""
  m = copy.deepcopy(robots.load_model("g1_velocity_flat"))
  m.opt.cone = 1
  m.geom_friction[:, 0] = 0.0
  check_supported(m)
  m.opt.impratio = 0.0
  with pytest.raises(ValueError, match="impratio"):
    check_supported(m)


def test_no_cpu_fallback():
  """The product path must fail loudly without a GPU (reference: device is always cuda, scripts/train.py:29)."""
  import torch

  if torch.cuda.is_available():
    pytest.skip("GPU present")
  with pytest.raises(RuntimeError, match="no CPU fallback"):
    Simulation(2, SimulationCfg(), robots.pendulum_model(), "cpu")


def test_bridge_is_read_only_and_views_are_stable():
  """reference tests/test_sim_data.py:62-81 semantics on the bridge class itself."""
  import torch

  t = torch.zeros(4, 3)
  b = Bridge("sim.data", {"qpos": t}, {"nworld": 4})
  with pytest.raises(AttributeError, match="read-only"):
    b.qpos = torch.ones(4, 3)
  ptr = b.qpos.data_ptr()
  b.qpos[1:3] = 5.0
  assert b.qpos.data_ptr() == ptr and float(t[1, 0]) == 5.0
  assert b.nworld == 4
  with pytest.raises(AttributeError):
    _ = b.nope
  assert np.array_equal(b.qpos.numpy()[0], np.zeros(3))


def test_nan_guard_on_host_tensors(tmp_path):
  """The guard itself is device-agnostic torch code: exercised here on CPU tensors (reference cfg fields
  utils/nan_guard.py:18-25; dump keys read by scripts/nan_viz.py)."""
  import types

  import torch

  from mjlab_amd.nan_guard import NanGuard, NanGuardCfg
  from mjlab_amd.sim import SimulationCfg

  assert SimulationCfg().nan_guard == NanGuardCfg() and not NanGuardCfg().enabled
  cfg = NanGuardCfg(enabled=True, buffer_size=3, output_dir=str(tmp_path), max_envs_to_capture=2, check_every=2)
  m = robots.pendulum_model()
  g = NanGuard(cfg, 4, m)
  data = types.SimpleNamespace(qpos=torch.zeros(4, m.nq), qvel=torch.zeros(4, m.nv), qacc=torch.zeros(4, m.nv), qacc_warmstart=torch.zeros(4, m.nv))
  for k in range(5):
    with g.watch(data):
      data.qpos += 1.0
      if k == 2:
        data.qacc[3, 0] = float("inf")  # seen at the next read-back (check_every = 2), remembered in between
  dumps = list(tmp_path.glob("nan_dump_*.npz"))
  assert len(dumps) == 1
  z = np.load(dumps[0], allow_pickle=True)
  meta = z["_metadata"].item()
  assert meta["nan_env_ids"] == [3] and meta["detection_step"] == 4
  assert sorted(k for k in z.files if k != "_metadata") == ["states_step_000001", "states_step_000002", "states_step_000003"]
  assert z["states_step_000003"][0, 0] == 3.0  # pre-step state of step 3
  off = NanGuard(NanGuardCfg(), 4, m)
  with off.watch(data):
    pass
  assert not off.check_and_dump(data)
