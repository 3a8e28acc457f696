"""NaN guard of Simulation.step -- the cases of the reference's tests/test_nan_guard.py
(:46-55 disabled by default, :58-106 capture + dump, :109-143 env ids, :146-176 model saved,
:179-215 complex model: Go1 rough) against this package's Simulation."""

import tempfile
from pathlib import Path

import numpy as np
import pytest

from mjlab_amd import mjcf, robots
from mjlab_amd.nan_guard import NanGuard, NanGuardCfg

SIMPLE_XML = """
<mujoco>
  <worldbody>
    <body>
      <inertial pos="0 0 0" mass="8" diaginertia="0.0533 0.0533 0.0533"/>
      <freejoint/>
      <geom type="box" size="0.1 0.1 0.1"/>
    </body>
  </worldbody>
</mujoco>
"""


def simple_model() -> mjcf.Model:
  return mjcf.Spec.from_string(SIMPLE_XML).compile()


def test_disabled_guard_is_inert():
  g = NanGuard(NanGuardCfg(), 4, simple_model())
  assert not g.enabled
  with g.watch(None):  # touches nothing when disabled
    pass
  assert g.check_and_dump(None) is False


def _sim(model, num_envs, **kw):
  from mjlab_amd.sim import Simulation, SimulationCfg

  return Simulation(num_envs, SimulationCfg(nan_guard=NanGuardCfg(enabled=True, **kw)), model, "cuda:0")


@pytest.mark.gpu
def test_nan_guard_disabled_by_default():
  from mjlab_amd.sim import Simulation, SimulationCfg

  sim = Simulation(2, SimulationCfg(), simple_model(), "cuda:0")
  assert not sim.nan_guard.enabled
  sim.step()
  sim.close()


@pytest.mark.gpu
def test_nan_guard_captures_and_dumps_on_nan():
  with tempfile.TemporaryDirectory() as tmpdir:
    sim = _sim(simple_model(), 4, buffer_size=5, output_dir=tmpdir, max_envs_to_capture=2)
    for _ in range(3):
      sim.step()
    sim.data.qpos[1, 0] = float("nan")
    sim.step()
    dump_files = list(Path(tmpdir).glob("nan_dump_*.npz"))
    assert len(dump_files) == 1
    dump = np.load(dump_files[0], allow_pickle=True)
    metadata = dump["_metadata"].item()
    assert metadata["num_envs_total"] == 4 and metadata["num_envs_captured"] == 2
    assert 1 in metadata["nan_env_ids"]
    assert metadata["buffer_size"] == 4  # 3 clean steps + the one with the NaN injected
    for k in range(4):
      assert f"states_step_{k:06d}" in dump
    state = dump["states_step_000000"]
    assert state.shape == (2, 7 + 6) and metadata["state_size"] == 13
    assert np.isnan(dump["states_step_000003"][1, 0])  # the injected state is the last one captured
    # only one dump per run
    sim.step()
    assert len(list(Path(tmpdir).glob("nan_dump_*.npz"))) == 1


@pytest.mark.gpu
def test_nan_guard_detects_correct_env_ids():
  with tempfile.TemporaryDirectory() as tmpdir:
    sim = _sim(simple_model(), 10, buffer_size=5, output_dir=tmpdir)
    for _ in range(3):
      sim.step()
    sim.data.qpos[2, 0] = float("nan")
    sim.data.qvel[5, 1] = float("nan")
    sim.data.qvel[7, 2] = float("inf")
    sim.step()
    dump = np.load(next(Path(tmpdir).glob("nan_dump_*.npz")), allow_pickle=True)
    assert set(dump["_metadata"].item()["nan_env_ids"]) == {2, 5, 7}


@pytest.mark.gpu
def test_nan_guard_saves_model_simple_and_go1_rough():
  for model in (simple_model(), robots.load_model("go1_velocity_rough")):
    with tempfile.TemporaryDirectory() as tmpdir:
      sim = _sim(model, 2, buffer_size=3, output_dir=tmpdir)
      for _ in range(2):
        sim.step()
      sim.data.qpos[0, 0] = float("nan")
      sim.step()
      dump = np.load(next(Path(tmpdir).glob("nan_dump_*.npz")), allow_pickle=True)
      path = Path(tmpdir) / dump["_metadata"].item()["model_file"]
      assert path.exists()
      loaded = mjcf.Model.load(path)
      assert loaded.nq == model.nq and loaded.nv == model.nv and loaded.ngeom == model.ngeom
