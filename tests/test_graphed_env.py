"""``mjlab_amd.graphed_env.GraphedRlEnv`` on the CPU: the mask-based control step (the Python body that the GPU captures into one
hipGraph, here uncaptured over the fp32 oracle) against the reference's own ``ManagerBasedRlEnv.step``, teacher-forced --
tests/_graphed_check.py.  The GPU twin (captured, over ``mjlab_amd.Simulation``) is tests/test_gpu_reference_env.py."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "tests"))

import reference_env  # noqa: E402

pytestmark = pytest.mark.skipif(reference_env.locate_reference() is None, reason="reference checkout not present")


def test_mask_based_control_step_equals_the_reference_step():
  import _graphed_check
  from _oracle_simulation import OracleSimulation

  def make(n, device, edit):
    return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=n, device=device, sim_cls=OracleSimulation, seed=11, cfg_edit=edit)

  st = _graphed_check.run(make, "cpu", num_envs=32, steps=70, capture=False)
  print(st)
  assert st["resets"] >= 32 and st["pushes"] >= 32 and st["resamples"] >= 1 and st["quiet_env_steps"] >= 500 and 0 < st["forward_steps"] < 70


def test_unsupported_terms_are_refused_loudly():
  """A term without a mask-based restatement must raise at construction, not be skipped."""
  from _oracle_simulation import OracleSimulation

  from mjlab_amd.graphed_env import GraphedRlEnv

  def edit(cfg):
    cfg.events.push_robot.is_global_time = True

  env = reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=4, device="cpu", sim_cls=OracleSimulation, cfg_edit=edit)
  with pytest.raises(NotImplementedError, match="push_robot"):
    GraphedRlEnv(env, capture=False)
