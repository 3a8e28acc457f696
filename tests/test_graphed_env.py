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

  st = _graphed_check.run(make, "cpu", num_envs=32, steps=70, capture=False, reset_at=30)
  print(st)
  assert st["resets"] >= 32 and st["pushes"] >= 32 and st["resamples"] >= 1 and st["quiet_env_steps"] >= 500 and 0 < st["forward_steps"] < 70


def test_wrapper_surface_and_replaced_episode_length_buffer():
  """GraphedRlEnv stands where the reference's RslRlVecEnvWrapper expects an environment (``unwrapped``, attribute delegation), and
  a caller that REPLACES ``episode_length_buf`` (the wrapper's setter, as rsl_rl's init_at_random_ep_len does) does not detach the
  step from it."""
  import torch
  from _oracle_simulation import OracleSimulation

  from mjlab_amd.graphed_env import GraphedRlEnv

  env = reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=8, device="cpu", sim_cls=OracleSimulation, seed=3)
  g = GraphedRlEnv(env, capture=False)
  assert g.unwrapped is env and g.num_envs == 8 and g.max_episode_length == env.max_episode_length and g.action_manager is env.action_manager
  g.reset()
  buf = env.episode_length_buf
  env.episode_length_buf = torch.full((8,), 7, dtype=buf.dtype)  # what `wrapper.episode_length_buf = ...` does
  g.step(torch.zeros(8, 29))
  assert env.episode_length_buf is buf and bool((buf == 8).all())
  # the property caches live inside the step body only: an eager reset() between steps (the reference's own code, reading
  # robot.data.* through the cache's proxy) sees the state it has just written, not tensors of the last step
  for _ in range(3):
    g.step(torch.rand(8, 29) * 2 - 1)
  robot = env.scene["robot"]
  before = robot.data.root_link_pos_w.clone()
  obs, _ = g.reset()
  root = int(robot.indexing.root_body_id)
  assert torch.equal(robot.data.root_link_pos_w, env.sim.data.xpos[:, root]) and not torch.equal(robot.data.root_link_pos_w, before)
  fresh = env.observation_manager.compute()
  for grp in obs:
    if grp == "critic":  # (the policy group draws new noise at every call)
      assert torch.equal(obs[grp], fresh[grp]), grp
  jp = robot.data.joint_pos - robot.data.default_joint_pos
  assert torch.equal(obs["critic"][:, 9 : 9 + 29], jp)  # base_lin_vel 3, base_ang_vel 3, projected_gravity 3, then joint_pos


def test_unsupported_terms_are_refused_loudly():
  """A term without a mask-based restatement must raise at construction, not be skipped."""
  from _oracle_simulation import OracleSimulation

  from mjlab_amd.graphed_env import GraphedRlEnv

  def edit(cfg):
    cfg.events.push_robot.is_global_time = True

  env = reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=4, device="cpu", sim_cls=OracleSimulation, cfg_edit=edit)
  with pytest.raises(NotImplementedError, match="push_robot"):
    GraphedRlEnv(env, capture=False)


_TRACKING = """
import json, sys
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env, _graphed_check
from _motion_fixture import write_full_motion
from _oracle_simulation import OracleSimulation
write_full_motion({motion!r})
def make(n, device, edit):
  def both(cfg):
    cfg.commands.motion.motion_file = {motion!r}
    edit(cfg)
  return reference_env.make_env("Mjlab-Tracking-Flat-Unitree-G1", num_envs=n, device=device, sim_cls=OracleSimulation, seed=7, cfg_edit=both)
print("RESULT " + json.dumps(_graphed_check.run_tracking(make, "cpu", num_envs=32, steps=40, capture=False)))
"""


def test_mask_based_control_step_of_the_tracking_task(tmp_path):
  """The second task north_star names: ``MotionCommand`` (adaptive sampling of the reset phase, resampling when a motion ends,
  body-relative targets) behind GraphedRlEnv, against the reference's own step.  Own process (the tracking task's configs)."""
  import json
  import subprocess

  code = _TRACKING.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"), motion=str(tmp_path / "motion.npz"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print(st)
  assert st["resets"] >= 8 and st["ended"] >= 8 and st["pushes"] >= 16 and st["quiet_env_steps"] >= 400


def test_mask_based_control_step_of_the_tracking_task_without_state_estimation(tmp_path):
  """``Mjlab-Tracking-Flat-Unitree-G1-No-State-Estimation`` (reference tasks/tracking/config/g1/__init__.py:24):
  the tracking task whose policy group drops ``motion_anchor_pos_b`` and ``base_lin_vel`` -- the shared observation terms and the
  group assembly see another layout."""
  import json
  import subprocess

  code = _TRACKING.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"), motion=str(tmp_path / "motion.npz")).replace(
    '"Mjlab-Tracking-Flat-Unitree-G1"', '"Mjlab-Tracking-Flat-Unitree-G1-No-State-Estimation"')
  assert "No-State-Estimation" in code
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print(st)
  assert st["resets"] >= 8 and st["ended"] >= 8 and st["pushes"] >= 16 and st["quiet_env_steps"] >= 400


_GO1 = """
import json, sys
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env, _graphed_check
from _oracle_simulation import OracleSimulation
def make(n, device, edit):
  def both(cfg):
    edit(cfg)
    cfg.curriculum.command_vel.params["velocity_stages"] = [dict(step=30, range=(-3.0, 3.0))]  # the stage switches inside the run
  return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-Go1", num_envs=n, device=device, sim_cls=OracleSimulation, seed=11, cfg_edit=both)
st = _graphed_check.run(make, "cpu", num_envs=16, steps=60, capture=False)
print("RESULT " + json.dumps(st))
"""


def test_go1_task_with_its_host_side_curriculum():
  """``Mjlab-Velocity-Flat-Unitree-Go1`` keeps the ``commands_vel`` curriculum (widens the command ranges at the first reset after a
  step threshold): GraphedRlEnv keeps the ranges in device tensors and applies the same rule on the device -- bit for bit across the
  switch at step 30."""
  import json
  import subprocess

  code = _GO1.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  assert st["resets"] >= 16 and st["pushes"] >= 16 and st["quiet_env_steps"] >= 400


_ROUGH = """
import json, sys
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env, _graphed_check
from _oracle_simulation import OracleSimulation
def make(n, device, edit):
  return reference_env.make_env("Mjlab-Velocity-Rough-Unitree-G1", num_envs=n, device=device, sim_cls=OracleSimulation, seed=11, cfg_edit=edit)
def post(env):
  # "walked far enough" after 0.3 m instead of half a sub-terrain (4 m), so that the 30-step episodes of the check move up as well as
  # down (the curriculum reads the size at every call)
  t = env.scene.terrain
  g = t.cfg.terrain_generator
  g.size = (0.6, g.size[1])
  t.terrain_levels[:4] = t.max_terrain_level - 1  # four environments start on the hardest row: moving up from there draws a random row
  t.env_origins[:] = t.terrain_origins[t.terrain_levels, t.terrain_types]
st = _graphed_check.run(make, "cpu", num_envs=16, steps=70, capture=False, post_make=post)
print("RESULT " + json.dumps(st))
"""


def test_rough_task_with_its_terrain_curriculum():
  """``Mjlab-Velocity-Rough-Unitree-G1``: the reference's terrain generator (box stairs, 10 x 20 sub-terrains) through the MjSpec shim,
  and the ``terrain_levels_vel`` curriculum mask based -- the same level moves and spawn origins as the reference's eager step."""
  import json
  import subprocess

  code = _ROUGH.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  assert st["resets"] >= 16 and st["pushes"] >= 16 and st["quiet_env_steps"] >= 400 and st["level_moves"] >= 8 and st["level_draws"] >= 1, st


def test_go1_rough_task_with_both_curricula():
  """``Mjlab-Velocity-Rough-Unitree-Go1`` (reference tasks/velocity/config/go1/__init__.py:4): the Go1 on
  the generated stairs -- a moving BOX (the trunk) against terrain boxes in the physics, ``terrain_levels_vel`` AND the Go1's
  ``commands_vel`` curriculum in the managers."""
  import json
  import subprocess

  code = _ROUGH.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests")).replace('"Mjlab-Velocity-Rough-Unitree-G1"', '"Mjlab-Velocity-Rough-Unitree-Go1"')
  assert "Rough-Unitree-Go1" in code
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print(st)
  assert st["resets"] >= 16 and st["pushes"] >= 16 and st["quiet_env_steps"] >= 400 and st["level_moves"] >= 4, st


def test_forward_on_the_reset_worlds_only_is_the_reference_on_those_worlds_and_the_last_step_on_the_others():
  """``GraphedRlEnv(forward="reset_worlds")`` (opt-in): in a step where some environment resets, the worlds that reset
  hold what the reference's all-worlds ``forward()`` gives them, the others what the physics step left (what the reference shows in a
  step without resets); state, rewards and terminations are the reference's in every world."""
  import torch
  from _oracle_simulation import OracleSimulation

  from mjlab_amd.graphed_env import GraphedRlEnv

  def edit(cfg):
    for group in ("policy", "critic"):
      getattr(cfg.observations, group).enable_corruption = False
    cfg.episode_length_s = 0.2  # time-outs after 10 control steps

  def make():
    env = reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=8, device="cpu", sim_cls=OracleSimulation, seed=5, cfg_edit=edit)
    env.reset()
    return env

  torch.manual_seed(0)
  a, b = make(), make()
  ga, gb = GraphedRlEnv(a, capture=False), GraphedRlEnv(b, capture=False, forward="reset_worlds")
  with pytest.raises(ValueError):
    GraphedRlEnv(b, capture=False, forward="sometimes")
  a.episode_length_buf[:4] = 5  # half of the environments time out five steps before the others
  b.episode_length_buf[:4] = 5
  seen = 0
  import _graphed_check

  for k in range(12):
    _graphed_check._sync(a, b)  # (teacher forced: the reference's all-worlds forward() also refreshes qacc_warmstart of the worlds that did not reset)
    act = torch.rand(8, 29) * 2 - 1
    torch.manual_seed(100 + k)
    oa, ra, ta, toa, _ = ga.step(act.clone())
    torch.manual_seed(100 + k)
    ob, rb, tb, tob, _ = gb.step(act.clone())
    assert torch.equal(ra, rb) and torch.equal(ta, tb) and torch.equal(toa, tob)
    assert torch.equal(a.sim.data.qpos, b.sim.data.qpos) and torch.equal(a.sim.data.qvel, b.sim.data.qvel)
    reset = ta | toa
    if reset.any() and not reset.all():
      seen += 1
      assert torch.equal(a.sim.data.xpos[reset], b.sim.data.xpos[reset]) and torch.equal(oa["critic"][reset], ob["critic"][reset])
      quiet = ~reset
      assert not torch.equal(a.sim.data.xpos[quiet], b.sim.data.xpos[quiet])  # the reference moved them to the post-step pose
    elif not reset.any():
      assert torch.equal(a.sim.data.xpos, b.sim.data.xpos) and torch.equal(oa["critic"], ob["critic"])
  assert seen >= 1


def test_stock_event_terms_without_a_restatement_run_generically():
  """``reset_scene_to_default`` (reset) and ``apply_external_force_torque`` (interval), envs/mdp/events.py:27-171: the reference's
  functions on all environments, kept where the mask is set -- teacher-forced against the eager reference (the deterministic reset term
  bit for bit, the random wrenches in range, every untouched environment bit for bit)."""
  import _graphed_check
  from _oracle_simulation import OracleSimulation

  def make(n, device, edit):
    return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=n, device=device, sim_cls=OracleSimulation, seed=13, cfg_edit=edit)

  st = _graphed_check.run_generic_events(make, "cpu", num_envs=32, steps=60, capture=False)
  print(st)
  assert st["resets"] >= 32 and st["wrenches"] >= 100 and st["default_states_compared"] >= 20 and st["quiet_env_steps"] >= 100


def test_command_term_of_another_class_runs_generically():
  """A CommandTerm subclass GraphedRlEnv has no restatement for (managers/command_manager.py:19-84 is its whole contract): its own
  ``_resample_command`` on all environments, kept where the mask is set; ``_update_metrics`` / ``_update_command`` as they are."""
  import _graphed_check
  from _oracle_simulation import OracleSimulation

  def make(n, device, edit):
    return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=n, device=device, sim_cls=OracleSimulation, seed=17, cfg_edit=edit)

  st = _graphed_check.run_toy_command(make, "cpu", num_envs=32, steps=60, capture=False)
  print(st)
  assert st["resets"] >= 32 and st["resamples"] >= 60 and st["quiet_env_steps"] >= 300
