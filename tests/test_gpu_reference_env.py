"""Row N1 on the MI355X: the reference's registered task ``Mjlab-Velocity-Flat-Unitree-G1`` -- its own ManagerBasedRlEnv, Scene,
Entity, managers and MDP terms, unmodified -- stepped over ``mjlab_amd.sim.Simulation`` ("do this" item 3).

The reference source cannot be committed and ``/root/reference`` does not exist on the GPU box: ``tools/stage_reference.sh``
copies it next to the repository (``gpurun_ref/``, git-ignored) for the duration of one ``gpurun`` call; without it the test
skips (the same stack runs over the CPU oracle in tests/test_reference_env.py every round).  The log of the staged run is
committed under profiles/."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))

import reference_env  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(reference_env.locate_reference() is None, reason="reference source not staged (tools/stage_reference.sh)")]


def test_registered_g1_velocity_task_runs_over_the_hip_simulation():
  import torch

  from mjlab_amd.sim import Simulation

  env = reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=256, device="cuda:0")
  assert isinstance(env.sim, Simulation) and env.sim.num_envs == 256
  assert env.sim.host_model.nstaticsite == 256  # the reference's env-origin marker sites: posed once, not 256 x 256 per step
  d = env.sim.data
  ptr = {f: getattr(d, f).data_ptr() for f in ("qpos", "qvel", "ctrl", "xfrc_applied", "sensordata", "xpos", "cvel")}
  resets, finite = [], []

  def on_step(k, obs, rew, terminated, time_out):
    resets.append(int((terminated | time_out).sum()))
    finite.append(bool(torch.isfinite(rew).all()) and all(bool(torch.isfinite(o).all()) for o in obs.values()))

  out = reference_env.random_rollout(env, 100, seed=1, on_step=on_step)
  torch.cuda.synchronize()
  assert all(finite), "non-finite observations / rewards"
  for name, o in out["obs"].items():
    assert o.shape == (256, 99) and o.is_cuda, name
  assert {f: getattr(d, f).data_ptr() for f in ptr} == ptr  # in-place writes only: graph-safe (sim_data.py:181-185)
  assert sum(resets) > 0, "no env terminated in 100 steps of random actions"
  assert env.sim.overflow_report() == {"nconmax": 0, "njmax": 0, "terrain_candidates": 0}
  # the robots stand on the plane or are being reset: nobody fell through it or flew away
  z = d.qpos[:, 2]
  assert bool(((z > 0.05) & (z < 1.2)).all()), (float(z.min()), float(z.max()))
  # per-env friction drawn by the reference's startup event reached the device tables
  fr = env.sim.model.geom_friction[:, :, 0]
  assert float(fr.std()) > 0.0
  print(f"reference G1 velocity task over mjlab_amd.Simulation: 100 env steps x 256 envs, {sum(resets)} resets, mean reward {out['mean_reward']:.4f}")


_ELLIPTIC_GPU = """
import json, sys
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import torch
import reference_env
from mjlab_amd.graphed_env import GraphedRlEnv
def edit(cfg):
  cfg.sim.mujoco.cone = "elliptic"  # reference sim/sim.py:49,52
env = reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=256, device="cuda:0", cfg_edit=edit)
st = {{"cone": int(env.sim.mj_model.opt.cone), "fuse": env.sim.fuse}}
out = reference_env.random_rollout(env, 40, seed=1)
torch.cuda.synchronize()
d = env.sim.data
st["eager_finite"] = all(bool(torch.isfinite(o).all()) for o in out["obs"].values()) and bool(torch.isfinite(d.qpos).all())
st["elliptic_rows"] = int((d.efc_type == 7).sum())
g = GraphedRlEnv(env)  # the whole control step as one hipGraph: k_control_step_cone inside
gen = torch.Generator(device="cuda:0"); gen.manual_seed(3)
fin, resets = True, 0
for k in range(60):
  obs, rew, term, tout, _ = g.step(2.0 * torch.rand((256, 29), device="cuda:0", generator=gen) - 1.0)
  fin = fin and bool(torch.isfinite(rew).all()) and all(bool(torch.isfinite(o).all()) for o in obs.values())
  resets += int((term | tout).sum())
torch.cuda.synchronize()
st["graphed_finite"], st["graphed_resets"], st["graph"] = fin, resets, g.graph is not None
z = d.qpos[:, 2]
st["z_range"] = [float(z.min()), float(z.max())]
st["overflow"] = env.sim.overflow_report()
print("RESULT " + json.dumps(st))
"""


def test_velocity_task_with_elliptic_cones_eager_and_as_one_graph():
  """``MujocoCfg(cone="elliptic")`` on the registered G1 velocity task over the HIP simulation: the reference's eager ``env.step`` (the
  fused cone kernels under ``sim.step``) and ``GraphedRlEnv`` (``k_control_step_cone`` inside the captured control step)."""
  import json
  import subprocess

  code = _ELLIPTIC_GPU.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print("reference G1 velocity task with elliptic cones:", st)
  assert st["cone"] == 1 and st["fuse"] == "step" and st["eager_finite"] and st["elliptic_rows"] >= 3 * 128
  assert st["graph"] and st["graphed_finite"] and st["graphed_resets"] > 0
  assert 0.05 < st["z_range"][0] and st["z_range"][1] < 1.2 and st["overflow"] == {"nconmax": 0, "njmax": 0, "terrain_candidates": 0}, st


def test_graphed_env_matches_the_reference_env():
  """the whole control step of the reference's environment as ONE hipGraph (mjlab_amd/graphed_env.py) against the
  reference's own eager ``env.step`` over the same Simulation class, teacher-forced (tests/_graphed_check.py): terminations and
  rewards bit for bit in every environment, observations / state / commands bit for bit in every environment that drew no random
  number in the step, the others in distribution against the task's event and command configuration."""
  sys.path.insert(0, str(ROOT / "tests"))
  import _graphed_check

  def make(n, device, edit):
    return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=n, device=device, seed=11, cfg_edit=edit)

  st = _graphed_check.run(make, "cuda:0", num_envs=256, steps=70, capture=True, reset_at=30)
  print("graphed env vs reference env:", st)
  assert st["graph"] and st["resets"] >= 256 and st["pushes"] >= 256 and st["resamples"] >= 8 and st["quiet_env_steps"] >= 4000 and 0 < st["forward_steps"] < 70


def test_fused_environment_terms_equal_their_torch_restatements():
  """mjlab_amd/env_terms.py (one HIP launch per event / command term) inside the captured step against the torch restatements of
  the same terms, on the same uniforms, for EVERY environment -- resets, command resampling and pushes included."""
  sys.path.insert(0, str(ROOT / "tests"))
  import _graphed_check

  def make(n, device, edit):
    return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=n, device=device, seed=11, cfg_edit=edit)

  st = _graphed_check.run_fused_vs_torch(make, "cuda:0", num_envs=256, steps=60)
  print("fused environment terms vs torch restatements:", st)
  w = st["worst"]
  assert st["resets"] >= 200 and st["pushes"] >= 200 and st["noise_checks"] == 60
  assert w["qpos"] <= 2e-6 and w["qvel"] <= 1e-5 and w["command"] <= 1e-6 and w["time_left"] <= 1e-6 and w["obs"] <= 1e-5, w


def test_entity_data_from_the_fused_read_back_equals_the_reference_chains():
  """SURVEY 8f row 1 inside the reference's own environment: every ``EntityData`` base quantity of a phase from ONE
  ``mjlab_entity_readback`` launch (GraphedRlEnv's default on the GPU) against the reference's own chains of small kernels
  (``fused_entity_data=False`` on the other side, which also keeps the torch restatements of the terms): teacher-forced, terminations,
  rewards and observations equal bit for bit, state to the push term's 1 ulp."""
  sys.path.insert(0, str(ROOT / "tests"))
  import _graphed_check

  def make(n, device, edit):
    return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=n, device=device, seed=11, cfg_edit=edit)

  st = _graphed_check.run_fused_vs_torch(make, "cuda:0", num_envs=256, steps=60, a_kwargs={"fused_entity_data": True}, b_kwargs={"fused_entity_data": False})
  print("EntityData from the fused read-back vs the reference's chains:", st)
  w = st["worst"]
  assert st["resets"] >= 200 and st["pushes"] >= 200
  assert w["qpos"] == 0.0 and w["qvel"] <= 1e-5 and w["command"] == 0.0 and w["obs"] == 0.0, w


_TRACKING_GPU = """
import json, sys
import torch
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env
from _motion_fixture import write_full_motion
write_full_motion({motion!r})
def edit(cfg):
  cfg.commands.motion.motion_file = {motion!r}
env = reference_env.make_env("Mjlab-Tracking-Flat-Unitree-G1", num_envs=256, device="cuda:0", cfg_edit=edit)
from mjlab_amd.sim import Simulation
assert isinstance(env.sim, Simulation)
fin = []
out = reference_env.random_rollout(env, 60, seed=2, on_step=lambda k, o, r, t, to: fin.append(bool(torch.isfinite(r).all()) and all(bool(torch.isfinite(x).all()) for x in o.values())))
torch.cuda.synchronize()
cmd = env.command_manager.get_term("motion")
z = env.sim.data.qpos[:, 2]
print("RESULT " + json.dumps({{"finite": all(fin), "resets": out["resets"], "obs": {{k: list(v.shape) for k, v in out["obs"].items()}}, "zmin": float(z.min()), "zmax": float(z.max()),
      "overflow": env.sim.overflow_report(), "phase_max": int(cmd.time_steps.max()), "mean_reward": out["mean_reward"], "sensor": float(env.sim.data.sensordata.abs().max())}}))
"""


def test_registered_g1_tracking_task_runs_over_the_hip_simulation(tmp_path):
  """BASELINE config 4's task, Mjlab-Tracking-Flat-Unitree-G1, unmodified over ``mjlab_amd.sim.Simulation`` on the MI355X (own
  process: the reference's task configs share mutable defaults between tasks).  The motion file is synthetic
  (tests/_motion_fixture.py: the oracle's forward kinematics of mjlab_amd.rollout.synthetic_motion)."""
  import json
  import subprocess

  code = _TRACKING_GPU.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"), motion=str(tmp_path / "motion.npz"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  res = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  assert res["finite"] and res["resets"] > 0 and res["obs"] == {"policy": [256, 160], "critic": [256, 286]}
  assert res["overflow"] == {"nconmax": 0, "njmax": 0, "terrain_candidates": 0} and 0.0 < res["zmin"] and res["zmax"] < 1.3 and res["phase_max"] < 500
  print(f"reference G1 tracking task over mjlab_amd.Simulation: 60 env steps x 256 envs, {res['resets']} resets, mean reward {res['mean_reward']:.4f}")


_TRACKING_GRAPHED = """
import json, sys
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env, _graphed_check
from _motion_fixture import write_full_motion
write_full_motion({motion!r})
def make(n, device, edit):
  def both(cfg):
    cfg.commands.motion.motion_file = {motion!r}
    edit(cfg)
  return reference_env.make_env("Mjlab-Tracking-Flat-Unitree-G1", num_envs=n, device=device, seed=7, cfg_edit=both)
print("RESULT " + json.dumps(_graphed_check.run_tracking(make, "cuda:0", num_envs=128, steps=40, capture=True)))
"""


def test_graphed_tracking_env_matches_the_reference_env(tmp_path):
  """The tracking task's whole control step as one hipGraph (``MotionCommand`` restated mask based: adaptive phase sampling by inverse
  CDF, resampling when a motion ends) against the reference's eager ``env.step``, teacher-forced, on the MI355X."""
  import json
  import subprocess

  code = _TRACKING_GRAPHED.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"), motion=str(tmp_path / "motion.npz"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print("graphed tracking env vs reference env:", st)
  assert st["graph"] and st["resets"] >= 32 and st["ended"] >= 32 and st["pushes"] >= 64 and st["quiet_env_steps"] >= 2000  # measured: 71 / 108 / 339 / 4618
  assert st["relative_rounding"] >= 8, st  # the relative body poses came from the one launch, in a rounding calibrated against the reference's chain


def test_graphed_tracking_env_without_state_estimation(tmp_path):
  """``Mjlab-Tracking-Flat-Unitree-G1-No-State-Estimation`` (reference tasks/tracking/config/g1/__init__.py:24)
  as one hipGraph, teacher-forced against the reference's eager step: the policy group without ``motion_anchor_pos_b`` / ``base_lin_vel``."""
  import json
  import subprocess

  code = _TRACKING_GRAPHED.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"), motion=str(tmp_path / "motion.npz")).replace(
    '"Mjlab-Tracking-Flat-Unitree-G1"', '"Mjlab-Tracking-Flat-Unitree-G1-No-State-Estimation"')
  assert "No-State-Estimation" in code
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print("graphed tracking env (no state estimation) vs reference env:", st)
  assert st["graph"] and st["resets"] >= 32 and st["ended"] >= 32 and st["pushes"] >= 64 and st["quiet_env_steps"] >= 2000


_GO1_GRAPHED = """
import json, sys
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env, _graphed_check
import mjlab_amd.graphed_env as ge
ncap = [0]
orig = ge.GraphedRlEnv.capture
def counting(self, warmup=2):
  ncap[0] += 1
  return orig(self, warmup)
ge.GraphedRlEnv.capture = counting
def make(n, device, edit):
  def both(cfg):
    edit(cfg)
    cfg.curriculum.command_vel.params["velocity_stages"] = [dict(step=30, range=(-3.0, 3.0))]
  return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-Go1", num_envs=n, device=device, seed=11, cfg_edit=both)
st = _graphed_check.run(make, "cuda:0", num_envs=128, steps=60, capture=True)
st["captures"] = ncap[0]
print("RESULT " + json.dumps(st))
"""


def test_graphed_go1_env_with_its_curriculum(tmp_path):
  """The Go1 velocity task (BASELINE config 2's robot) as one hipGraph, its ``commands_vel`` curriculum switching the command ranges
  after step 30 INSIDE the graph (ranges in device tensors, the reference's rule -- first reset after the threshold -- evaluated on
  the device): captured once, and the comparison with the reference's eager step holds bit for bit across the switch."""
  import json
  import subprocess

  code = _GO1_GRAPHED.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print("graphed Go1 env vs reference env:", st)
  assert st["graph"] and st["captures"] == 1 and st["resets"] >= 64 and st["pushes"] >= 64 and st["quiet_env_steps"] >= 2000


_ROUGH_GRAPHED = """
import json, sys
import torch
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env, _graphed_check
def make(n, device, edit):
  return reference_env.make_env("Mjlab-Velocity-Rough-Unitree-G1", num_envs=n, device=device, seed=11, cfg_edit=edit)
def post(env):
  t = env.scene.terrain
  g = t.cfg.terrain_generator
  g.size = (0.6, g.size[1])  # "walked far enough" after 0.3 m: the 30-step episodes of the check move up as well as down
  t.terrain_levels[:16] = t.max_terrain_level - 1  # moving up from the hardest row draws a random row
  t.env_origins[:] = t.terrain_origins[t.terrain_levels, t.terrain_types]
st = _graphed_check.run(make, "cuda:0", num_envs=128, steps=70, capture=True, post_make=post)
env = make(256, "cuda:0", None)
env.reset()
z = []
for k in range(60):
  out = env.step(torch.rand((256, 29), device="cuda:0") * 2 - 1)
  z.append(bool(torch.isfinite(out[1]).all()))
st["eager_finite"] = all(z)
st["overflow"] = env.sim.overflow_report()
st["ngeom"] = int(env.sim.mj_model.ngeom)
print("RESULT " + json.dumps(st))
"""


def test_graphed_rough_env_with_its_terrain_curriculum(tmp_path):
  """``Mjlab-Velocity-Rough-Unitree-G1`` -- the terrain the reference's own generator builds (box stairs, 10 x 20 sub-terrains) through
  the MjSpec shim, over the HIP simulation -- as one hipGraph with the ``terrain_levels_vel`` curriculum mask based: the same level moves
  and spawn origins as the reference's eager step (tests/_graphed_check.py)."""
  import json
  import subprocess

  code = _ROUGH_GRAPHED.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print("graphed rough G1 env vs reference env:", st)
  assert st["graph"] and st["resets"] >= 128 and st["pushes"] >= 128 and st["quiet_env_steps"] >= 2000 and st["level_moves"] >= 64 and st["level_draws"] >= 1
  assert st["eager_finite"] and st["overflow"] == {"nconmax": 0, "njmax": 0, "terrain_candidates": 0}, st


def test_graphed_go1_rough_env_with_both_curricula(tmp_path):
  """``Mjlab-Velocity-Rough-Unitree-Go1`` (reference tasks/velocity/config/go1/__init__.py:4) as one hipGraph:
  the Go1's trunk BOX against the generated stairs in the physics, ``terrain_levels_vel`` and ``commands_vel`` inside the graph."""
  import json
  import subprocess

  code = _ROUGH_GRAPHED.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests")).replace('"Mjlab-Velocity-Rough-Unitree-G1"', '"Mjlab-Velocity-Rough-Unitree-Go1"').replace(
    "(256, 29)", "(256, 12)")
  assert "Rough-Unitree-Go1" in code
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print("graphed rough Go1 env vs reference env:", st)
  assert st["graph"] and st["resets"] >= 128 and st["pushes"] >= 128 and st["quiet_env_steps"] >= 2000 and st["level_moves"] >= 32
  assert st["eager_finite"] and st["overflow"] == {"nconmax": 0, "njmax": 0, "terrain_candidates": 0}, st


_TRACKING_FUSED = """
import json, sys
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env, _graphed_check
from _motion_fixture import write_full_motion
write_full_motion({motion!r})
def make(n, device, edit):
  def both(cfg):
    cfg.commands.motion.motion_file = {motion!r}
    edit(cfg)
  return reference_env.make_env("Mjlab-Tracking-Flat-Unitree-G1", num_envs=n, device=device, seed=7, cfg_edit=both)
print("RESULT " + json.dumps(_graphed_check.run_fused_vs_torch_tracking(make, "cuda:0", num_envs=128, steps=40)))
"""


def test_fused_motion_command_equals_its_torch_restatement(tmp_path):
  """MotionCommand's state write and relative body poses as HIP launches (mjlab_amd/env_terms.py) inside the captured tracking step
  against the torch restatements on the same uniforms: every environment, resets and ended motions included."""
  import json
  import subprocess

  code = _TRACKING_FUSED.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"), motion=str(tmp_path / "motion.npz"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print("fused MotionCommand vs torch restatement:", st)
  w = st["worst"]
  assert st["resets"] >= 32 and st["ended"] >= 32
  # (the two sides' states differ by an ulp after a resample -- the state write is 1 ulp from its torch restatement -- so the relative poses
  # are compared to a tolerance here; bit for bit against the eager reference on identical inputs: test_graphed_tracking_env_matches_the_reference_env)
  assert w["qpos"] <= 2e-6 and w["qvel"] <= 1e-5 and w["body_pos_relative_w"] <= 2e-6 and w["body_quat_relative_w"] <= 2e-6 and w["obs"] <= 1e-4 and w["reward"] <= 1e-5, w
  assert st["relative_rounding"] >= 8, st  # the launch was calibrated (a rounding that reproduces the reference's chain was found), not replaced by the torch chain


_GENERIC_EVENTS_GRAPHED = """
import json, sys
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env, _graphed_check
def make(n, device, edit):
  return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=n, device=device, seed=13, cfg_edit=edit)
st = _graphed_check.run_generic_events(make, "cuda:0", num_envs=128, steps=60, capture=True)
# a term that writes per-world MODEL fields at reset (domain randomisation in reset mode) has no masked form: refused at construction, by name
from mjlab_amd.graphed_env import GraphedRlEnv
def dr_at_reset(cfg):
  cfg.events.foot_friction.mode = "reset"
env = make(8, "cuda:0", dr_at_reset)
env.reset()
try:
  GraphedRlEnv(env, capture=False)
  st["refused"] = ""
except NotImplementedError as e:
  st["refused"] = str(e)
print("RESULT " + json.dumps(st))
"""


def test_graphed_env_with_stock_event_terms_run_generically():
  """The reference's ``reset_scene_to_default`` (reset) and ``apply_external_force_torque`` (interval) -- stock event terms without a
  restatement -- inside the captured control step: the reference's own functions on all environments, kept where the mask is set;
  teacher-forced against the eager reference on the MI355X (tests/_graphed_check.py::run_generic_events)."""
  import json
  import subprocess

  code = _GENERIC_EVENTS_GRAPHED.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print("graphed env with generically run event terms vs reference env:", st)
  assert st["graph"] and st["resets"] >= 100 and st["wrenches"] >= 400 and st["default_states_compared"] >= 80 and st["quiet_env_steps"] >= 1000
  assert "randomize_field" in st["refused"] and "model.geom_friction" in st["refused"], st["refused"]


_TOY_COMMAND_GRAPHED = """
import json, sys
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env, _graphed_check
def make(n, device, edit):
  return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=n, device=device, seed=17, cfg_edit=edit)
print("RESULT " + json.dumps(_graphed_check.run_toy_command(make, "cuda:0", num_envs=128, steps=60, capture=True)))
"""


def test_graphed_env_with_a_command_term_of_another_class():
  """A CommandTerm subclass without a restatement inside the captured control step (``_generic_command_resample``: the term's own
  ``_resample_command`` on all environments, kept where the mask is set), teacher-forced against the eager reference on the MI355X."""
  import json
  import subprocess

  code = _TOY_COMMAND_GRAPHED.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  st = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  print("graphed env with a generically run command term vs reference env:", st)
  assert st["graph"] and st["resets"] >= 100 and st["resamples"] >= 200 and st["quiet_env_steps"] >= 1000
