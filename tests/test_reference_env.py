"""Row N1 ("existing tasks run unmodified"): the reference's OWN ``ManagerBasedRlEnv`` -- Scene, Entity, every manager, the MDP
terms and the registered task configuration, imported unmodified from the reference checkout -- constructed and stepped over
this repository's boundary: ``mjlab_amd.mujoco_shim`` as ``mujoco`` (model building) and a ``Simulation``-shaped object.

Here, without a GPU, the Simulation is tests/_oracle_simulation.py (the CPU oracle behind the same Bridges; test
infrastructure).  tests/test_gpu_reference_env.py runs the same stack over ``mjlab_amd.sim.Simulation`` on the MI355X.
Reference call sites: envs/manager_based_rl_env.py:84-147, envs/manager_based_env.py:54-160, scene/scene.py:25-147,
entity/entity.py:120-214,325-423,588-652."""

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "tests"))

import reference_env  # noqa: E402

pytestmark = pytest.mark.skipif(reference_env.locate_reference() is None, reason="reference checkout not present")


@pytest.fixture(scope="module")
def g1_env():
  from _oracle_simulation import OracleSimulation

  return reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=16, device="cpu", sim_cls=OracleSimulation)


def test_scene_compiled_by_the_reference_equals_the_committed_model(g1_env):
  """The model the reference's Scene / Entity / spec_config code builds through the shim is, field for field, the model
  mjlab_amd/assets/g1_velocity_flat.npz holds (which the parity tests and bench.py run on) -- plus the reference's marker sites."""
  from mjlab_amd import robots

  m, ref = g1_env.sim.mj_model, robots.load_model("g1_velocity_flat")
  n = g1_env.num_envs
  assert (m.nq, m.nv, m.nu, m.nbody, m.ngeom, m.npair, m.nsensor) == (ref.nq, ref.nv, ref.nu, ref.nbody, ref.ngeom, ref.npair, ref.nsensor)
  assert m.nsite == ref.nsite + n and m.nstaticsite == n  # one env_origin site per environment on the world body, posed once
  for kind in ("body", "joint", "geom", "actuator", "sensor"):
    assert m.names[kind] == ref.names[kind], kind
  skip = {"site_bodyid", "site_pos", "site_quat"}
  for k, v in ref.__dict__.items():
    if not isinstance(v, np.ndarray) or k in skip:
      continue
    a = getattr(m, k)
    assert a.shape == v.shape, k
    if v.dtype.kind == "f":
      np.testing.assert_allclose(a, v, rtol=1e-12, atol=1e-14, err_msg=k)
    else:
      assert np.array_equal(a, v), k
  assert m.opt.__dict__ == ref.opt.__dict__
  assert np.array_equal(m.site_bodyid[n:], ref.site_bodyid) and np.allclose(m.site_pos[n:], ref.site_pos)


def test_entity_indexing_and_spec_views_after_attach(g1_env):
  """mujoco's attach-by-reference semantics the reference relies on: the entity's own spec keeps listing its elements, with
  prefixed names and the ids of the compiled SCENE (entity/entity.py:189-214,588-652)."""
  robot = g1_env.scene["robot"]
  m = g1_env.sim.mj_model
  assert robot.joint_names[0] == "left_hip_pitch_joint" and len(robot.joint_names) == 29 and robot.num_actuators == 29
  assert robot.spec.bodies[1].name == "robot/pelvis" and robot.indexing.root_body_id == m.names["body"].index("robot/pelvis")
  assert robot.indexing.body_ids.tolist() == list(range(2, 32))  # world, terrain, then the robot's 30 bodies
  assert robot.indexing.ctrl_ids.tolist() == list(range(29))
  assert robot.indexing.free_joint_q_adr.tolist() == list(range(7)) and robot.indexing.joint_q_adr.tolist() == list(range(7, 36))
  assert set(robot.indexing.sensor_adr) == {"left_foot_ground_contact", "right_foot_ground_contact"}
  # the foot sensors reference the SCENE's terrain body, which is outside the robot spec: the name keeps no prefix
  assert m.sensor_refid.tolist() == [1, 1] and m.names["body"][1] == "terrain"
  # startup domain randomisation went through expand_model_fields + randomize_field on the per-world friction table
  fr = g1_env.sim.model.geom_friction
  assert fr.shape[0] == g1_env.num_envs and fr.stride(0) > 0 and float(fr[:, :, 0].std()) > 0.0


def test_step_the_registered_task(g1_env):
  import torch

  env = g1_env
  ptr = {f: getattr(env.sim.data, f).data_ptr() for f in ("qpos", "qvel", "ctrl", "xfrc_applied", "sensordata")}
  calls0 = env.sim.step_calls
  seen_reset = []
  out = reference_env.random_rollout(env, 25, seed=3, on_step=lambda k, o, r, t, to: seen_reset.append(int((t | to).sum())))
  assert env.sim.step_calls - calls0 == 25 * env.cfg.decimation  # 4 physics steps per env step (manager_based_rl_env.py:109-114)
  for name, o in out["obs"].items():
    assert o.shape == (env.num_envs, 99) and bool(torch.isfinite(o).all()), name
  assert {f: getattr(env.sim.data, f).data_ptr() for f in ptr} == ptr  # in-place writes only (tests/test_sim_data.py:62-70)
  assert np.isfinite(out["mean_reward"])
  # force terminations: tip every robot over -> fell_over fires, the reset events write the keyframe state back
  env.sim.data.qpos[:, 3:7] = torch.tensor([0.0, 1.0, 0.0, 0.0])
  fwd0 = env.sim.forward_calls
  _, _, terminated, _, _ = env.step(torch.zeros(env.num_envs, 29))
  assert bool(terminated.all()) and env.sim.forward_calls == fwd0 + 1  # forward() after the reset (manager_based_rl_env.py:128-132)
  z = env.sim.data.qpos[:, 2]
  assert bool(((z > 0.7) & (z < 0.85)).all()) and bool((env.episode_length_buf == 0).all())


_GO1_SCRIPT = """
import json, sys
import numpy as np
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env
from _oracle_simulation import OracleSimulation
from mjlab_amd import robots
env = reference_env.make_env("Mjlab-Velocity-Flat-Unitree-Go1", num_envs=4, device="cpu", sim_cls=OracleSimulation)
ref, m = robots.load_model("go1_velocity_flat"), env.sim.mj_model
sizes = [(int(getattr(m, k)), int(getattr(ref, k))) for k in ("nq", "nv", "nu", "nbody", "ngeom", "npair", "nsensordata")]
worst = max(float(np.abs(np.asarray(getattr(m, k), float) - np.asarray(getattr(ref, k), float)).max()) for k in
            ("body_mass", "body_inertia", "dof_armature", "actuator_gainprm", "actuator_biasprm", "geom_friction", "geom_condim", "pair_geom", "dof_invweight0"))
out = reference_env.random_rollout(env, 5)
print("RESULT " + json.dumps({{"sizes": sizes, "worst": worst, "finite": all(bool(np.isfinite(o.numpy()).all()) for o in out["obs"].values())}}))
"""


def test_go1_task_constructs_and_steps():
  """A second registered task (config 2's robot).  In its own process: the reference's task configs share mutable default
  objects between robots (the velocity task's `foot_friction` SceneEntityCfg is edited in place by each robot's cfg), so two
  different tasks cannot be built in one interpreter -- with any engine."""
  import json
  import subprocess

  code = _GO1_SCRIPT.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-2000:]
  res = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  assert all(a == b for a, b in res["sizes"]), res["sizes"]
  assert res["worst"] < 1e-12 and res["finite"]


_ELLIPTIC_SCRIPT = """
import json, sys
import numpy as np
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env
from _oracle_simulation import OracleSimulation
def edit(cfg):
  cfg.sim.mujoco.cone = "elliptic"  # reference sim/sim.py:49,52: reaches the compiled model through MujocoCfg.edit_spec
env = reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=8, device="cpu", sim_cls=OracleSimulation, cfg_edit=edit)
out = reference_env.random_rollout(env, 6)
d = env.sim.data
nefc = d.nefc.numpy().ravel()
types = [int((d.efc_type.numpy()[w, : nefc[w]] == 7).sum()) for w in range(8)]
print("RESULT " + json.dumps({{"cone": int(env.sim.mj_model.opt.cone), "elliptic_rows": types, "nefc": nefc.tolist(),
                              "finite": all(bool(np.isfinite(o.numpy()).all()) for o in out["obs"].values()) and bool(np.isfinite(d.qpos.numpy()).all())}}))
"""


def test_velocity_task_with_elliptic_cones_constructs_and_steps():
  """``MujocoCfg(cone="elliptic")`` on a registered task (no task ships it; the option is the reference's: sim/sim.py:49,52): the
  reference's own configuration path -- ``edit_spec`` over this repository's ``mujoco`` shim, ``Scene.compile`` -- hands the
  Simulation a model with elliptic cones, and the environment steps on it (three rows per foot contact).  In its own process, like
  the other tasks (shared mutable defaults in the reference's configs)."""
  import json
  import subprocess

  code = _ELLIPTIC_SCRIPT.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-2000:]
  res = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  assert res["cone"] == 1 and res["finite"]
  assert sum(res["elliptic_rows"]) >= 3 * 8 and all(k % 3 == 0 for k in res["elliptic_rows"]), res


_TRACKING_SCRIPT = """
import json, sys
import numpy as np
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env
from _motion_fixture import write_full_motion
from _oracle_simulation import OracleSimulation
from mjlab_amd import robots
write_full_motion({motion!r})
def edit(cfg):
  cfg.commands.motion.motion_file = {motion!r}
env = reference_env.make_env("Mjlab-Tracking-Flat-Unitree-G1", num_envs=8, device="cpu", sim_cls=OracleSimulation, cfg_edit=edit)
ref, m = robots.load_model("g1_tracking_flat"), env.sim.mj_model
sizes = [(int(getattr(m, k)), int(getattr(ref, k))) for k in ("nq", "nv", "nu", "nbody", "ngeom", "npair", "nsensor", "nsensordata")]
worst = max(float(np.abs(np.asarray(getattr(m, k), float) - np.asarray(getattr(ref, k), float)).max()) for k in
            ("body_mass", "body_inertia", "dof_armature", "actuator_gainprm", "actuator_biasprm", "geom_friction", "geom_condim", "pair_geom", "dof_invweight0", "sensor_intprm"))
cmd = env.command_manager.get_term("motion")
env.reset()
# what the REFERENCE's reset wrote (MotionCommand._resample_command): the invariants tests/test_gpu_fullsize.py asserts for
# this repository's in-kernel restatement of it (mjlab_motion_reset_t)
t = cmd.time_steps
q0 = env.sim.data.qpos.clone()
root = env.scene["robot"].indexing.root_body_id
ref_z = cmd.motion.body_pos_w[t, 0, 2] + env.scene.env_origins[:, 2]
anchor_joint_dev = float((q0[:, 7:] - cmd.motion.joint_pos[t]).abs().max())
root_xy_dev = float((q0[:, :2] - (cmd.motion._body_pos_w[t, 0, :2] + env.scene.env_origins[:, :2])).abs().max())
root_z_dev = float((q0[:, 2] - (cmd.motion._body_pos_w[t, 0, 2] + env.scene.env_origins[:, 2])).abs().max())
qvel_root = float(env.sim.data.qvel[:, :3].abs().max())
out = reference_env.random_rollout(env, 12)
print("RESULT " + json.dumps({{"sizes": sizes, "worst": worst, "finite": all(bool(np.isfinite(o.numpy()).all()) for o in out["obs"].values()),
      "obs": {{k: list(v.shape) for k, v in out["obs"].items()}}, "resets": out["resets"], "frames": int(cmd.motion.time_step_total),
      "phase_max": int(cmd.time_steps.max()), "dr": [float(env.sim.model.body_ipos.std(dim=0).max()), float(env.sim.model.qpos0.std(dim=0).max())],
      "reset": [anchor_joint_dev, root_xy_dev, root_z_dev, qvel_root]}}))
"""


def test_tracking_task_constructs_and_steps(tmp_path):
  """The SECOND task north_star names, Mjlab-Tracking-Flat-Unitree-G1 (BASELINE config 4): the reference's MotionCommand,
  its body-tracking rewards, anchor / end-effector terminations, the self-collision contact sensor and the startup
  randomisation of torso com and joint zero offsets, unmodified, over the boundary -- fed with a synthetic motion file
  (tests/_motion_fixture.py; none is in the reference tree).  Own process: see test_go1_task_constructs_and_steps."""
  import json
  import subprocess

  code = _TRACKING_SCRIPT.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"), motion=str(tmp_path / "motion.npz"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-2000:]
  res = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  assert all(a == b for a, b in res["sizes"]), res["sizes"]
  assert res["worst"] < 1e-12 and res["finite"]
  assert res["obs"] == {"policy": [8, 160], "critic": [8, 286]} and res["frames"] == 500 and res["phase_max"] < 500
  assert res["resets"] > 0  # random actions leave the motion: the anchor terminations fire, MotionCommand resamples a phase
  assert res["dr"][0] > 0.0 and res["dr"][1] > 0.0  # per-world body_ipos and qpos0 went through expand_model_fields
  # the reference's own reset, read back: joints within the cfg's +-0.1 of the sampled motion frame, the root within the pose
  # noise (x, y +-0.05, z +-0.01) of the motion's root, a velocity kick in qvel -- the same invariants the GPU test asserts for
  # the in-kernel restatement (mjlab_motion_reset_t)
  jdev, xy, z, v = res["reset"]
  assert jdev <= 0.1 + 1e-5 and xy <= 0.05 + 1e-5 and z <= 0.01 + 1e-5 and 0.05 < v <= 0.5 + 1e-5, res["reset"]


_RESET_ROWS = """
import json, sys
import numpy as np
import torch
sys.path.insert(0, {tools!r}); sys.path.insert(0, {tests!r})
import reference_env
from _motion_fixture import write_full_motion
from _oracle_simulation import OracleSimulation
from mjlab_amd import robots
from mjlab_amd.rollout import TRACKING_TASK_EVENTS, motion_reset_rows, motion_reset_tables, synthetic_motion
write_full_motion({motion!r})
def edit(cfg):
  cfg.commands.motion.motion_file = {motion!r}
n = 64
env = reference_env.make_env("Mjlab-Tracking-Flat-Unitree-G1", num_envs=n, device="cpu", sim_cls=OracleSimulation, cfg_edit=edit)
env.reset()
cmd = env.command_manager.get_term("motion")
model = robots.load_model("g1_tracking_flat")
mo = motion_reset_tables(model, synthetic_motion(model), TRACKING_TASK_EVENTS["g1"]["motion_reset"], 4, n, "cpu")
assert mo["bins"] == cmd.bin_count and mo["nframe"] == cmd.motion.time_step_total
gen = torch.Generator().manual_seed(5)
u = torch.rand((n, 14 + model.nq - 7), generator=gen)
mo["rnd"].copy_(u)
# ---- the reference's own _resample_command, its random draws replaced by the SAME uniforms, in its call order:
#   _adaptive_sampling: torch.multinomial(p, n) -> the bin (uniform bins while nothing has failed), sample_uniform(0, 1, (n,)) -> the phase in the bin
#   _resample_command:  sample_uniform(pose ranges, (n, 6)), sample_uniform(velocity ranges, (n, 6)), sample_uniform(joint range, (n, nj))
import mjlab.tasks.tracking.mdp.commands as C
queue = [u[:, 1], u[:, 2:8], u[:, 8:14], u[:, 14:]]
def fed(lower, upper, size, device=None):
  x = queue.pop(0)
  assert tuple(x.shape) == tuple(size if not isinstance(size, int) else (size,)), (x.shape, size)
  return lower + (upper - lower) * x
C.sample_uniform = fed
real_multinomial = torch.multinomial
torch.multinomial = lambda p, k, replacement=True: torch.clamp((u[:, 0] * cmd.bin_count).long(), max=cmd.bin_count - 1)
try:
  cmd._resample_command(torch.arange(n))
finally:
  torch.multinomial = real_multinomial
assert not queue
t_new = motion_reset_rows(mo, env.scene.env_origins)
d = env.sim.data
dq = (d.qpos - mo["reset_qpos"]).abs()
dv = (d.qvel - mo["reset_qvel"]).abs()
print("RESULT " + json.dumps({{"t_equal": bool(torch.equal(cmd.time_steps, t_new)), "qpos": float(dq.max()), "qvel": float(dv.max()),
      "root_quat": float(dq[:, 3:7].max()), "clipped": int((mo["reset_qpos"][:, 7:] == mo["soft_lo"]).sum() + (mo["reset_qpos"][:, 7:] == mo["soft_hi"]).sum()),
      "t_spread": int(t_new.max() - t_new.min())}}))
"""


def test_tracking_reset_rows_equal_the_reference_resample_command_on_the_same_uniforms(tmp_path):
  """``mjlab_amd.rollout.motion_reset_rows`` -- the rows the control kernel's in-kernel tracking reset reproduces bit
  for bit (tests/test_gpu_fullsize.py) -- against the reference's OWN ``MotionCommand._resample_command`` (tasks/tracking/mdp/
  commands.py:255-363) with its random draws replaced by the same uniforms: sampled phases equal, every qpos / qvel entry to float
  rounding.  Own process (the tracking task's configs; torch.multinomial is patched)."""
  import json
  import subprocess

  code = _RESET_ROWS.format(tools=str(ROOT / "tools"), tests=str(ROOT / "tests"), motion=str(tmp_path / "motion.npz"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
  assert r.returncode == 0, r.stderr[-3000:]
  res = json.loads(next(line for line in r.stdout.splitlines() if line.startswith("RESULT "))[7:])
  assert res["t_equal"] and res["t_spread"] > 100
  assert res["qpos"] <= 2e-6 and res["qvel"] <= 2e-6, res  # float32 rounding of the same arithmetic in a different association
