"""Record what the REFERENCE'S OWN environment computes, step by step, so that the GPU box -- which has no reference tree --
can check the HIP path against it (VERDICT round 3 "do this" item 3: driver-verifiable row N1).

Runs where the reference checkout exists (the build container).  For each of the two tasks ``north_star`` names,

  Mjlab-Velocity-Flat-Unitree-G1      (reference tasks/velocity/config/g1/__init__.py:23-31)
  Mjlab-Tracking-Flat-Unitree-G1      (reference tasks/tracking/config/g1/__init__.py)

the reference's ``ManagerBasedRlEnv`` -- Scene, Entity, every manager, the MDP terms and the registered task configuration,
imported unmodified (tools/reference_env.py) -- is built over tests/_oracle_simulation.py (the fp32 CPU oracle behind the same
``Simulation`` surface; ``ls_parallel`` as the reference's SimulationCfg says: on) with 16 envs, observation noise off, and stepped
80 / 20 times with recorded random actions (reference scripts/play.py:159-172).  ``env.step`` (reference
envs/manager_based_rl_env.py:106-147) is observed at the points where it hands state between the physics and the managers:

  pre_*      qpos / qvel / qacc_warmstart when step() is entered                 (what the 4 substeps start from)
  action     the policy action; ctrl = what the action manager wrote into sim.data.ctrl   (:107, :111)
  post_*     qpos / qvel after the 4 x [apply_action, sim.step()]                 (:109-114: the physics' own result)
  terminated / time_out / reward, episode_length                                    (:121-126, computed from the post state)
  reset_*    qpos / qvel after _reset_idx + sim.forward() (only when an env reset; :129-133), reset_ids
  final_*    qpos / qvel after command update + interval events (pushes)           (:135-138)
  obs_<group>  the observation groups (:140), term names and widths in the json side-car
  per task   commands (twist | motion anchor / time_steps), the per-world model fields after the startup randomisation,
             env origins, default joint state, the soft joint limits -- everything a consumer needs to rebuild the observation
             terms from mjData WITHOUT the reference's code.

-> tests/golden/env_<task>.npz + .json.  tests/test_env_golden.py replays them: on the CPU over the oracle (bit for bit: the
recorder is deterministic), on the MI355X over ``mjlab_amd.Simulation`` + ``EntityReadback`` (-m gpu, no reference tree needed).
"""

from __future__ import annotations

import json
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "tests"))

TASKS = {"g1_velocity_flat": "Mjlab-Velocity-Flat-Unitree-G1", "g1_tracking_flat": "Mjlab-Tracking-Flat-Unitree-G1",
         # beyond north_star's two: the rough variant -- the terrain the reference's own generator builds (10 x 20 sub-terrains of box
         # stairs, 3632 geoms) under the seed mjlab_amd.robots uses for its scene of the same name, and the terrain curriculum
         "g1_velocity_rough": "Mjlab-Velocity-Rough-Unitree-G1"}
NUM_ENVS, SEED = 16, 42
# control steps: the tracking task resets within the first step (below); the velocity task needs ~50 steps of random actions before
# robots fall (fell_over) and the first interval pushes (every U(1, 3) s = 50..150 steps) fire
NUM_STEPS = {"g1_velocity_flat": 80, "g1_tracking_flat": 20, "g1_velocity_rough": 80}
DR_FIELDS = ("geom_friction", "body_ipos", "qpos0")


def record(scene: str, num_envs: int = NUM_ENVS, num_steps: int | None = None, seed: int = SEED) -> tuple[dict, dict]:
  """-> (arrays, meta) of one task.  Must run in a fresh interpreter per task (the reference's task configs share mutable
  defaults between tasks: tests/test_reference_env.py::test_go1_task_constructs_and_steps)."""
  import reference_env
  import torch
  from _oracle_simulation import OracleSimulation

  task = TASKS[scene]
  num_steps = num_steps or NUM_STEPS[scene]
  tmp = tempfile.mkdtemp()

  def edit(cfg):
    for group in ("policy", "critic"):
      getattr(cfg.observations, group).enable_corruption = False  # observation noise off: the terms themselves are compared
    if scene == "g1_velocity_rough":
      from mjlab_amd import robots

      cfg.scene.terrain.terrain_generator.seed = robots.ROUGH_TERRAIN_SEED
    if scene == "g1_tracking_flat":
      from _motion_fixture import write_full_motion

      write_full_motion(str(Path(tmp) / "motion.npz"))
      cfg.commands.motion.motion_file = str(Path(tmp) / "motion.npz")

  env = reference_env.make_env(task, num_envs=num_envs, device="cpu", sim_cls=OracleSimulation, seed=seed, cfg_edit=edit)
  sim, robot = env.sim, env.scene["robot"]
  d = sim.data
  if scene == "g1_velocity_rough":  # the consumer rebuilds the model with mjlab_amd.robots.load_model(scene): it must be this one
    from mjlab_amd import robots

    own = robots.load_model(scene)
    for f in ("geom_pos", "geom_size", "geom_quat", "geom_type", "geom_bodyid", "geom_friction", "geom_priority", "body_pos", "body_mass"):
      assert np.array_equal(np.asarray(getattr(env.sim.mj_model, f)), np.asarray(getattr(own, f))), f
  snap = lambda *names: {n: getattr(d, n).detach().clone().numpy() for n in names}  # noqa: E731
  steps: list[dict] = []
  cur: dict = {}

  # -- observation points inside env.step(): wrap the two manager calls that sit right after the physics / right after the reset
  term_compute, cmd_compute = env.termination_manager.compute, env.command_manager.compute

  def termination_compute():
    cur.update({"post_" + k: v for k, v in snap("qpos", "qvel", "qacc").items()})
    cur["ctrl"] = d.ctrl.detach().clone().numpy()
    cur["episode_length"] = env.episode_length_buf.clone().numpy()
    return term_compute()

  def command_compute(dt):
    cur.update({"reset_" + k: v for k, v in snap("qpos", "qvel").items()})
    return cmd_compute(dt=dt)

  env.termination_manager.compute = termination_compute
  env.command_manager.compute = command_compute

  gen = torch.Generator(device="cpu")
  gen.manual_seed(seed + 1)
  na = sum(env.action_manager.action_term_dim)
  obs, _ = env.reset()
  obs0 = {g: o.clone().numpy() for g, o in obs.items()}
  for _ in range(num_steps):
    cur = {"pre_" + k: v for k, v in snap("qpos", "qvel", "qacc_warmstart").items()}
    action = 2.0 * torch.rand((num_envs, na), generator=gen) - 1.0
    cur["action"] = action.numpy().copy()
    fwd0 = sim.forward_calls
    obs, rew, terminated, time_out, _ = env.step(action)
    cur.update({"final_" + k: v for k, v in snap("qpos", "qvel").items()})
    cur.update(reward=rew.clone().numpy(), terminated=terminated.clone().numpy(), time_out=time_out.clone().numpy(),
               forward_ran=np.array(sim.forward_calls - fwd0))
    for g, o in obs.items():
      cur["obs_" + g] = o.clone().numpy()
    if scene.startswith("g1_velocity"):
      cur["command"] = env.command_manager.get_command("twist").clone().numpy()
      if scene == "g1_velocity_rough":
        cur["terrain_levels"] = env.scene.terrain.terrain_levels.clone().numpy()
        cur["env_origins_step"] = env.scene.env_origins.clone().numpy()
    else:
      cmd = env.command_manager.get_term("motion")
      cur.update(command=cmd.command.clone().numpy(), time_steps=cmd.time_steps.clone().numpy(), anchor_pos_w=cmd.anchor_pos_w.clone().numpy(),
                 anchor_quat_w=cmd.anchor_quat_w.clone().numpy())
    steps.append(cur)
  arrays = {k: np.stack([s[k] for s in steps]) for k in steps[0]}
  arrays.update({"obs0_" + g: o for g, o in obs0.items()})
  # -- what a consumer needs besides the trajectory
  for f in DR_FIELDS:
    t = getattr(sim.model, f)
    if t.stride(0) != 0:  # per-world after the startup events (expand_model_fields + randomize_field)
      arrays["dr_" + f] = t.detach().clone().numpy()
  arrays["env_origins"] = env.scene.env_origins.clone().numpy()
  arrays["default_joint_pos"] = robot.data.default_joint_pos.clone().numpy()
  arrays["default_joint_vel"] = robot.data.default_joint_vel.clone().numpy()
  term = env.action_manager.get_term("joint_pos")
  arrays["action_scale"] = np.broadcast_to(np.asarray(term._scale.numpy() if hasattr(term._scale, "numpy") else term._scale, np.float32), (num_envs, na)).copy()
  arrays["action_offset"] = np.broadcast_to(np.asarray(term._offset.numpy() if hasattr(term._offset, "numpy") else term._offset, np.float32), (num_envs, na)).copy()
  meta = {
    "task": task, "scene": scene, "num_envs": num_envs, "num_steps": num_steps, "seed": seed, "decimation": int(env.cfg.decimation),
    "physics_dt": float(env.physics_dt), "max_episode_length": int(env.max_episode_length),
    "njmax": int(env.cfg.sim.njmax), "ls_parallel": bool(env.cfg.sim.ls_parallel),
    "obs_terms": {g: list(zip(env.observation_manager.active_terms[g], [int(np.prod(s)) for s in env.observation_manager.group_obs_term_dim[g]]))
                  for g in env.observation_manager.active_terms},
    "termination_terms": list(env.termination_manager.active_terms),
    "reward_terms": list(env.reward_manager.active_terms),
    "resets": int(arrays["terminated"].sum() + arrays["time_out"].sum()),
  }
  if scene == "g1_tracking_flat":
    cmd = env.command_manager.get_term("motion")
    body_names = list(robot.body_names)
    meta["anchor_body"] = body_names.index(cmd.cfg.anchor_body_name)
    meta["tracked_bodies"] = [body_names.index(n) for n in cmd.cfg.body_names]
  return arrays, meta


def main() -> None:
  import subprocess

  if len(sys.argv) > 2 and sys.argv[1] == "--one":
    scene, out = sys.argv[2], Path(sys.argv[3])
    arrays, meta = record(scene)
    np.savez_compressed(out / f"env_{scene}.npz", **arrays)
    (out / f"env_{scene}.json").write_text(json.dumps(meta, indent=1) + "\n")
    print(scene, {k: v.shape for k, v in arrays.items() if k.startswith(("obs_", "pre_qpos"))}, "resets", meta["resets"])
    return
  out = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "tests" / "golden"
  out.mkdir(parents=True, exist_ok=True)
  for scene in TASKS:  # one interpreter per task
    subprocess.run([sys.executable, __file__, "--one", scene, str(out)], check=True)


if __name__ == "__main__":
  main()
