"""Record the inputs and the EAGER REFERENCE'S OWN results of the pieces of a control step that ``GraphedRlEnv`` restates
(mjlab_amd/env_core.py), so that the GPU box -- which has no reference tree -- can replay the restatements against them
(a reference-free test surface for row f3).

Runs where the reference checkout exists.  Per task (fresh interpreter each), the reference's unmodified ``ManagerBasedRlEnv`` over
tests/_oracle_simulation.py, 16 environments, observation noise off, random actions with a few flailing robots; observed:

  reset bookkeeping   ``_reset_idx`` (reference envs/manager_based_rl_env.py:214-249): the buffers the managers' reset() fill and the
                      vectors they sum (enumerated by mjlab_amd.graphed_env.bookkeeping_plan on the SAME environment object)
                      before and after the reference's own call, the environment ids, and the ``extras["log"]`` it leaves;
  reward accumulation ``RewardManager.compute`` (managers/reward_manager.py:77-89): every active term's raw output, the weights, and
                      reward_buf / _step_reward / _episode_sums after the reference's own loop;
  observations        ``ObservationManager.compute`` (managers/observation_manager.py:144-188): every term's raw output per group
                      and the assembled groups;
  velocity command    ``UniformVelocityCommand._update_command`` (tasks/velocity/mdp/velocity_command.py:88-102): its inputs and the
                      command it leaves.

-> tests/golden/graphed_core_<scene>.npz + .json; replayed by tests/test_graphed_core_golden.py (CPU, and -m gpu through the torch
twins AND the HIP launches of mjlab_amd/env_terms.py).

  python tools/make_graphed_golden.py            (all tasks, one subprocess each)
  python tools/make_graphed_golden.py <scene>    (one task, in this interpreter)
"""

from __future__ import annotations

import json
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tools", ROOT / "tests"):
  sys.path.insert(0, str(p))

TASKS = {"g1_velocity_flat": "Mjlab-Velocity-Flat-Unitree-G1", "g1_tracking_flat": "Mjlab-Tracking-Flat-Unitree-G1",
         # the two registered tasks no GraphedRlEnv test had run (reference tasks/tracking/config/g1/__init__.py:24,
         # tasks/velocity/config/go1/__init__.py:4)
         "g1_tracking_flat_nse": "Mjlab-Tracking-Flat-Unitree-G1-No-State-Estimation", "go1_velocity_rough": "Mjlab-Velocity-Rough-Unitree-Go1"}
NUM_ENVS, SEED = 16, 21
NUM_STEPS = {"g1_velocity_flat": 90, "g1_tracking_flat": 24, "g1_tracking_flat_nse": 24, "go1_velocity_rough": 90}
MAX_EVENTS = {"reset": 10, "reward": 12, "obs": 6, "velocity": 8}  # recorded calls per kind (the first ones with content)


def record(scene: str) -> tuple[dict, dict]:
  import reference_env
  import torch
  from _oracle_simulation import OracleSimulation

  from mjlab_amd.graphed_env import _as_slice, bookkeeping_plan

  tracking = scene.startswith("g1_tracking_flat")
  motion = None
  if tracking:
    from _motion_fixture import write_full_motion

    motion = str(Path(tempfile.mkdtemp()) / "motion.npz")
    write_full_motion(motion)

  def edit(cfg):
    for group in ("policy", "critic"):
      getattr(cfg.observations, group).enable_corruption = False
    if tracking:
      cfg.commands.motion.motion_file = motion
    else:
      cfg.episode_length_s = 1.2  # time-outs inside the run, next to the falls
      cfg.events.push_robot.interval_range_s = (0.2, 0.6)

  torch.manual_seed(SEED)
  env = reference_env.make_env(TASKS[scene], num_envs=NUM_ENVS, device="cpu", sim_cls=OracleSimulation, seed=SEED, cfg_edit=edit)
  env.reset()
  n = NUM_ENVS
  robot = env.scene["robot"]
  ix = robot.indexing
  # the index ranges GraphedRlEnv._index_slices derives (qfrc_applied over the free joint's dofs, xfrc_applied over the robot's bodies, ctrl over its actuators)
  slices = (_as_slice(ix.free_joint_v_adr), _as_slice(ix.body_ids), _as_slice(ix.ctrl_ids), None)
  n_reset_terms = len(env.event_manager._mode_term_cfgs.get("reset", []))
  arrays: dict = {}
  meta: dict = {"scene": scene, "task": TASKS[scene], "num_envs": n, "dt": float(env.step_dt), "max_episode_length_s": float(env.max_episode_length_s),
                "reset": [], "reward": [], "obs": [], "velocity": []}

  def put(key: str, t) -> str:
    arrays[key] = t.detach().cpu().numpy().copy() if hasattr(t, "detach") else np.asarray(t)
    return key

  # ---- reset bookkeeping
  orig_reset = env._reset_idx

  def reset_idx(env_ids):
    fills, vectors, rkeys, mkeys, tkeys, whole_clear = bookkeeping_plan(env, robot, slices, n_reset_terms, env.episode_length_buf)
    k = len(meta["reset"])
    rec = len(env_ids) > 0 and k < MAX_EVENTS["reset"]
    if rec:
      tag = f"reset{k}"
      mask = torch.zeros(n, dtype=torch.bool)
      mask[env_ids] = True
      entry = {"mask": put(f"{tag}.mask", mask), "rkeys": rkeys, "mkeys": mkeys, "tkeys": tkeys, "whole_clear": whole_clear,
               "fills": [{"name": nm, "value": (int(v) if not isinstance(v, float) else v), "pre": put(f"{tag}.fill_pre.{i}", t)} for i, (nm, t, v) in enumerate(fills)],
               "vectors": [{"name": nm, "pre": put(f"{tag}.vec.{i}", t)} for i, (nm, t) in enumerate(vectors)]}
    orig_reset(env_ids)
    if rec:
      for i, (nm, t, v) in enumerate(fills):
        entry["fills"][i]["post"] = put(f"{tag}.fill_post.{i}", t)
      entry["log"] = {key: float(val) for key, val in env.extras["log"].items() if isinstance(val, (int, float)) or (hasattr(val, "numel") and val.numel() == 1)}
      meta["reset"].append(entry)

  env._reset_idx = reset_idx

  # ---- reward accumulation: raw term outputs through recording wrappers
  rm = env.reward_manager
  raw_reward: dict = {}
  for name, cfg in zip(rm._term_names, rm._term_cfgs, strict=True):
    def wrapped(e, _f=cfg.func, _n=name, **params):
      out = _f(e, **params)
      raw_reward[_n] = out.clone()
      return out

    if hasattr(cfg.func, "reset"):  # a class-based term: the manager calls its reset() (managers/reward_manager.py:72-74)
      wrapped.reset = cfg.func.reset
    cfg.func = wrapped
  orig_compute = rm.compute

  def compute(dt):
    pre = {k: v.clone() for k, v in rm._episode_sums.items()}
    raw_reward.clear()
    out = orig_compute(dt)
    k = len(meta["reward"])
    if k < MAX_EVENTS["reward"]:
      tag = f"reward{k}"
      active = [(i, nm, cfg) for i, (nm, cfg) in enumerate(zip(rm._term_names, rm._term_cfgs, strict=True)) if cfg.weight != 0.0]
      meta["reward"].append({"dt": float(dt), "names": [nm for _, nm, _ in active], "columns": [i for i, _, _ in active], "weights": [float(c.weight) for _, _, c in active],
                             "nterms": len(rm._term_names),
                             "values": put(f"{tag}.values", torch.stack([raw_reward[nm] for _, nm, _ in active])),
                             "sums_pre": put(f"{tag}.sums_pre", torch.stack([pre[nm] for _, nm, _ in active])),
                             "sums_post": put(f"{tag}.sums_post", torch.stack([rm._episode_sums[nm] for _, nm, _ in active])),
                             "reward_buf": put(f"{tag}.reward_buf", out), "step_reward": put(f"{tag}.step_reward", rm._step_reward)})
    return out

  rm.compute = compute

  # ---- observations
  om = env.observation_manager
  raw_obs: dict = {}
  for group, cfgs in om._group_obs_term_cfgs.items():
    for name, cfg in zip(om._group_obs_term_names[group], cfgs, strict=True):
      def wrapped(e, _f=cfg.func, _k=(group, name), **params):
        out = _f(e, **params)
        raw_obs[_k] = out.clone()
        return out

      cfg.func = wrapped
  orig_obs = om.compute

  def obs_compute(*a, **kw):
    raw_obs.clear()
    out = orig_obs(*a, **kw)
    k = len(meta["obs"])
    if k < MAX_EVENTS["obs"] and env.common_step_counter % 7 == 3:
      tag = f"obs{k}"
      meta["obs"].append({"groups": {g: {"terms": [put(f"{tag}.{g}.{nm}", raw_obs[(g, nm)]) for nm in om._group_obs_term_names[g]], "out": put(f"{tag}.{g}.out", out[g])}
                                     for g in om._group_obs_term_names}})
    return out

  om.compute = obs_compute

  # ---- velocity command update
  if not tracking:
    term = env.command_manager.get_term("twist")
    orig_update = term._update_command

    def update_command():
      k = len(meta["velocity"])
      rec = k < MAX_EVENTS["velocity"] and env.common_step_counter % 9 == 4
      if rec:
        tag = f"vel{k}"
        cfg = term.cfg
        entry = {"heading_command": bool(cfg.heading_command), "stiffness": float(cfg.heading_control_stiffness), "ang_vel_z": [float(x) for x in cfg.ranges.ang_vel_z],
                 "pre": put(f"{tag}.pre", term.vel_command_b), "heading_target": put(f"{tag}.heading_target", term.heading_target),
                 "heading_w": put(f"{tag}.heading_w", term.robot.data.heading_w), "is_heading_env": put(f"{tag}.is_heading_env", term.is_heading_env),
                 "is_standing_env": put(f"{tag}.is_standing_env", term.is_standing_env)}
      orig_update()
      if rec:
        entry["post"] = put(f"{tag}.post", term.vel_command_b)
        meta["velocity"].append(entry)

    term._update_command = update_command

  gen = torch.Generator().manual_seed(SEED + 1)
  na = sum(env.action_manager.action_term_dim)
  for k in range(NUM_STEPS[scene]):
    action = torch.rand((n, na), generator=gen) * 2 - 1
    if k > 10:
      action[: n // 4] *= 6.0  # a few robots flail and fall: terminations besides the time-outs
    env.step(action)
  assert len(meta["reset"]) >= 3 and len(meta["reward"]) >= 6 and len(meta["obs"]) >= 2, {k: len(v) for k, v in meta.items() if isinstance(v, list)}
  return arrays, meta


if __name__ == "__main__":
  out_dir = ROOT / "tests" / "golden"
  if len(sys.argv) > 1:
    arrays, meta = record(sys.argv[1])
    np.savez_compressed(out_dir / f"graphed_core_{sys.argv[1]}.npz", **arrays)
    (out_dir / f"graphed_core_{sys.argv[1]}.json").write_text(json.dumps(meta, indent=1))
    print(sys.argv[1], {k: len(v) for k, v in meta.items() if isinstance(v, list)}, f"{sum(a.nbytes for a in arrays.values()) / 1e3:.0f} kB")
  else:
    for scene in TASKS:
      subprocess.run([sys.executable, __file__, scene], check=True)
