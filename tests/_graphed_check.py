"""TEST INFRASTRUCTURE shared by the CPU and the GPU test of ``mjlab_amd.graphed_env.GraphedRlEnv``: the graph-captured control
step against the reference's own ``ManagerBasedRlEnv.step``, teacher-forced.

Two environments of the same registered task are built with the same seed: A is stepped by the reference's ``env.step`` (eager,
index lists, host syncs), B by ``GraphedRlEnv.step`` (mask based; one hipGraph on the GPU, the same Python body uncaptured on the
CPU).  Before every step the COMPLETE state of A -- every mjData array, every manager / term buffer, the counters -- is copied
into B, both get the same action, and the outputs are compared:

  * ``terminated``, ``time_outs`` and ``reward`` are computed before any random number is drawn: equal bit for bit in EVERY env;
  * environments that did not reset, resample their command or get pushed in this step drew no random numbers: their
    observations, state and commands are equal bit for bit;
  * the others match in distribution, checked against the event / command configuration (reset pose inside the cfg's ranges
    around the default root state, default joints, zero velocity, zeroed action history and episode length; commands inside
    their ranges; pushes inside the velocity range).
"""

from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))

from mjlab_amd.graphed_env import GraphedRlEnv, _state_tensors  # noqa: E402


def _edit(cfg):
  for group in ("policy", "critic"):
    getattr(cfg.observations, group).enable_corruption = False  # (noise draws differ between the two by construction)
  cfg.episode_length_s = 0.6  # 30 control steps: time-outs in every run
  cfg.events.push_robot.interval_range_s = (0.1, 0.4)  # pushes every 5..20 steps
  cfg.commands.twist.resampling_time_range = (0.2, 0.5)  # command resampling inside an episode, not only at resets


def _managers(env):
  return (env.action_manager, env.reward_manager, env.termination_manager, env.command_manager, env.observation_manager, env.event_manager)


def _sync(a, b) -> None:
  """Everything env.step reads: mjData, manager / term state, counters."""
  for k, t in a.sim._data.items():
    b.sim._data[k].copy_(t)
  n = a.num_envs
  sa, sb = [], []
  for ma, mb in zip(_managers(a), _managers(b), strict=True):
    any_shape = ma is a.command_manager  # command terms also carry global state (the tracking task's sampler)
    _state_tensors(ma, None if any_shape else n, set(), sa)
    _state_tensors(mb, None if any_shape else n, set(), sb)
  assert [p for *_, p in sa] == [p for *_, p in sb], "the two environments do not hold the same state tensors"
  for (_, _, ta, _), (_, _, tb, _) in zip(sa, sb, strict=True):
    tb.copy_(ta)
  b.episode_length_buf.copy_(a.episode_length_buf)
  b._sim_step_counter, b.common_step_counter = a._sim_step_counter, a.common_step_counter
  ta, tb = getattr(a.scene, "terrain", None), getattr(b.scene, "terrain", None)
  if ta is not None and getattr(ta, "terrain_origins", None) is not None:  # the terrain curriculum's state
    tb.terrain_levels.copy_(ta.terrain_levels)
    tb.env_origins.copy_(ta.env_origins)


def _same_logs(a, b, k, reset) -> None:
  """``extras["log"]`` of a step in which some environment reset (reference envs/manager_based_rl_env.py:214-249: the managers' reset()
  logging -- episode reward sums, command metrics, termination counts, curriculum state -- as Python floats; GraphedRlEnv: 0-dim device
  tensors): the same keys, the same numbers to float rounding."""
  if not bool(reset.any()):
    return
  la, lb = a.extras["log"], b.extras["log"]
  missing = [key for key in la if key not in lb]
  assert not missing, (k, missing)
  for key, va in la.items():
    va = float(va.item() if isinstance(va, torch.Tensor) else va)
    vb = float(lb[key].float().mean().item() if isinstance(lb[key], torch.Tensor) else lb[key])
    if key == "Curriculum/terrain_levels":  # (an environment leaving the hardest row draws a random one: each side logs the mean of ITS levels)
      for env, v in ((a, va), (b, vb)):
        assert abs(v - float(env.scene.terrain.terrain_levels.float().mean())) <= 1e-6, (k, key, v)
      continue
    assert abs(va - vb) <= 1e-5 * (1.0 + abs(va)), (k, key, va, vb)


def _same_reward_state(a, b, k) -> None:
  """RewardManager's per-term bookkeeping (managers/reward_manager.py:84-88): episode sums and the per-step term values, bit for bit."""
  ra, rb = a.reward_manager, b.reward_manager
  assert torch.equal(ra._step_reward, rb._step_reward), (k, (ra._step_reward - rb._step_reward).abs().max())
  for name in ra._episode_sums:
    assert torch.equal(ra._episode_sums[name], rb._episode_sums[name]), (k, name)


def _same_event_bookkeeping(a, b, k) -> None:
  """EventManager's per-environment stamps of reset-mode terms (managers/event_manager.py:139-148): the env-step count of the reset and
  the triggered-once flag -- the step count is a value of the STEP, not of the construction / capture."""
  ea, eb = a.event_manager, b.event_manager
  for ta, tb in zip(ea._reset_term_last_triggered_step_id, eb._reset_term_last_triggered_step_id, strict=True):
    assert torch.equal(ta, tb), (k, ta, tb)
  for ta, tb in zip(ea._reset_term_last_triggered_once, eb._reset_term_last_triggered_once, strict=True):
    assert torch.equal(ta, tb), k


def _state_sweep(a, b, quiet, k, skip=()) -> int:
  """EVERY per-environment tensor the managers of the two environments hold (the same enumeration _sync copies from) and every
  per-world mjData array, in the environments that drew no random number in this step: bit for bit.  Returns the number compared."""
  n = a.num_envs
  sa, sb = [], []
  for ma, mb in zip(_managers(a), _managers(b), strict=True):
    _state_tensors(ma, n, set(), sa)
    _state_tensors(mb, n, set(), sb)
  assert [p for *_, p in sa] == [p for *_, p in sb], k
  bad = []
  for (_, _, ta, path), (_, _, tb, _) in zip(sa, sb, strict=True):
    if any(x in path for x in skip):
      continue
    if (".metrics.error_" in path or ".metrics.sampling_" in path) and ta.device.type == "cuda":
      # MotionCommand's logging metrics come from ONE launch each on the GPU (mjlab_command_motion_metrics; the sampler's entropy and
      # top-bin probability from mjlab_command_motion_sampler): a few ulp from the reference's mean / norm / log reductions,
      # documented in GraphedRlEnv's options; nothing but extras["log"] reads them
      if not torch.allclose(ta[quiet], tb[quiet], rtol=2e-5, atol=1e-6):
        bad.append(path)
    elif not torch.equal(ta[quiet], tb[quiet]):
      bad.append(path)
  private = ("fold_reuse", "world_mask", "sched_thr", "profile")  # (this library's scheduling scratch: which stages a pass could skip, which worlds a masked pass took)
  for f, ta in a.sim._data.items():
    tb = b.sim._data[f]
    if ta.dim() >= 1 and ta.shape[0] == n and f not in skip and f not in private and not torch.equal(ta[quiet], tb[quiet]):
      bad.append("mjData." + f)
  assert not bad, (k, bad)
  return len(sa)


def run(make_env, device: str, num_envs: int = 64, steps: int = 70, capture: bool = True, post_make=None, g_kwargs: dict | None = None,
        reset_at: int | None = None) -> dict:
  """``reset_at``: before that step both environments are reset through their public ``reset()`` (the reference's eager reset of all
  environments -- what ``RslRlVecEnvWrapper.__init__`` does AFTER the graph was captured): the reference rebinds ``obs_buf`` and
  ``extras["log"]`` there, and the replays that follow must still hand out the tensors they write."""
  torch.manual_seed(0)
  a = make_env(num_envs, device, _edit)
  b = make_env(num_envs, device, _edit)
  if post_make is not None:
    post_make(a)
    post_make(b)
  a.reset()
  b.reset()
  g = GraphedRlEnv(b, capture=capture, **(g_kwargs or {}))
  gen = torch.Generator(device=device)
  gen.manual_seed(3)
  robot = a.scene["robot"]
  ix = robot.indexing
  cmd_a, cmd_b = a.command_manager.get_term("twist"), b.command_manager.get_term("twist")
  ev = a.event_manager
  push_cfg = next(c for c in ev._mode_term_cfgs["interval"])
  reset_cfg = next(c for c in ev._mode_term_cfgs["reset"] if c.func.__name__ == "reset_root_state_uniform")
  stats = {"resets": 0, "resamples": 0, "pushes": 0, "quiet_env_steps": 0, "forward_steps": 0}
  dt = a.step_dt
  na = sum(a.action_manager.action_term_dim)
  prev_log: dict = {}
  for k in range(steps):
    if reset_at is not None and k == reset_at:
      a.reset()
      g.reset()
      stats["public_resets"] = stats.get("public_resets", 0) + 1
    _sync(a, b)
    action = torch.rand((num_envs, na), device=device, generator=gen) * 2 - 1
    if k > 20:
      action[: num_envs // 8] *= 6.0  # a few robots flail and fall: fell_over terminations besides the time-outs
    # which envs will draw random numbers in this step (a function of the synced pre-step state)
    resample = (cmd_a.time_left - dt) <= 0.0
    push = (ev._interval_term_time_left[0] - dt) < 1e-6
    terr = a.scene.terrain if getattr(getattr(a.scene, "terrain", None), "terrain_origins", None) is not None else None
    levels_before = terr.terrain_levels.clone() if terr is not None else None
    obs_a, rew_a, term_a, to_a, _ = a.step(action.clone())
    obs_b, rew_b, term_b, to_b, _ = g.step(action.clone())
    if device != "cpu":
      torch.cuda.synchronize()
    assert torch.equal(term_a, term_b) and torch.equal(to_a, to_b), k
    assert torch.equal(rew_a, rew_b), (k, (rew_a - rew_b).abs().max())
    _same_reward_state(a, b, k)
    reset = term_a | to_a
    _same_logs(a, b, k, reset)
    # a step without resets leaves extras["log"] as the last step with resets wrote it (the reference runs _reset_idx only then)
    cur_log = {key: float(v.float().mean().item()) for key, v in b.extras["log"].items() if isinstance(v, torch.Tensor)}
    if not bool(reset.any()) and prev_log and not (reset_at is not None and k == reset_at):
      assert cur_log == prev_log, (k, {key: (cur_log[key], prev_log.get(key)) for key in cur_log if cur_log[key] != prev_log.get(key)})
      stats["steps_with_a_kept_log"] = stats.get("steps_with_a_kept_log", 0) + 1
    prev_log = cur_log
    if terr is not None:  # the terrain curriculum (terrain_levels_vel): the same moves; a random level only past the hardest row
      tb = b.scene.terrain
      drew = reset & (levels_before + 1 >= terr.max_terrain_level)
      assert torch.equal(terr.terrain_levels[~drew], tb.terrain_levels[~drew]), k
      assert torch.equal(terr.env_origins[~drew], tb.env_origins[~drew]), k
      for t in (terr, tb):
        assert bool((t.terrain_levels >= 0).all()) and bool((t.terrain_levels < t.max_terrain_level).all())
        assert torch.equal(t.env_origins, t.terrain_origins[t.terrain_levels, t.terrain_types])
      stats["level_moves"] = stats.get("level_moves", 0) + int((terr.terrain_levels != levels_before).sum())
      stats["level_draws"] = stats.get("level_draws", 0) + int((drew & (terr.terrain_levels != levels_before + 1)).sum())
    noisy = reset | resample | push
    quiet = ~noisy
    for grp in obs_a:
      assert torch.equal(obs_a[grp][quiet], obs_b[grp][quiet]), (k, grp, (obs_a[grp][quiet] - obs_b[grp][quiet]).abs().max())
    for f in ("qpos", "qvel", "ctrl", "xpos", "cvel", "qacc_warmstart"):
      assert torch.equal(getattr(a.sim.data, f)[quiet], getattr(b.sim.data, f)[quiet]), (k, f)
    assert torch.equal(cmd_a.command[quiet], cmd_b.command[quiet]) and torch.equal(cmd_a.time_left[quiet], cmd_b.time_left[quiet])
    assert torch.equal(a.episode_length_buf, b.episode_length_buf)
    assert torch.equal(a.action_manager.action, b.action_manager.action) and torch.equal(a.action_manager.prev_action, b.action_manager.prev_action)
    _same_event_bookkeeping(a, b, k)
    stats["state_tensors_swept"] = _state_sweep(a, b, quiet, k)
    # ---- the environments that drew random numbers: same distribution, checked against the configuration
    for env in (a, b):
      d = env.sim.data
      if reset.any():
        q, v = d.qpos[reset], d.qvel[reset]
        root0 = env.scene["robot"].data.default_root_state[reset]
        org = env.scene.env_origins[reset]
        pr = reset_cfg.params["pose_range"]
        dx = q[:, 0:3] - root0[:, 0:3] - org
        assert bool((dx[:, 0].abs() <= pr["x"][1] + 1e-6).all()) and bool((dx[:, 1].abs() <= pr["y"][1] + 1e-6).all()) and bool((dx[:, 2].abs() <= 1e-6).all())
        assert bool(((q[:, 3:7].norm(dim=1) - 1).abs() < 1e-5).all()) and bool((q[:, 4:6].abs() < 1e-6).all())  # a yaw rotation of the default orientation
        pushed_now = push[reset]
        assert bool((v[~pushed_now] == 0).all())  # velocity_range {} and joint velocity scale (0, 0); a push may follow in the same step
        jp = q[:, 7:]
        lim = env.scene["robot"].data.soft_joint_pos_limits[reset]
        exp = env.scene["robot"].data.default_joint_pos[reset].clamp(lim[..., 0], lim[..., 1])
        assert torch.equal(jp, exp)
        assert bool((env.episode_length_buf[reset] == 0).all()) and bool((env.action_manager.action[reset] == 0).all())
      cm = env.command_manager.get_term("twist")
      c, rg = cm.command, cm.cfg.ranges
      assert bool((c[:, 0] >= rg.lin_vel_x[0] - 1e-6).all()) and bool((c[:, 0] <= rg.lin_vel_x[1] + 1e-6).all())
      assert bool((c[:, 1] >= rg.lin_vel_y[0] - 1e-6).all()) and bool((c[:, 1] <= rg.lin_vel_y[1] + 1e-6).all())
      assert bool((c[:, 2] >= rg.ang_vel_z[0] - 1e-6).all()) and bool((c[:, 2] <= rg.ang_vel_z[1] + 1e-6).all())
      rs = resample | reset
      if rs.any():
        lo, hi = cm.cfg.resampling_time_range  # (a reset resamples, then the same step's compute() takes dt off)
        assert bool((cm.time_left[rs] >= lo - dt - 1e-6).all()) and bool((cm.time_left[rs] <= hi + 1e-6).all())
      if push.any():
        tl = env.event_manager._interval_term_time_left[0][push]
        assert bool((tl >= push_cfg.interval_range_s[0] - 1e-6).all()) and bool((tl <= push_cfg.interval_range_s[1] + 1e-6).all())
    if (push & ~reset).any():
      # a push adds U(range) to the root's world velocity: A and B differ there by at most the width of the range, x and y only
      m = push & ~reset
      dv = (a.sim.data.qvel[m][:, :3] - b.sim.data.qvel[m][:, :3]).abs()
      w = push_cfg.params["velocity_range"]
      assert bool((dv[:, 0] <= w["x"][1] - w["x"][0] + 1e-5).all()) and bool((dv[:, 1] <= w["y"][1] - w["y"][0] + 1e-5).all()) and bool((dv[:, 2] <= 1e-5).all())
      assert dv.max() > 0, "two independent pushes came out identical"
    stats["resets"] += int(reset.sum()); stats["resamples"] += int((resample & ~reset).sum()); stats["pushes"] += int(push.sum())
    stats["quiet_env_steps"] += int(quiet.sum()); stats["forward_steps"] += int(reset.any())
  stats["graph"] = g.graph is not None
  return stats


def run_fused_vs_torch(make_env, device: str, num_envs: int = 256, steps: int = 60, edit=_edit, noise: bool = True, a_kwargs: dict | None = None,
                       reward_tol: float = 0.0, b_kwargs: dict | None = None) -> dict:
  """The event / command terms as HIP launches (``fused_terms=True``, mjlab_amd/env_terms.py) against their torch restatements
  (``fused_terms=False``) inside two captured environments of the same task: both consume the same block of uniforms per step
  (the same seed is set before each replay), so EVERY environment -- the ones that reset, resample or get pushed included --
  must agree: decisions bit for bit, values to rounding (the torch chain's jit-scripted helpers may contract into fma)."""

  def both(cfg):
    edit(cfg)
    if noise:  # the policy group corrupted as the task ships it, the critic group clean (tasks/velocity/velocity_env_cfg.py:118-126)
      cfg.observations.policy.enable_corruption = True
      cfg.observations.critic.enable_corruption = False

  torch.manual_seed(0)
  a = make_env(num_envs, device, both)
  b = make_env(num_envs, device, both)
  a.reset()
  b.reset()
  ga, gb = GraphedRlEnv(a, fused_terms=True, **(a_kwargs or {})), GraphedRlEnv(b, fused_terms=False, **(b_kwargs or {}))
  gen = torch.Generator(device=device)
  gen.manual_seed(9)
  na = sum(a.action_manager.action_term_dim)
  worst = {"obs": 0.0, "qpos": 0.0, "qvel": 0.0, "command": 0.0, "time_left": 0.0}
  stats = {"resets": 0, "pushes": 0, "resamples": 0}  # (resamples: commands that ran out inside an episode)
  for k in range(steps):
    _sync(a, b)
    action = torch.rand((num_envs, na), device=device, generator=gen) * 2 - 1
    if k > 20:
      action[: num_envs // 8] *= 6.0
    push = (a.event_manager._interval_term_time_left[0] - a.step_dt) < 1e-6
    stats["resamples"] += sum(int(((a.command_manager.get_term(nm).time_left - a.step_dt) <= 0.0).sum()) for nm in a.command_manager.active_terms)
    outs = []
    for g in (ga, gb):
      torch.manual_seed(1000 + k)
      outs.append(g.step(action.clone()))
    torch.cuda.synchronize()
    (obs_a, rew_a, term_a, to_a, _), (obs_b, rew_b, term_b, to_b, _) = outs
    assert torch.equal(term_a, term_b) and torch.equal(to_a, to_b), k
    if reward_tol == 0.0:
      assert torch.equal(rew_a, rew_b), k
    else:
      worst["reward"] = max(worst.get("reward", 0.0), float((rew_a - rew_b).abs().max()))
      assert worst["reward"] <= reward_tol, (k, worst["reward"])
    for grp in obs_a:
      worst["obs"] = max(worst["obs"], float((obs_a[grp] - obs_b[grp]).abs().max()))
    if noise and "critic" in obs_a and obs_a["critic"].shape == obs_a["policy"].shape:
      # the policy group is the critic group plus UniformNoiseCfg noise (tasks/velocity/velocity_env_cfg.py:88-116): inside its bounds,
      # centred, and using the whole width of each term's interval
      om = a.observation_manager
      diff, c0 = obs_a["policy"] - obs_a["critic"], 0
      for cfg, dim in zip(om._group_obs_term_cfgs["policy"], om._group_obs_term_dim["policy"], strict=True):
        d = diff[:, c0 : c0 + dim[0]]
        c0 += dim[0]
        if cfg.noise is None:
          assert float(d.abs().max()) == 0.0
          continue
        lo, hi = float(cfg.noise.n_min), float(cfg.noise.n_max)
        tol = 1e-6 * max(1.0, float(obs_a["critic"].abs().max()))
        assert float(d.min()) >= lo - tol and float(d.max()) <= hi + tol, (k, lo, hi, float(d.min()), float(d.max()))
        assert float(d.max()) - float(d.min()) > 0.8 * (hi - lo) and abs(float(d.mean())) < 0.1 * (hi - lo), (k, lo, hi, float(d.mean()))
      stats["noise_checks"] = stats.get("noise_checks", 0) + 1
    for f in ("qpos", "qvel"):
      worst[f] = max(worst[f], float((getattr(a.sim.data, f) - getattr(b.sim.data, f)).abs().max()))
    for name in a.command_manager.active_terms:
      ta, tb = a.command_manager.get_term(name), b.command_manager.get_term(name)
      worst["command"] = max(worst["command"], float((ta.command - tb.command).abs().max()))
      worst["time_left"] = max(worst["time_left"], float((ta.time_left - tb.time_left).abs().max()))
      assert torch.equal(ta.command_counter, tb.command_counter), k
      for flag in ("is_heading_env", "is_standing_env"):
        if hasattr(ta, flag):
          assert torch.equal(getattr(ta, flag), getattr(tb, flag)), (k, flag)
    assert torch.equal(a.episode_length_buf, b.episode_length_buf)
    for ia, ib in zip(a.event_manager._interval_term_time_left, b.event_manager._interval_term_time_left, strict=True):
      worst["time_left"] = max(worst["time_left"], float((ia - ib).abs().max()))
    stats["resets"] += int((term_a | to_a).sum()); stats["pushes"] += int(push.sum())
  stats["worst"] = worst
  return stats


def run_fused_vs_torch_tracking(make_env, device: str, num_envs: int = 128, steps: int = 40) -> dict:
  """``run_fused_vs_torch`` for ``Mjlab-Tracking-Flat-Unitree-G1``: MotionCommand's state write and relative body poses as HIP launches
  against their torch restatements, on the same uniforms, every environment compared."""

  def edit(cfg):
    cfg.events.push_robot.interval_range_s = (0.1, 0.4)

  torch.manual_seed(0)
  a = make_env(num_envs, device, edit)
  b = make_env(num_envs, device, edit)
  a.reset()
  b.reset()
  ga, gb = GraphedRlEnv(a, fused_terms=True, fused_relative_poses=True), GraphedRlEnv(b, fused_terms=False)
  gen = torch.Generator(device=device)
  gen.manual_seed(9)
  ca, cb = a.command_manager.get_term("motion"), b.command_manager.get_term("motion")
  for g in (ga, gb):
    torch.manual_seed(999)
    g.step(torch.zeros((num_envs, 29), device=device))  # (MotionCommand._update_metrics creates two of its metric entries on its first call)
  total = ca.motion.time_step_total
  worst = {"obs": 0.0, "qpos": 0.0, "qvel": 0.0, "body_pos_relative_w": 0.0, "body_quat_relative_w": 0.0, "reward": 0.0}
  stats = {"resets": 0, "ended": 0}
  for k in range(steps):
    _sync(a, b)
    if k % 7 == 3:
      ca.time_steps[: num_envs // 8] = total - 2
      cb.time_steps[: num_envs // 8] = total - 2
    action = (torch.rand((num_envs, 29), device=device, generator=gen) * 2 - 1) * 0.3
    stats["ended"] += int(((ca.time_steps + 1) >= total).sum())
    outs = []
    for g in (ga, gb):
      torch.manual_seed(1000 + k)
      outs.append(g.step(action.clone()))
    torch.cuda.synchronize()
    (obs_a, rew_a, term_a, to_a, _), (obs_b, rew_b, term_b, to_b, _) = outs
    assert torch.equal(term_a, term_b) and torch.equal(to_a, to_b), k
    assert torch.equal(ca.time_steps, cb.time_steps) and torch.equal(ca.bin_failed_count, cb.bin_failed_count), k
    worst["reward"] = max(worst["reward"], float((rew_a - rew_b).abs().max()))
    for grp in obs_a:
      worst["obs"] = max(worst["obs"], float((obs_a[grp] - obs_b[grp]).abs().max()))
    for f in ("qpos", "qvel"):
      worst[f] = max(worst[f], float((getattr(a.sim.data, f) - getattr(b.sim.data, f)).abs().max()))
    for f in ("body_pos_relative_w", "body_quat_relative_w"):
      worst[f] = max(worst[f], float((getattr(ca, f) - getattr(cb, f)).abs().max()))
    stats["resets"] += int((term_a | to_a).sum())
  stats["worst"] = worst
  stats["relative_rounding"] = getattr(ga, "relative_rounding", None)  # which rounding of the relative-poses launch reproduced the reference's chain (-1: none, torch chain in use)
  return stats


def run_tracking(make_env, device: str, num_envs: int = 32, steps: int = 40, capture: bool = True, g_kwargs: dict | None = None) -> dict:
  """The same teacher-forced comparison for ``Mjlab-Tracking-Flat-Unitree-G1`` (``MotionCommand``: resets to motion phases drawn by
  the adaptive sampler, resampling when a motion ends, the sampler's global failure statistics)."""

  def edit(cfg):
    for group in ("policy", "critic"):
      getattr(cfg.observations, group).enable_corruption = False
    cfg.events.push_robot.interval_range_s = (0.1, 0.4)

  torch.manual_seed(0)
  a = make_env(num_envs, device, edit)
  b = make_env(num_envs, device, edit)
  a.reset()
  b.reset()
  g = GraphedRlEnv(b, capture=capture, **(g_kwargs or {}))
  gen = torch.Generator(device=device)
  gen.manual_seed(5)
  cmd_a, cmd_b = a.command_manager.get_term("motion"), b.command_manager.get_term("motion")
  ev = a.event_manager
  # MotionCommand._update_metrics creates two of its metric entries on its first call: one step on each side, so that both hold them
  a.step(torch.zeros((num_envs, 29), device=device))
  g.step(torch.zeros((num_envs, 29), device=device))
  total = cmd_a.motion.time_step_total
  stats = {"resets": 0, "ended": 0, "pushes": 0, "quiet_env_steps": 0}
  dt = a.step_dt
  for k in range(steps):
    _sync(a, b)
    if k % 7 == 3:  # some motions run out: resample in _update_command without a reset
      cmd_a.time_steps[: num_envs // 8] = total - 2
      cmd_b.time_steps[: num_envs // 8] = total - 2
    action = (torch.rand((num_envs, 29), device=device, generator=gen) * 2 - 1) * 0.3
    ended = (cmd_a.time_steps + 1) >= total
    push = (ev._interval_term_time_left[0] - dt) < 1e-6
    obs_a, rew_a, term_a, to_a, _ = a.step(action.clone())
    obs_b, rew_b, term_b, to_b, _ = g.step(action.clone())
    if device != "cpu":
      torch.cuda.synchronize()
    assert torch.equal(term_a, term_b) and torch.equal(to_a, to_b), k
    assert torch.equal(rew_a, rew_b), (k, (rew_a - rew_b).abs().max())
    _same_reward_state(a, b, k)
    reset = term_a | to_a
    _same_logs(a, b, k, reset)
    quiet = ~(reset | ended | push)
    for grp in obs_a:
      assert torch.equal(obs_a[grp][quiet], obs_b[grp][quiet]), (k, grp, (obs_a[grp][quiet] - obs_b[grp][quiet]).abs().max())
    for f in ("qpos", "qvel", "ctrl", "xpos", "cvel"):
      assert torch.equal(getattr(a.sim.data, f)[quiet], getattr(b.sim.data, f)[quiet]), (k, f)
    assert torch.equal(cmd_a.time_steps[quiet], cmd_b.time_steps[quiet])
    assert torch.equal(cmd_a.body_pos_relative_w[quiet], cmd_b.body_pos_relative_w[quiet]) and torch.equal(cmd_a.body_quat_relative_w[quiet], cmd_b.body_quat_relative_w[quiet])
    assert torch.equal(cmd_a.bin_failed_count, cmd_b.bin_failed_count), k  # the sampler's global statistics: no randomness in them
    assert torch.equal(a.episode_length_buf, b.episode_length_buf)
    _same_event_bookkeeping(a, b, k)
    stats["state_tensors_swept"] = _state_sweep(a, b, quiet, k)
    # ---- resampled environments: a motion frame plus the cfg's noise (MotionCommand._resample_command), on both sides
    rs = reset | ended
    if rs.any():
      for env, cm in ((a, cmd_a), (b, cmd_b)):
        t = cm.time_steps[rs]
        assert bool((t >= 0).all()) and bool((t < total).all())
        q, v = env.sim.data.qpos[rs], env.sim.data.qvel[rs]
        # the command advanced by one frame after the resample only for resets (reset -> compute -> time_steps += 1); compare with both
        for off in (0, 1):
          tt = (t - off).clamp(min=0)
          jdev = (q[:, 7:] - cm.motion.joint_pos[tt]).abs().amax(dim=1)
          ok = jdev <= 0.1 + 1e-5 if off == 0 else ok | (jdev <= 0.1 + 1e-5)
        assert bool(ok.all()), jdev.max()
        pr = cm.cfg.pose_range
        root = cm.motion._body_pos_w[(t - 1).clamp(min=0), cm.body_indexes[0]] + env.scene.env_origins[rs]
        root0 = cm.motion._body_pos_w[t, cm.body_indexes[0]] + env.scene.env_origins[rs]
        dz = torch.minimum((q[:, 2] - root[:, 2]).abs(), (q[:, 2] - root0[:, 2]).abs())
        pushed_now = push[rs]
        assert bool((dz <= pr["z"][1] + 1e-5).all()), dz.max()
        assert bool((env.episode_length_buf[reset] == 0).all())
    stats["resets"] += int(reset.sum()); stats["ended"] += int((ended & ~reset).sum()); stats["pushes"] += int(push.sum()); stats["quiet_env_steps"] += int(quiet.sum())
  stats["graph"] = g.graph is not None
  stats["relative_rounding"] = getattr(g, "relative_rounding", None)
  return stats


def generic_events_edit(cfg):
  """The velocity task with two of the reference's stock event terms GraphedRlEnv has no restatement for, appended to its events:
  ``reset_scene_to_default`` (reset mode, LAST: whatever the random reset terms drew, a reset environment ends in the default state --
  deterministic, so comparable bit for bit) and ``apply_external_force_torque`` (interval mode: random wrenches on every body)."""
  import dataclasses

  from mjlab.envs.mdp import events as ref_events
  from mjlab.managers.manager_term_config import EventTermCfg

  _edit(cfg)
  base = cfg.events
  extra = [("zz_default", EventTermCfg, dataclasses.field(default_factory=lambda: EventTermCfg(func=ref_events.reset_scene_to_default, mode="reset"))),
           ("zz_wrench", EventTermCfg, dataclasses.field(default_factory=lambda: EventTermCfg(
             func=ref_events.apply_external_force_torque, mode="interval", interval_range_s=(0.06, 0.3), params={"force_range": (-5.0, 5.0), "torque_range": (-1.0, 1.0)})))]
  plus = dataclasses.make_dataclass("EventCfgWithStockTerms", extra, bases=(type(base),))
  cfg.events = plus(**{f.name: getattr(base, f.name) for f in dataclasses.fields(base)})


def run_generic_events(make_env, device: str, num_envs: int = 32, steps: int = 60, capture: bool = True) -> dict:
  """GraphedRlEnv with event terms it runs through ``_generic_event`` (the reference's function on all environments, kept where the mask
  is set) against the eager reference, teacher-forced."""
  torch.manual_seed(0)
  a = make_env(num_envs, device, generic_events_edit)
  b = make_env(num_envs, device, generic_events_edit)
  a.reset()
  b.reset()
  g = GraphedRlEnv(b, capture=capture)
  assert [fn for fn, _ in g._reset_terms].count("generic_event") == 1 and sum(not isinstance(t[2], torch.Tensor) for t in g._interval_terms) == 1
  gen = torch.Generator(device=device)
  gen.manual_seed(5)
  ev = a.event_manager
  names = ev.active_terms["interval"]
  ipush, iwrench = names.index("push_robot"), names.index("zz_wrench")
  cmd_a = a.command_manager.get_term("twist")
  robot = a.scene["robot"]
  bodies = robot.indexing.body_ids
  dt = a.step_dt
  na = sum(a.action_manager.action_term_dim)
  stats = {"resets": 0, "wrenches": 0, "default_states_compared": 0, "quiet_env_steps": 0}
  for k in range(steps):
    _sync(a, b)
    action = torch.rand((num_envs, na), device=device, generator=gen) * 2 - 1
    if k > 20:
      action[: num_envs // 8] *= 6.0
    resample = (cmd_a.time_left - dt) <= 0
    push = (ev._interval_term_time_left[ipush] - dt) < 1e-6
    wrench = (ev._interval_term_time_left[iwrench] - dt) < 1e-6
    xfrc_before = a.sim.data.xfrc_applied.clone()
    _, rew_a, term_a, to_a, _ = a.step(action)
    _, rew_b, term_b, to_b, _ = g.step(action)
    if device != "cpu":
      torch.cuda.synchronize()
    assert torch.equal(term_a, term_b) and torch.equal(to_a, to_b) and torch.equal(rew_a, rew_b), k
    reset = term_a | to_a
    # ---- the reset term run generically: a reset environment ends in the default state on both sides (a push may follow)
    m = reset & ~push
    if m.any():
      for f in ("qpos", "qvel"):
        assert torch.equal(getattr(a.sim.data, f)[m], getattr(b.sim.data, f)[m]), (k, f)
      d0 = robot.data.default_root_state[m].clone()
      d0[:, :3] += a.scene.env_origins[m]
      assert torch.equal(b.sim.data.qpos[m][:, :7], d0[:, :7]) and torch.equal(b.sim.data.qpos[m][:, 7:], robot.data.default_joint_pos[m])
      stats["default_states_compared"] += int(m.sum())
    # ---- the interval term run generically: wrenches in range where its timer ran out, untouched elsewhere
    for env in (a, b):
      x = env.sim.data.xfrc_applied[:, bodies]
      if wrench.any():
        xw = x[wrench]
        assert bool((xw[..., :3].abs() <= 5.0).all()) and bool((xw[..., 3:].abs() <= 1.0).all()) and float(xw[..., :3].abs().max()) > 1.0
        tl = env.event_manager._interval_term_time_left[iwrench][wrench]
        assert bool((tl >= 0.06 - 1e-6).all()) and bool((tl <= 0.3 + 1e-6).all())
      keep = ~wrench & ~reset
      assert torch.equal(env.sim.data.xfrc_applied[keep], xfrc_before[keep]), k
      clr = reset & ~wrench
      assert bool((env.sim.data.xfrc_applied[clr] == 0).all())  # (EntityData.clear_state of the reset, no new wrench)
    if wrench.any():
      assert not torch.equal(a.sim.data.xfrc_applied[wrench], b.sim.data.xfrc_applied[wrench]), "two independent draws came out identical"
    assert torch.equal(a.event_manager._interval_term_time_left[iwrench][~wrench], b.event_manager._interval_term_time_left[iwrench][~wrench])
    quiet = ~(reset | resample | push | wrench)
    _same_event_bookkeeping(a, b, k)
    _state_sweep(a, b, quiet, k)
    stats["resets"] += int(reset.sum()); stats["wrenches"] += int(wrench.sum()); stats["quiet_env_steps"] += int(quiet.sum())
  stats["graph"] = g.graph is not None
  return stats


def toy_command_edit(cfg):
  """The velocity task with its command term replaced by a class GraphedRlEnv has no restatement for: a CommandTerm subclass written
  here (uniform planar velocity + yaw rate, resampled by ``_resample_command`` with plain tensor indexing; ``_update_command`` clips the
  yaw rate by the speed -- deterministic), no curriculum on it."""
  import dataclasses

  from mjlab.managers.command_manager import CommandTerm
  from mjlab.managers.manager_term_config import CommandTermCfg

  class ToyVelocityCommand(CommandTerm):
    def __init__(self, cfg, env):
      super().__init__(cfg, env)
      self.robot = env.scene["robot"]
      self.vel_command_b = torch.zeros(self.num_envs, 3, device=self.device)
      self.metrics["error_vel_xy"] = torch.zeros(self.num_envs, device=self.device)

    @property
    def command(self):
      return self.vel_command_b

    def _update_metrics(self):
      horizon = self.cfg.resampling_time_range[1] / self._env.step_dt
      self.metrics["error_vel_xy"] += torch.norm(self.vel_command_b[:, :2] - self.robot.data.root_link_lin_vel_b[:, :2], dim=-1) / horizon

    def _resample_command(self, env_ids):
      r = torch.rand((len(env_ids), 3), device=self.device)
      self.vel_command_b[env_ids] = r * 2.0 - 1.0

    def _update_command(self):
      speed = torch.norm(self.vel_command_b[:, :2], dim=-1)
      self.vel_command_b[:, 2] = torch.minimum(self.vel_command_b[:, 2], 1.2 - speed)

  @dataclasses.dataclass(kw_only=True)
  class ToyVelocityCommandCfg(CommandTermCfg):
    class_type: type = ToyVelocityCommand

  _edit(cfg)
  cfg.commands.twist = ToyVelocityCommandCfg(resampling_time_range=(0.2, 0.5))
  if getattr(cfg, "curriculum", None) is not None and hasattr(cfg.curriculum, "command_vel"):
    cfg.curriculum.command_vel = None


def run_toy_command(make_env, device: str, num_envs: int = 32, steps: int = 60, capture: bool = True) -> dict:
  """GraphedRlEnv with a command term of a class it has no restatement for (``_generic_command_resample``) against the eager reference,
  teacher-forced: untouched environments bit for bit over every state tensor, resampled ones in range, the deterministic update bit for bit."""
  torch.manual_seed(0)
  a = make_env(num_envs, device, toy_command_edit)
  b = make_env(num_envs, device, toy_command_edit)
  a.reset()
  b.reset()
  g = GraphedRlEnv(b, capture=capture)
  cmd_a, cmd_b = a.command_manager.get_term("twist"), b.command_manager.get_term("twist")
  assert type(cmd_b).__name__ == "ToyVelocityCommand" and id(cmd_b) in g._generic_commands and len(g._generic_commands[id(cmd_b)]) == 1
  gen = torch.Generator(device=device)
  gen.manual_seed(6)
  ev = a.event_manager
  dt = a.step_dt
  na = sum(a.action_manager.action_term_dim)
  stats = {"resets": 0, "resamples": 0, "quiet_env_steps": 0}
  for k in range(steps):
    _sync(a, b)
    action = torch.rand((num_envs, na), device=device, generator=gen) * 2 - 1
    if k > 20:
      action[: num_envs // 8] *= 6.0
    resample = (cmd_a.time_left - dt) <= 0
    push = (ev._interval_term_time_left[0] - dt) < 1e-6
    counter_before = cmd_a.command_counter.clone()
    obs_a, rew_a, term_a, to_a, _ = a.step(action)
    obs_b, rew_b, term_b, to_b, _ = g.step(action)
    if device != "cpu":
      torch.cuda.synchronize()
    assert torch.equal(term_a, term_b) and torch.equal(to_a, to_b) and torch.equal(rew_a, rew_b), k
    reset = term_a | to_a
    _same_logs(a, b, k, reset)
    drew = reset | resample
    for c in (cmd_a, cmd_b):
      v = c.command
      assert bool((v[:, :2].abs() <= 1.0).all()) and bool((v[:, 2] <= 1.2 - torch.norm(v[:, :2], dim=-1) + 1e-6).all())
      assert bool((c.time_left[drew] >= 0.2 - dt - 1e-6).all()) and bool((c.time_left[drew] <= 0.5 + 1e-6).all())
    assert torch.equal(cmd_a.command_counter, cmd_b.command_counter)
    assert torch.equal(cmd_b.command_counter[~reset], (counter_before + resample.to(counter_before.dtype))[~reset]) and bool((cmd_b.command_counter[reset] == 1).all())
    if drew.any():
      assert not torch.equal(cmd_a.command[drew], cmd_b.command[drew]), "two independent draws came out identical"
    quiet = ~(reset | resample | push)
    for grp in obs_a:
      assert torch.equal(obs_a[grp][quiet], obs_b[grp][quiet]), (k, grp)
    _same_event_bookkeeping(a, b, k)
    _state_sweep(a, b, quiet, k)
    stats["resets"] += int(reset.sum()); stats["resamples"] += int((resample & ~reset).sum()); stats["quiet_env_steps"] += int(quiet.sum())
  stats["graph"] = g.graph is not None
  return stats
