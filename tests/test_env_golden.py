"""Row N1, driver-verifiable: the HIP path against what the REFERENCE'S OWN environment computed, without the reference tree.

tests/golden/env_<scene>.npz (tools/make_env_golden.py, run where the reference checkout exists) hold, for the two tasks
``north_star`` names, every hand-over point of the reference's ``ManagerBasedRlEnv.step`` (reference
envs/manager_based_rl_env.py:106-147) over 80 / 20 control steps x 16 envs: the state it entered with, the action and the ctrl
its action manager wrote, the state after its 4 x ``sim.step()``, ``terminated`` / ``time_out``, the state after its reset events +
``sim.forward()``, the state after its interval pushes, and the observation groups its ObservationManager assembled (noise off).

Here each recorded step is replayed on a ``Simulation`` -- pre-step state in, ctrl = offset + scale * action, 4 substeps, the
recorded reset / push rows written where the reference's events wrote them, ``forward()`` where the reference ran it -- and the
observation TERMS are rebuilt from mjData (entity/data.py:190-516 semantics) and compared with the reference's:

  CPU   over the fp32 oracle (the engine the recording ran on): post-step state bit for bit, terms to float rounding --
        pins the replay + rebuild code itself, in every round;
  GPU   over ``mjlab_amd.Simulation`` + ``EntityReadback`` (one fused launch): the product path against the reference
        environment's numbers, on the driver's box, no reference tree needed.

Tolerances are per term, absolute + 1e-5 relative, stated at the assertion (measured on the GPU x 3).
"""

import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "tests"))

GOLD = ROOT / "tests" / "golden"
SCENES = ("g1_velocity_flat", "g1_tracking_flat", "g1_velocity_rough")


# ------------------------------------------------------------------------------------------------ quaternion helpers (w, x, y, z)
def q_conj(q):
  return q * np.array([1.0, -1.0, -1.0, -1.0], q.dtype)


def q_mul(a, b):
  aw, ax, ay, az = np.moveaxis(a, -1, 0)
  bw, bx, by, bz = np.moveaxis(b, -1, 0)
  return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                   aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def q_apply(q, v):
  w, xyz = q[..., :1], q[..., 1:]
  t = 2.0 * np.cross(xyz, v)
  return v + w * t + np.cross(xyz, t)


def q_mat(q):
  w, x, y, z = np.moveaxis(q, -1, 0)
  return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                   2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                   2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], axis=-1).reshape(*q.shape[:-1], 3, 3)


def subtract_frames(t01, q01, t02, q02):
  """T12 = T01^-1 T02 (reference third_party/isaaclab/isaaclab/utils/math.py:826-857)."""
  q10 = q_conj(q01)
  return q_apply(q10, t02 - t01), q_mul(q10, q02)


# ------------------------------------------------------------------------------------------------ mjData -> entity quantities
class Derived:
  """What EntityData derives (reference entity/data.py:190-260,487-516), for the robot = bodies [root, nbody)."""

  def __init__(self, root, xpos, xquat, cvel, subtree_com, qpos, qvel):
    self.pos, self.quat = xpos[:, root:], xquat[:, root:]
    cv, sub = cvel[:, root:], subtree_com[:, root][:, None, :]
    self.lin_w = cv[..., 3:6] - np.cross(cv[..., 0:3], sub - self.pos)  # compute_velocity_from_cvel (data.py:20-31)
    self.ang_w = cv[..., 0:3]
    rq = q_conj(self.quat[:, 0])
    self.root_lin_b, self.root_ang_b = q_apply(rq, self.lin_w[:, 0]), q_apply(rq, self.ang_w[:, 0])
    self.gravity_b = q_apply(rq, np.broadcast_to(np.array([0.0, 0.0, -1.0], xpos.dtype), self.pos[:, 0].shape))
    self.joint_pos, self.joint_vel = qpos[:, 7:], qvel[:, 6:]

  @classmethod
  def from_readback(cls, rb):
    """The same quantities from the fused read-back kernel (mjlab_entity_readback)."""
    self = cls.__new__(cls)
    c = lambda t: t.detach().cpu().numpy().astype(np.float64)  # noqa: E731
    pose, vel = c(rb.body_link_pose_w), c(rb.body_link_vel_w)
    self.pos, self.quat, self.lin_w, self.ang_w = pose[..., :3], pose[..., 3:], vel[..., :3], vel[..., 3:]
    self.root_lin_b, self.root_ang_b, self.gravity_b = c(rb.root_link_lin_vel_b), c(rb.root_link_ang_vel_b), c(rb.projected_gravity_b)
    self.joint_pos, self.joint_vel = c(rb.joint_pos), c(rb.joint_vel)
    return self


def rebuild_terms(meta, z, k, dv: Derived) -> dict:
  """Observation terms of step k from the derived quantities (reference envs/mdp/observations.py:24-89,
  tasks/tracking/mdp/observations.py:18-76); pass-through terms (command, actions) come from the recording."""
  t = {"base_lin_vel": dv.root_lin_b, "base_ang_vel": dv.root_ang_b, "projected_gravity": dv.gravity_b,
       "joint_pos": dv.joint_pos - z["default_joint_pos"], "joint_vel": dv.joint_vel - z["default_joint_vel"],
       # last_action: the action manager's buffer, which _reset_idx zeroes for the envs it resets (managers/action_manager.py reset)
       "actions": np.where((z["terminated"][k] | z["time_out"][k])[:, None], 0.0, z["action"][k]), "command": z["command"][k]}
  if meta["scene"] == "g1_tracking_flat":
    a, bodies = meta["anchor_body"], meta["tracked_bodies"]
    ap, aq = dv.pos[:, a], dv.quat[:, a]
    p, q = subtract_frames(ap, aq, z["anchor_pos_w"][k].astype(ap.dtype), z["anchor_quat_w"][k].astype(ap.dtype))
    t["motion_anchor_pos_b"] = p
    t["motion_anchor_ori_b"] = q_mat(q)[..., :2].reshape(len(p), -1)
    pb, qb = subtract_frames(ap[:, None], aq[:, None], dv.pos[:, bodies], dv.quat[:, bodies])
    t["body_pos"] = pb.reshape(len(p), -1)
    t["body_ori"] = q_mat(qb)[..., :2].reshape(len(p), -1)
  return t


def split_obs(meta, obs, group):
  out, i = {}, 0
  for name, w in meta["obs_terms"][group]:
    out[name] = obs[:, i : i + w]
    i += w
  assert i == obs.shape[1]
  return out


def load(scene):
  return json.loads((GOLD / f"env_{scene}.json").read_text()), np.load(GOLD / f"env_{scene}.npz")


# per term: absolute floor (+ 1e-5 relative).  CPU: the recording's own engine -> float rounding of the rebuild.  GPU: fp32 HIP vs the
# fp32 oracle after 4 substeps under the grid line search (measured x 3; velocities carry the solver's qacc noise x 4 dt)
ATOL_CPU = {"base_lin_vel": 2e-6, "base_ang_vel": 2e-6, "projected_gravity": 1e-6, "joint_pos": 1e-6, "joint_vel": 1e-5, "actions": 0.0, "command": 0.0,
            "motion_anchor_pos_b": 1e-6, "motion_anchor_ori_b": 1e-6, "body_pos": 2e-6, "body_ori": 2e-6}
ATOL_GPU = {"base_lin_vel": 2e-4, "base_ang_vel": 1e-3, "projected_gravity": 2e-5, "joint_pos": 2e-5, "joint_vel": 5e-3, "actions": 0.0, "command": 0.0,
            "motion_anchor_pos_b": 2e-5, "motion_anchor_ori_b": 2e-5, "body_pos": 2e-5, "body_ori": 5e-5}
_MARGIN: dict = {}


# GPU: the share of (world, step) rows that must lie entirely within 1 x / WORST x a term's bound, and the sanity cap on the rest
# (worlds whose solve parted under the grid search: see state_tol in the GPU test)
WORST, ROWS_1X, ROWS_WORST, SANITY = 3.0, 0.99, 1.0, 3.0  # measured r04_v11: every row within 1 x but 1 of 320 (joint_vel, tracking: 1.08 x)
# the rough scene: a foot on a stair edge gains or loses a contact row between two fp32 engines now and then, and that world's step
# differs at the 1e-4 level (the parity gate's "deep / capped" class); such rows are bounded in number and size
# (measured r04_v27 and r05_v10: 3-4 of 1280 world-steps beyond 3 x, the worst at 1.9e-2 rad in joint_pos (820 x its bound) and 3.8 rad/s
# in joint_vel (760 x): one 20 ms control step with / without one foot contact's force.  The cap on such rows is that event's size
# x 3.5 -- 0.06 rad, 15 rad/s --, not infinity: VERDICT round 4, item 3d)
ROUGH = {"rows_worst": 0.995, "sanity": 3000.0}


def compare_terms(meta, z, k, dv, atol, tag, stats):
  """Per term: every element within WORST x (atol + 1e-5 |ref|) (CPU: within 1 x); the share of (world, step) rows entirely within
  1 x the bound is accumulated in `stats` and asserted by the caller over the whole replay."""
  ref = split_obs(meta, z["obs_critic"][k], "critic")  # the noise-free group carries every term of the policy group
  mine = rebuild_terms(meta, z, k, dv)
  for name, r in ref.items():
    a = np.asarray(mine[name], np.float64).reshape(r.shape)
    err = np.abs(a - r)
    bound = atol[name] + 1e-5 * np.abs(r)
    ratio = np.where(bound > 0, err / np.maximum(bound, 1e-300), np.where(err > 0, np.inf, 0.0)).max(axis=1)  # worst element per world
    key = (tag, meta["scene"], name)
    _MARGIN[key] = max(_MARGIN.get(key, 0.0), float(err.max()))
    st = stats.setdefault(name, [0, 0, 0.0, 0])
    st[0] += int((ratio <= 1.0).sum()); st[1] += len(ratio); st[2] = max(st[2], float(ratio.max())); st[3] += int((ratio <= WORST).sum())
    sanity = ROUGH["sanity"] if meta["scene"].endswith("rough") else SANITY
    assert ratio.max() <= (1.0 if tag == "cpu" else sanity), (meta["scene"], k, name, float(err.max()), atol[name], float(ratio.max()))
  return len(ref)


def teardown_module(module):
  out = ROOT / "gpurun_out"
  if out.is_dir() and _MARGIN:
    with open(out / "env_golden_margins.txt", "w") as f:
      for (tag, scene, name), v in sorted(_MARGIN.items()):
        if isinstance(v, tuple):
          f.write(f"{tag:4s} {scene:18s} {name:58s} " + " ".join(f"{x:.3e}" for x in v) + "\n")
        else:
          f.write(f"{tag:4s} {scene:18s} {name:58s} worst abs error {v:.3e}\n")


# ------------------------------------------------------------------------------------------------------------------------ CPU
def test_golden_files_hold_resets_pushes_and_every_term():
  for scene in SCENES:
    meta, z = load(scene)
    n, ns = meta["num_envs"], meta["num_steps"]
    assert z["pre_qpos"].shape == (ns, n, 36) and z["obs_critic"].shape[:2] == (ns, n) and meta["ls_parallel"] is True and meta["decimation"] == 4
    assert (z["terminated"] | z["time_out"]).sum() >= 16, "the recording must contain resets"
    assert z["forward_ran"].sum() >= 2 and z["forward_ran"].sum() < ns  # both branches of manager_based_rl_env.py:129-133
    assert sum(w for _, w in meta["obs_terms"]["policy"]) == z["obs_policy"].shape[2]
    # the chain of steps is closed: what a step ends with is what the next one starts from
    assert np.array_equal(z["final_qpos"][:-1], z["pre_qpos"][1:]) and np.array_equal(z["final_qvel"][:-1], z["pre_qvel"][1:])
    # ctrl is offset + scale * action (reference envs/mdp/actions/joint_actions.py), float32, no fused multiply-add
    assert np.array_equal(z["ctrl"], (z["action_offset"] + z["action_scale"] * z["action"]).astype(np.float32))
  _, zv = load("g1_velocity_flat")
  assert ((zv["final_qvel"] != zv["reset_qvel"]).any(axis=2)).sum() >= 1, "no interval push in the velocity recording"


def _replay(scene, make_sim, derive, atol, tag, state_tol):
  """The replay shared by the CPU and the GPU test.  `make_sim(meta, z)` -> object with set(field, rows|None, array),
  get(field), step4(), forward(); `derive(sim)` -> Derived.  `state_tol[field]` = (median, p90, p99, outlier line, largest
  share of world-steps beyond it) for the per-world relative error of the 4-substep state over all replayed world-steps."""
  meta, z = load(scene)
  sim = make_sim(meta, z)
  nterms = 0
  state_err: dict = {"qpos": [], "qvel": []}
  stats: dict = {}
  for k in range(meta["num_steps"]):
    for f in ("qpos", "qvel", "qacc_warmstart"):
      sim.set(f, None, z["pre_" + f][k])
    sim.set("ctrl", None, (z["action_offset"] + z["action_scale"] * z["action"][k]).astype(np.float32))
    sim.step4()
    for f in ("qpos", "qvel"):
      a, r = sim.get(f).astype(np.float64), z["post_" + f][k].astype(np.float64)
      err = np.abs(a - r).max(axis=1) / np.maximum(np.abs(r).max(axis=1), 1e-6)
      state_err[f].append(err)
      assert np.isfinite(a).all(), (scene, k, f)
    done = z["terminated"][k] | z["time_out"][k]
    if scene.startswith("g1_velocity"):
      # the task's terminations from the replayed state (velocity_env_cfg.py:219-223): fell_over = tilt beyond 70 degrees
      # (envs/mdp/terminations.py bad_orientation: acos(-projected_gravity_z) > limit), time_out = episode length
      g = derive(sim).gravity_b
      fell = np.arccos(np.clip(-g[:, 2], -1.0, 1.0)) > np.deg2rad(70.0)
      assert np.array_equal(fell, z["terminated"][k]), (k, fell, z["terminated"][k])
      assert np.array_equal(z["episode_length"][k] >= meta["max_episode_length"], z["time_out"][k])
    if done.any():  # the reference's reset events wrote these rows (:130), then sim.forward() on ALL worlds (:132)
      assert int(z["forward_ran"][k]) == 1
      rows = np.nonzero(done)[0]
      sim.set("qpos", rows, z["reset_qpos"][k][rows])
      sim.set("qvel", rows, z["reset_qvel"][k][rows])
      sim.forward()
    pushed = np.nonzero((z["final_qvel"][k] != z["reset_qvel"][k]).any(axis=1))[0]
    if len(pushed):  # interval event push_by_setting_velocity (:137-138) -- after the forward, before the observations
      sim.set("qvel", pushed, z["final_qvel"][k][pushed])
    nterms += compare_terms(meta, z, k, derive(sim), atol, tag, stats)
  for name, (ok, tot, worst, okw) in stats.items():
    _MARGIN[(tag, scene, name + f" rows within 1x / {WORST:.0f}x bound, worst ratio")] = (ok / tot, okw / tot, worst)
    if tag != "cpu":  # (world, step) rows whose every element is within the term's bound
      assert ok >= ROWS_1X * tot and okw >= (ROUGH["rows_worst"] if scene.endswith("rough") else ROWS_WORST) * tot, (scene, name, ok, okw, tot, worst)
  for f, errs in state_err.items():
    e = np.concatenate(errs)
    q = tuple(float(x) for x in (np.median(e), np.percentile(e, 90), np.percentile(e, 99)))
    out = float((e > state_tol[f][3]).mean())  # share of world-steps beyond the outlier line
    _MARGIN[(tag, scene, f"post_{f} median/p90/p99/max/outliers")] = q + (float(e.max()), out)
    assert all(a <= b for a, b in zip(q, state_tol[f][:3], strict=True)) and out <= state_tol[f][4], (scene, f, q, float(e.max()), out, state_tol[f])
  return nterms


class _OracleReplay:
  def __init__(self, meta, z):
    from mjlab_amd import robots
    from oracle.oracle import OracleSim

    self.model = robots.load_model(meta["scene"])
    self.o = OracleSim(self.model, meta["num_envs"], njmax=meta["njmax"], precision="f32", ls_parallel=meta["ls_parallel"])
    for key in z.files:
      if key.startswith("dr_"):
        self.o.expand_model_field(key[3:])[:] = z[key]
    self.o.forward(nthread=8)

  def set(self, f, rows, v):
    getattr(self.o, f)[slice(None) if rows is None else rows] = v

  def get(self, f):
    return getattr(self.o, f).copy()

  def step4(self):
    self.o.step(4, nthread=8)

  def forward(self):
    self.o.forward(nthread=8)


@pytest.mark.parametrize("scene", SCENES)
def test_replay_over_the_oracle_reproduces_the_reference_environment(scene):
  """The recording's own engine: the post-step state is reproduced bit for bit (state_tol 0), so every difference in a term would
  be the rebuild code's."""

  def derive(s):
    o = s.o
    return Derived(int(s.model.jnt_bodyid[0]), o.xpos.astype(np.float64), o.xquat.astype(np.float64), o.cvel.astype(np.float64),
                   o.subtree_com.astype(np.float64), o.qpos.astype(np.float64), o.qvel.astype(np.float64))

  n = _replay(scene, _OracleReplay, derive, ATOL_CPU, "cpu", {"qpos": (0.0,) * 5, "qvel": (0.0,) * 5})
  meta, _ = load(scene)
  assert n == meta["num_steps"] * len(meta["obs_terms"]["critic"])


def test_golden_is_what_the_reference_environment_computes_today(tmp_path):
  """Where the reference checkout exists: re-record and compare with the committed files (the recorder is deterministic)."""
  import subprocess

  import reference_env

  if reference_env.locate_reference() is None:
    pytest.skip("reference checkout not present")
  r = subprocess.run([sys.executable, str(ROOT / "tools" / "make_env_golden.py"), str(tmp_path)], capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-2000:]
  for scene in SCENES:
    a, b = np.load(tmp_path / f"env_{scene}.npz"), np.load(GOLD / f"env_{scene}.npz")
    assert sorted(a.files) == sorted(b.files)
    for key in a.files:
      assert np.array_equal(a[key], b[key]), (scene, key)
    assert json.loads((tmp_path / f"env_{scene}.json").read_text()) == json.loads((GOLD / f"env_{scene}.json").read_text())


# ------------------------------------------------------------------------------------------------------------------------ GPU
class _HipReplay:
  def __init__(self, meta, z):
    import torch

    from mjlab_amd import robots
    from mjlab_amd.entity_data import EntityReadback
    from mjlab_amd.sim import Simulation, SimulationCfg

    self.torch = torch
    model = robots.load_model(meta["scene"])
    self.sim = Simulation(meta["num_envs"], SimulationCfg(njmax=meta["njmax"], ls_parallel=meta["ls_parallel"]), model, "cuda:0")
    dr = [key[3:] for key in z.files if key.startswith("dr_")]
    self.sim.expand_model_fields(dr)
    for f in dr:
      getattr(self.sim.model, f)[:] = torch.from_numpy(z["dr_" + f]).cuda()
    self.sim.create_graph()
    self.sim.forward()
    self.rb = EntityReadback(self.sim)

  def set(self, f, rows, v):
    t = self.torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
    if rows is None:
      getattr(self.sim.data, f)[:] = t
    else:
      getattr(self.sim.data, f)[self.torch.from_numpy(rows).cuda()] = t

  def get(self, f):
    self.torch.cuda.synchronize()
    return getattr(self.sim.data, f).cpu().numpy()

  def step4(self):
    for _ in range(4):  # the reference's call pattern: ctrl write + sim.step(), decimation times (manager_based_rl_env.py:109-114)
      self.sim.step()

  def forward(self):
    self.sim.forward()


@pytest.mark.gpu
@pytest.mark.parametrize("scene", SCENES)
def test_hip_path_reproduces_the_reference_environment(scene):
  """``mjlab_amd.Simulation`` (the grid line search the reference configures, hipGraph replay) + ``EntityReadback`` against the
  reference environment's recorded numbers: the 4-substep state per world at north_star's level (qpos 1e-5; qvel carries the
  solve's noise: gate literals), termination flags equal, every observation term within its floor."""

  def derive(s):
    s.rb.update()
    s.torch.cuda.synchronize()
    return Derived.from_readback(s.rb)

  # per-world relative error of the 4-substep state over the 1280 / 320 replayed world-steps: (median, p90, p99, outlier line, share
  # beyond it).  Measured (r04_v11, recordings made over the fp32 oracle whose grid search compares candidates by cost differences
  # like the kernel: oracle/Makefile): qpos median 6.5e-8 / p90 2.3e-7 / p99 1.0e-5 / max 2.8e-5; qvel 1.7e-6 / 4.7e-6 / 3.6e-4 /
  # 8.5e-4 -- north_star's 1e-5 on the state at the p99, on the driver's box, against the reference environment's own run.  (Recorded
  # over the LITERAL fp32 grid search the same replay showed qpos p90 1.2e-5, max 5e-3: that search's 1e-4 floor in qacc)
  tol = {"qpos": (5e-7, 1e-6, 3e-5, 1e-4, 0.0), "qvel": (5e-6, 1.5e-5, 1e-3, 3e-3, 0.0)}
  if scene.endswith("rough"):  # (see ROUGH above)
    # measured r04_v27: qpos median 1.7e-8 / p90 4.3e-8 / p99 1.4e-7 / max 3.2e-4; qvel 6.4e-6 / 2.1e-5 / 7.3e-5, one world-step of 1280 at 0.74
    tol = {"qpos": (1e-7, 2e-7, 1e-6, 1e-3, 0.0), "qvel": (2e-5, 6e-5, 3e-4, 3e-2, 0.003)}
  n = _replay(scene, _HipReplay, derive, ATOL_GPU, "gpu", tol)
  meta, _ = load(scene)
  assert n == meta["num_steps"] * len(meta["obs_terms"]["critic"])
