"""The functional core of GraphedRlEnv (mjlab_amd/env_core.py) against what the EAGER REFERENCE did with the same inputs
(tests/golden/graphed_core_*.npz, recorded by tools/make_graphed_golden.py from the reference's unmodified ManagerBasedRlEnv): the
reset bookkeeping of ``_reset_idx`` and the log it leaves, RewardManager.compute's accumulation, the observation groups, the velocity
command's update -- reference-free, so the driver's GPU box runs it too (row f3 beyond the term kernels).

CPU: the torch twins.  ``-m gpu``: the same on device tensors AND the HIP launches the captured step uses on the GPU
(mjlab_masked_sums / mjlab_masked_fill_rows / mjlab_reward_accumulate through mjlab_amd/env_terms.py) against the same recorded truth."""

import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from mjlab_amd import env_core

GOLD = Path(__file__).resolve().parent / "golden"
SCENES = ("g1_velocity_flat", "g1_tracking_flat", "g1_tracking_flat_nse", "go1_velocity_rough")
VELOCITY_SCENES = ("g1_velocity_flat", "go1_velocity_rough")


def load(scene):
  return json.loads((GOLD / f"graphed_core_{scene}.json").read_text()), np.load(GOLD / f"graphed_core_{scene}.npz")


def wrap_to_pi(angles):
  """reference third_party/isaaclab/isaaclab/utils/math.py wrap_to_pi: [-pi, pi], +pi kept for positive multiples"""
  w = (angles + math.pi) % (2 * math.pi)
  return torch.where((w == 0) & (angles > 0), math.pi, w - math.pi)


def _t(z, key, dev):
  return torch.from_numpy(np.array(z[key])).to(dev)


def check_reset(scene, dev, fused):
  meta, z = load(scene)
  book_log = env_core.LogBook(meta["max_episode_length_s"], dev)
  n = 0
  for e in meta["reset"]:
    mask = _t(z, e["mask"], dev)
    fills = [(_t(z, f["pre"], dev), f["value"]) for f in e["fills"]]
    vectors = [_t(z, v["pre"], dev) for v in e["vectors"]]
    # the summed vectors ARE some of the filled buffers (episode sums, metrics): one tensor object for both roles, as in the environment
    by_name = {f["name"]: t for f, (t, _) in zip(e["fills"], fills)}
    vectors = [by_name.get(v["name"], t) for v, t in zip(e["vectors"], vectors)]
    book = env_core.ResetBookkeeping(fills, vectors, e["rkeys"], [tuple(k) for k in e["mkeys"]], e["tkeys"], fused=fused)
    out = book.sums(mask)
    log = book_log.publish(book.log_entries(out), mask)
    book.fill(mask)
    if dev != "cpu":
      torch.cuda.synchronize()
    for f, (t, _) in zip(e["fills"], fills):  # the buffers after the reference's own _reset_idx: bit for bit
      if f["name"] in ("qfrc_applied", "xfrc_applied", "ctrl"):
        continue  # (compared below: the reference's reset events may write mjData after the clear)
      if f["name"].endswith(".command_counter"):
        t = t + mask.to(t.dtype)  # CommandTerm.reset zeroes the counter, the resample that follows in the same call counts one (managers/command_manager.py:44-66)
      assert torch.equal(t.cpu(), torch.from_numpy(np.array(z[f["post"]]))), (scene, f["name"])
    for f, (t, _) in zip(e["fills"], fills):
      if f["name"] in ("qfrc_applied", "xfrc_applied"):
        assert bool((t[mask] == 0).all()), f["name"]  # EntityData.clear_state for the reset environments
    for key, ref in e["log"].items():  # the log the reference's managers left (Python floats): same keys, same numbers to float rounding
      if key.startswith(("Episode_Reward/", "Episode_Termination/")) or (key.startswith("Metrics/") and key in log):
        assert key in log, (scene, key)
        assert abs(float(log[key]) - ref) <= 2e-6 * (1.0 + abs(ref)), (scene, key, float(log[key]), ref)
        n += 1
  assert n >= 20, n
  return n


def check_reward(scene, dev, fused):
  meta, z = load(scene)
  for e in meta["reward"]:
    values = _t(z, e["values"], dev)
    k, nenv = values.shape
    sums = [t.clone() for t in _t(z, e["sums_pre"], dev)]
    reward_buf = torch.full((nenv,), 7.0, device=dev)  # (whatever was there: the accumulation starts from zero)
    step_reward = torch.full((nenv, e["nterms"]), 3.0, device=dev)
    idle = [i for i in range(e["nterms"]) if i not in e["columns"]]
    for i in idle:
      step_reward[:, i] = 0.0
    weights = torch.tensor(e["weights"], dtype=torch.float32, device=dev)
    if fused:
      from mjlab_amd import env_terms

      sums = [s.contiguous() for s in sums]
      ptrs = torch.tensor([s.data_ptr() for s in sums], dtype=torch.int64, device=dev)
      env_terms.reward_accumulate(values, weights, torch.tensor(e["columns"], dtype=torch.int32, device=dev), e["dt"], reward_buf, ptrs, step_reward)
      torch.cuda.synchronize()
    else:
      env_core.reward_accumulate(values, weights, e["columns"], e["dt"], reward_buf, sums, step_reward)
    assert torch.equal(reward_buf.cpu(), torch.from_numpy(np.array(z[e["reward_buf"]]))), scene  # RewardManager.compute's results, bit for bit
    assert torch.equal(torch.stack(sums).cpu(), torch.from_numpy(np.array(z[e["sums_post"]]))), scene
    ref_step = torch.from_numpy(np.array(z[e["step_reward"]]))
    if dev == "cpu":
      assert torch.equal(step_reward.cpu(), ref_step), scene
    else:
      # `value / dt`: torch divides a device tensor by a Python scalar as a multiplication by the reciprocal (the eager reference on the
      # GPU does the same, and the HIP launch follows IT -- tests/test_gpu_reference_env.py compares them bit for bit); the recording
      # is the CPU reference's true division: equal to 1 ulp
      assert bool(((step_reward.cpu() - ref_step).abs() <= 1.2e-7 * ref_step.abs()).all()), scene
  return len(meta["reward"])


def check_obs(scene, dev):
  meta, z = load(scene)
  for e in meta["obs"]:
    for g, rec in e["groups"].items():
      out = env_core.assemble_observation([_t(z, key, dev) for key in rec["terms"]], False, None, None, None)
      assert torch.equal(out.cpu(), torch.from_numpy(np.array(z[rec["out"]]))), (scene, g)
  return len(meta["obs"])


def check_velocity(scene, dev):
  meta, z = load(scene)
  changed = 0
  for e in meta["velocity"]:
    v = _t(z, e["pre"], dev).clone()
    before = v.clone()
    env_core.update_uniform_velocity(v, _t(z, e["heading_target"], dev), _t(z, e["heading_w"], dev), _t(z, e["is_heading_env"], dev), _t(z, e["is_standing_env"], dev),
                                     e["heading_command"], e["stiffness"], e["ang_vel_z"], wrap_to_pi)
    assert torch.equal(v.cpu(), torch.from_numpy(np.array(z[e["post"]])))
    changed += int((v != before).any())
  assert changed >= 1
  return len(meta["velocity"])


@pytest.mark.parametrize("scene", SCENES)
def test_core_reproduces_the_reference_on_the_cpu(scene):
  assert check_reset(scene, "cpu", False) >= 20
  assert check_reward(scene, "cpu", False) >= 6
  assert check_obs(scene, "cpu") >= 2
  if scene in VELOCITY_SCENES:
    assert check_velocity(scene, "cpu") >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("scene", SCENES)
@pytest.mark.parametrize("fused", (False, True), ids=("torch", "hip"))
def test_core_reproduces_the_reference_on_the_device(scene, fused):
  assert check_reset(scene, "cuda:0", fused) >= 20
  assert check_reward(scene, "cuda:0", fused) >= 6
  if not fused:
    assert check_obs(scene, "cuda:0") >= 2
    if scene in VELOCITY_SCENES:
      assert check_velocity(scene, "cuda:0") >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("scene", SCENES)
def test_core_captured_as_one_hipgraph_replays_the_reference(scene):
  """The same recorded truth through a CAPTURED graph (reference-free, so the driver's box runs it): the reset bookkeeping's launches
  (mjlab_masked_sums -> mjlab_log_finish -> mjlab_masked_fill_rows) and RewardManager.compute's accumulation (mjlab_reward_accumulate)
  are captured ONCE on persistent buffers into one hipGraph, as GraphedRlEnv captures them; every recorded step is then copied into
  those buffers and answered by a replay -- buffers and log bit for bit / to float rounding as in the eager checks above, and the log
  entries handed out stay the same tensors."""
  from mjlab_amd import env_terms

  dev = "cuda:0"
  meta, z = load(scene)
  e0, r0 = meta["reset"][0], meta["reward"][0]
  assert all([f["name"] for f in e["fills"]] == [f["name"] for f in e0["fills"]] and e["rkeys"] == e0["rkeys"] for e in meta["reset"])
  # persistent buffers (what the environment's managers own)
  mask = _t(z, e0["mask"], dev).clone()
  fills = [(_t(z, f["pre"], dev).clone(), f["value"]) for f in e0["fills"]]
  by_name = {f["name"]: t for f, (t, _) in zip(e0["fills"], fills)}
  vectors = [by_name[v["name"]] if v["name"] in by_name else _t(z, v["pre"], dev).clone() for v in e0["vectors"]]
  book = env_core.ResetBookkeeping(fills, vectors, e0["rkeys"], [tuple(k) for k in e0["mkeys"]], e0["tkeys"], fused=True)
  book_log = env_core.LogBook(meta["max_episode_length_s"], dev)
  values = _t(z, r0["values"], dev).clone()
  k, nenv = values.shape
  sums = [t.clone().contiguous() for t in _t(z, r0["sums_pre"], dev)]
  ptrs = torch.tensor([s.data_ptr() for s in sums], dtype=torch.int64, device=dev)
  reward_buf = torch.zeros(nenv, device=dev)
  step_reward = torch.zeros(nenv, r0["nterms"], device=dev)
  weights = torch.tensor(r0["weights"], dtype=torch.float32, device=dev)
  columns = torch.tensor(r0["columns"], dtype=torch.int32, device=dev)
  handed = {}

  def body():
    env_terms.reward_accumulate(values, weights, columns, r0["dt"], reward_buf, ptrs, step_reward)
    out = book.sums(mask)
    handed["log"] = book_log.publish(book.log_entries(out), mask, out[-1])
    book.fill(mask)

  stream = torch.cuda.Stream()
  stream.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(stream):
    body()  # (allocates the log vectors through the torch lines)
    body()  # (the launch form, its pointer table built outside the capture)
  torch.cuda.current_stream().wait_stream(stream)
  torch.cuda.synchronize()
  assert book_log._ptrs is not None
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph, stream=stream):
    body()
  first_log = {key: t.data_ptr() for key, t in handed["log"].items()}
  book_log.clear()

  n = 0
  rewards = meta["reward"]
  for i, e in enumerate(meta["reset"]):
    r = rewards[i % len(rewards)]
    assert r["columns"] == r0["columns"] and r["weights"] == r0["weights"]
    mask.copy_(_t(z, e["mask"], dev))
    for f, (t, _) in zip(e["fills"], fills):
      t.copy_(_t(z, f["pre"], dev))
    for v, t in zip(e["vectors"], vectors):
      if v["name"] not in by_name:
        t.copy_(_t(z, v["pre"], dev))
    values.copy_(_t(z, r["values"], dev))
    for s, t in zip(sums, _t(z, r["sums_pre"], dev)):
      s.copy_(t)
    reward_buf.fill_(7.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(reward_buf.cpu(), torch.from_numpy(np.array(z[r["reward_buf"]]))), scene
    assert torch.equal(torch.stack(sums).cpu(), torch.from_numpy(np.array(z[r["sums_post"]]))), scene
    checked = 0
    for j, (f, (t, _)) in enumerate(zip(e["fills"], fills)):
      if f["name"] in ("qfrc_applied", "xfrc_applied"):
        assert bool((t[mask] == 0).all()), f["name"]
        continue
      if f["name"] == "ctrl" or f["value"] != e0["fills"][j]["value"]:
        continue  # (the event manager's step counter: a fill value that changes from step to step is a constant of the captured launch)
      if f["name"].endswith(".command_counter"):
        t = t + mask.to(t.dtype)
      assert torch.equal(t.cpu(), torch.from_numpy(np.array(z[f["post"]]))), (scene, f["name"])
      checked += 1
    assert checked >= len(fills) - 5
    log = handed["log"]
    assert {key: t.data_ptr() for key, t in log.items()} == first_log
    for key, ref in e["log"].items():
      if key.startswith(("Episode_Reward/", "Episode_Termination/")) or (key.startswith("Metrics/") and key in log):
        assert abs(float(log[key]) - ref) <= 2e-6 * (1.0 + abs(ref)), (scene, key, float(log[key]), ref)
        n += 1
  assert n >= 20, n
  # a step without resets leaves the log where the last reset step put it (the where(count > 0) of the launch, inside the graph)
  before = {key: float(t) for key, t in handed["log"].items()}
  mask.zero_()
  graph.replay()
  torch.cuda.synchronize()
  assert {key: float(t) for key, t in handed["log"].items()} == before
