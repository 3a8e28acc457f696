"""Env-sharded multi-process path on CPU (gloo, world_size 2): gather/scatter plumbing and
rank-sharded physics (on the oracle) equal to the single-process run."""

import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world_size, port, out_dir):
  sys.path.insert(0, str(ROOT))
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  from mjlab_amd import dist as mdist
  from mjlab_amd import robots
  from oracle.oracle import OracleSim

  info = mdist.init_from_env(envs_per_rank=4, backend="gloo")
  assert info.global_envs == 8 and info.env_slice == slice(rank * 4, rank * 4 + 4)
  model = robots.load_model("go1_velocity_flat")
  rng = np.random.default_rng(123)  # identical global initial state on every rank
  q = rng.normal(0, 0.05, (8, model.nq - 7))
  sim = OracleSim(model, 4)
  sim.reset(key=0)
  sim.qpos[:, 7:] += q[info.env_slice]
  # learner on rank 0 decides actions for all envs; ranks receive their slice
  actions = torch.from_numpy(rng.uniform(-1, 1, (8, model.nu)).astype(np.float32)) if rank == 0 else None
  mine = mdist.scatter_actions(info, actions, model.nu, "cpu")
  jn = model.actuator_trnid[:, 0]
  sim.ctrl[:] = model.key_qpos[0][model.jnt_qposadr[jn]] + 0.25 * mine.numpy().astype(np.float64)
  sim.step(4)
  rows = torch.from_numpy(np.concatenate([sim.qpos, sim.qvel], axis=1))
  gathered = mdist.gather_rollout(info, rows)  # to the learner (rank 0) only
  assert (gathered is None) == (rank != 0)
  everywhere = mdist.gather_rollout(info, rows, to_all=True)
  assert everywhere.shape == (8, model.nq + model.nv)
  if rank == 0:
    assert torch.equal(gathered, everywhere)
  assert mdist.all_rank_values(float(rank), "cpu") == [0.0, 1.0]
  t = mdist.max_over_ranks(float(rank + 1), "cpu")
  assert t == float(world_size)
  mdist.barrier()
  if rank == 0:
    np.save(Path(out_dir) / "gathered.npy", gathered.numpy())
    np.save(Path(out_dir) / "actions.npy", actions.numpy())
  dist.destroy_process_group()


def test_two_rank_shard_equals_single_process(tmp_path):
  port = 29511 + os.getpid() % 200
  mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  sys.path.insert(0, str(ROOT))
  from mjlab_amd import robots
  from oracle.oracle import OracleSim

  model = robots.load_model("go1_velocity_flat")
  rng = np.random.default_rng(123)
  q = rng.normal(0, 0.05, (8, model.nq - 7))
  sim = OracleSim(model, 8)
  sim.reset(key=0)
  sim.qpos[:, 7:] += q
  actions = np.load(tmp_path / "actions.npy")
  jn = model.actuator_trnid[:, 0]
  sim.ctrl[:] = model.key_qpos[0][model.jnt_qposadr[jn]] + 0.25 * actions.astype(np.float64)
  sim.step(4)
  expect = np.concatenate([sim.qpos, sim.qvel], axis=1)
  got = np.load(tmp_path / "gathered.npy")
  assert np.array_equal(got, expect)


def test_single_rank_passthrough():
  from mjlab_amd import dist as mdist

  info = mdist.ShardInfo(0, 1, 0, 5)
  rows = torch.arange(10.0).view(5, 2)
  assert mdist.gather_rollout(info, rows) is rows
  assert mdist.seed_for_rank(42, mdist.ShardInfo(3, 8, 3, 5)) == 45
