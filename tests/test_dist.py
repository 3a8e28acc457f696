"""Env-sharded multi-process path on CPU (gloo, world_size 2, 4 and 8 -- the sizes the driver's scaling run uses): gather/scatter
plumbing and rank-sharded physics (on the oracle) equal to the single-process run."""

import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _per_rank(world_size):
  return 4 if world_size == 2 else 2


def _worker(rank, world_size, port, out_dir):
  sys.path.insert(0, str(ROOT))
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  from mjlab_amd import dist as mdist
  from mjlab_amd import robots
  from oracle.oracle import OracleSim

  n = _per_rank(world_size)
  N = n * world_size
  info = mdist.init_from_env(envs_per_rank=n, backend="gloo")
  assert info.global_envs == N and info.env_slice == slice(rank * n, rank * n + n)
  model = robots.load_model("go1_velocity_flat")
  rng = np.random.default_rng(123)  # identical global initial state on every rank
  q = rng.normal(0, 0.05, (N, model.nq - 7))
  sim = OracleSim(model, n)
  sim.reset(key=0)
  sim.qpos[:, 7:] += q[info.env_slice]
  # learner on rank 0 decides actions for all envs; ranks receive their slice
  actions = torch.from_numpy(rng.uniform(-1, 1, (N, model.nu)).astype(np.float32)) if rank == 0 else None
  mine = mdist.scatter_actions(info, actions, model.nu, "cpu")
  jn = model.actuator_trnid[:, 0]
  sim.ctrl[:] = model.key_qpos[0][model.jnt_qposadr[jn]] + 0.25 * mine.numpy().astype(np.float64)
  sim.step(4)
  rows = torch.from_numpy(np.concatenate([sim.qpos, sim.qvel], axis=1))
  gathered = mdist.gather_rollout(info, rows)  # to the learner (rank 0) only
  assert (gathered is None) == (rank != 0)
  everywhere = mdist.gather_rollout(info, rows, to_all=True)
  assert everywhere.shape == (N, model.nq + model.nv)
  if rank == 0:
    assert torch.equal(gathered, everywhere)
  assert mdist.all_rank_values(float(rank), "cpu") == [float(r) for r in range(world_size)]
  t = mdist.max_over_ranks(float(rank + 1), "cpu")
  assert t == float(world_size)
  mdist.barrier()
  if rank == 0:
    np.save(Path(out_dir) / "gathered.npy", gathered.numpy())
    np.save(Path(out_dir) / "actions.npy", actions.numpy())
  dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 4, 8])
def test_rank_shards_equal_the_single_process(tmp_path, world_size):
  """world_size 4 and 8: the first 8-GPU run must not be the first time eight ranks exchange anything."""
  port = 29511 + os.getpid() % 200 + 7 * world_size
  mp.spawn(_worker, args=(world_size, port, str(tmp_path)), nprocs=world_size, join=True)
  sys.path.insert(0, str(ROOT))
  from mjlab_amd import robots
  from oracle.oracle import OracleSim

  N = _per_rank(world_size) * world_size
  model = robots.load_model("go1_velocity_flat")
  rng = np.random.default_rng(123)
  q = rng.normal(0, 0.05, (N, model.nq - 7))
  sim = OracleSim(model, N)
  sim.reset(key=0)
  sim.qpos[:, 7:] += q
  actions = np.load(tmp_path / "actions.npy")
  jn = model.actuator_trnid[:, 0]
  sim.ctrl[:] = model.key_qpos[0][model.jnt_qposadr[jn]] + 0.25 * actions.astype(np.float64)
  sim.step(4)
  expect = np.concatenate([sim.qpos, sim.qvel], axis=1)
  got = np.load(tmp_path / "gathered.npy")
  assert np.array_equal(got, expect)


def _pingpong_worker(rank, world_size, port, out_dir):
  sys.path.insert(0, str(ROOT))
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  from mjlab_amd import dist as mdist
  from mjlab_amd import robots
  from oracle.oracle import OracleSim

  info = mdist.init_from_env(envs_per_rank=4, backend="gloo")
  model = robots.load_model("go1_velocity_flat")
  jn = model.actuator_trnid[:, 0]
  default = model.key_qpos[0][model.jnt_qposadr[jn]]
  res = {}
  for overlap in (False, True):
    rng = np.random.default_rng(123)
    q = rng.normal(0, 0.05, (2, 4, model.nq - 7))  # [half][global world of the half]
    sims = []
    for h in range(2):
      s = OracleSim(model, 2)
      s.reset(key=0)
      s.qpos[:, 7:] += q[h, rank * 2 : rank * 2 + 2]
      sims.append(s)
    gens = [np.random.default_rng(1000 + h) for h in range(2)]
    log = []

    def learner(h, k, rows_all):  # rank 0 only: a policy that depends on what it last saw of THIS half
      a = gens[h].uniform(-1, 1, (4, model.nu)).astype(np.float32)
      if rows_all is not None:
        a += 0.1 * np.tanh(rows_all.numpy()[:, : model.nu]).astype(np.float32)
      log.append((h, k, a.copy()))
      return torch.from_numpy(a)

    def make(h):
      def step(a):
        sims[h].ctrl[:] = default + 0.25 * a.numpy().astype(np.float64)
        sims[h].step(4)

      return step, lambda: torch.from_numpy(np.concatenate([sims[h].qpos, sims[h].qvel], axis=1))

    out = mdist.pingpong_steps(info, 3, [make(0), make(1)], learner, model.nu, "cpu", overlap=overlap)
    res[overlap] = ([None if o is None else o.numpy().copy() for o in out], log)
  if rank == 0:
    for h in range(2):
      assert np.array_equal(res[False][0][h], res[True][0][h])
      np.save(Path(out_dir) / f"pp_rows_{h}.npy", res[True][0][h])
    for (h0, k0, a0), (h1, k1, a1) in zip(res[False][1], res[True][1]):
      assert (h0, k0) == (h1, k1) and np.array_equal(a0, a1)
    np.save(Path(out_dir) / "pp_actions.npy", np.stack([a for _, _, a in res[True][1]]))
    np.save(Path(out_dir) / "pp_order.npy", np.array([(h, k) for h, k, _ in res[True][1]]))
  else:
    assert all(o is None for o in res[True][0])
  mdist.barrier()
  dist.destroy_process_group()


def test_pingpong_half_batches_equal_the_sequential_exchange(tmp_path):
  """mjlab_amd.dist.pingpong_steps: two half batches per rank whose learner round trips are interleaved (the exchange of one half
  in flight while the other steps).  Two gloo ranks: the interleaved schedule gives the rows and the learner the actions of the
  strictly sequential one, bit for bit, and both equal a single process stepping the 8 worlds with those actions."""
  port = 29711 + os.getpid() % 200
  mp.spawn(_pingpong_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  sys.path.insert(0, str(ROOT))
  from mjlab_amd import robots
  from oracle.oracle import OracleSim

  model = robots.load_model("go1_velocity_flat")
  jn = model.actuator_trnid[:, 0]
  default = model.key_qpos[0][model.jnt_qposadr[jn]]
  rng = np.random.default_rng(123)
  q = rng.normal(0, 0.05, (2, 4, model.nq - 7))
  actions, order = np.load(tmp_path / "pp_actions.npy"), np.load(tmp_path / "pp_order.npy")
  for h in range(2):
    sim = OracleSim(model, 4)  # the half's 4 global worlds (2 per rank, rank order)
    sim.reset(key=0)
    sim.qpos[:, 7:] += q[h]
    for (hh, k), a in zip(order, actions):
      if hh == h:
        sim.ctrl[:] = default + 0.25 * a.astype(np.float64)
        sim.step(4)
    assert np.array_equal(np.load(tmp_path / f"pp_rows_{h}.npy"), np.concatenate([sim.qpos, sim.qvel], axis=1))


def test_single_rank_passthrough():
  from mjlab_amd import dist as mdist

  info = mdist.ShardInfo(0, 1, 0, 5)
  rows = torch.arange(10.0).view(5, 2)
  assert mdist.gather_rollout(info, rows) is rows
  assert mdist.seed_for_rank(42, mdist.ShardInfo(3, 8, 3, 5)) == 45


@pytest.mark.parametrize("world_size", [2, 4, 8])
def test_sharded_tracking_environment_equals_the_single_process_batch(tmp_path, world_size):
  """The FULL environment sharded: `world_size` gloo ranks, each with its slice of the reference's
  tracking task behind ``GraphedRlEnv(env, shard=...)`` over the oracle, against the single process' batch -- see
  tests/_sharded_env_worker.py for what is compared after each control step.  world_size 2: 6 environments per rank, 30 steps;
  4 and 8: 2 per rank, 40 steps."""
  import json
  import subprocess

  if not (ROOT.parent / "reference").exists():
    pytest.skip("needs the reference tree (the environment classes are the reference's own)")
  port = str(29911 + os.getpid() % 200 + 7 * world_size)
  motion, out = str(tmp_path / "motion.npz"), str(tmp_path / "stats.json")
  n, steps = (6, 30) if world_size == 2 else (2, 40)
  env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")  # (eight processes on eight cores)
  procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "_sharded_env_worker.py"), str(r), str(world_size), port, motion, out, str(n), str(steps)],
                            cwd=str(ROOT), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world_size)]
  outs = [p.communicate(timeout=1500) for p in procs]
  for p, (so, se) in zip(procs, outs):
    assert p.returncode == 0, se[-4000:]
  st = json.loads(Path(out).read_text())
  print(st)
  # the run must have exercised what the exchange is for: failures feeding the sampler, on one rank only in some steps
  few = 4 if world_size == 2 else 2
  assert st["failed"] >= few and st["resets"] >= few and st["ended"] >= few and st["bin_failed_mass"] > 0 and st["steps_with_global_failures_on_one_rank_only"] >= 1, st
