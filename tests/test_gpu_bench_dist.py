"""bench.py's multi-rank path end to end on ONE GPU: two ranks share the device over the gloo backend
(MJLAB_DIST_BACKEND=gloo; RCCL refuses two ranks on one device), launched exactly as the driver launches
the scaling run.  Checks the contract line: n_gpus, the exchange on the timed path (action scatter +
observation gather to the learner), per-rank diagnostics."""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_bench_two_ranks_on_one_gpu():
  env = dict(os.environ, MJLAB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
  port = 29600 + os.getpid() % 300
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--envs-per-gpu", "256", "--no-full-env"]  # fmt: skip
  p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert p.returncode == 0, p.stderr[-2000:]
  line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
  d = json.loads(line)
  assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 512 and d["scaling"] == "weak"
  assert d["value"] > 0 and d["steps"] == 6 and d["warmup"] == 2
  assert "action scatter" in d["config"]["parallelism"] and "gather" in d["config"]["parallelism"]
  assert len(d["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d["per_rank_ms_per_step"])
  assert d["exchange_ms_per_step"] is not None and d["exchange_ms_per_step"] > 0
  assert d["value_with_gather"] == d["value"]  # N > 1: the exchange is inside the timed region
  assert d["cpu_baseline"] is None  # reported at N = 1 only
  # the second view: two half batches per rank with interleaved learner round trips (gloo: the schedule, not the overlap)
  pl = d["pipelined"]
  assert "error" not in pl, pl
  assert pl["value"] > 0 and 0.0 <= pl["exchange_overlap_frac"] <= 1.0 and pl["ms_per_step_halves_compute_only"] > 0


def test_bench_eight_ranks_on_one_gpu():
  """The driver's 8-GPU launch line with eight ranks sharing the one GPU over gloo (256 envs each): the JSON contract of the N = 8 run --
  nothing about its speed -- so that the first run on an 8-GPU node cannot fail on plumbing."""
  env = dict(os.environ, MJLAB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
  port = 29300 + os.getpid() % 250
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2", "--envs-per-gpu", "256", "--settle", "20"]  # fmt: skip
  p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
  assert p.returncode == 0, p.stderr[-2000:]
  lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
  assert len(lines) == 1  # rank 0 alone prints
  d = json.loads(lines[0])
  assert d["n_gpus"] == 8 and d["config"]["global_envs"] == 2048 and d["scaling"] == "weak" and d["steps"] == 5 and d["warmup"] == 2
  assert d["value"] > 0 and d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
  assert len(d["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in d["per_rank_ms_per_step"])
  assert d["exchange_ms_per_step"] > 0 and d["allreduce_4_bytes_us"] > 0
  assert "x8" in d["config"]["parallelism"] and d["value_with_gather"] == d["value"]
  assert d["cpu_baseline"] is None and d["value_at_16384"] is None  # N = 1 legs
  assert "error" not in d["pipelined"], d["pipelined"]
  # the sharded full environment: measured where the reference's source is staged, else a note that says why not -- never an exception
  assert d["value_full_env_sharded"] is not None or "not measured" in d["value_full_env_sharded_note"], d["value_full_env_sharded_note"]
  assert d["roofline"]["kernel"].startswith("k_control_step") and d["roofline"]["frac"] > 0


def test_bench_exchange_over_rccl_with_one_rank():
  """The collectives of the N > 1 path (scatter of the actions, gather to the learner, max / all-gather of the timings,
  barrier) issued through RCCL itself -- backend "nccl" -- with a single rank, the only form a 1-GPU box allows."""
  env = dict(os.environ, MJLAB_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
  env.pop("MJLAB_DIST_BACKEND", None)
  port = 29900 + os.getpid() % 90
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--envs-per-gpu", "256", "--no-cpu-baseline", "--no-full-env"]  # fmt: skip
  p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert p.returncode == 0, p.stderr[-2000:]
  d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
  assert d["n_gpus"] == 1 and d["value"] > 0
  assert d["exchange_ms_per_step"] is not None and d["exchange_ms_per_step"] > 0
  assert d["allreduce_4_bytes_us"] > 0  # the sharded full environment's mid-step flag all-reduce, over RCCL
  assert "error" not in d["pipelined"], d["pipelined"]  # the side-stream exchange over RCCL, ordered by events
  out = ROOT / "gpurun_out"
  if out.is_dir():  # what a 1-GPU box can say about the exchange: recorded next to the run (copied to profiles/)
    (out / "exchange_one_rank_rccl.json").write_text(json.dumps({k: d[k] for k in ("value", "ms_per_step", "exchange_ms_per_step", "allreduce_4_bytes_us", "pipelined", "config")}, indent=1))
