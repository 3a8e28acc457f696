"""Env-sharded multi-GPU execution: one process per GPU, worlds partitioned across ranks.

The physics step has no cross-world term, so ranks never exchange data inside
``Simulation.step()``.  The only exchange the path has is the one BASELINE.json's
north_star names: per control step the per-env observation / reward / done rows are
all-gathered to the learner and the actions travel back.  The
reference itself is single-process (``scripts/train.py:29`` hard-codes ``cuda:0``), so
this module is new functionality, not a port.

Backend "nccl" is RCCL on ROCm; "gloo" runs the same logic on CPU tensors for tests.
"""

from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class ShardInfo:
  rank: int
  world_size: int
  local_rank: int
  envs_per_rank: int

  @property
  def global_envs(self) -> int:
    return self.envs_per_rank * self.world_size

  @property
  def env_slice(self) -> slice:
    return slice(self.rank * self.envs_per_rank, (self.rank + 1) * self.envs_per_rank)


def init_from_env(envs_per_rank: int, backend: str | None = None) -> ShardInfo:
  """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun)."""
  world_size = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  global _FORCE
  _FORCE = bool(os.environ.get("MJLAB_DIST_FORCE"))  # exercise the collectives with ONE rank (RCCL smoke on a 1-GPU box)
  if (world_size > 1 or _FORCE) and not dist.is_initialized():
    if backend is None:
      # MJLAB_DIST_BACKEND=gloo lets the multi-rank path be exercised on a box with fewer GPUs than
      # ranks (several ranks share a device; RCCL refuses that)
      backend = os.environ.get("MJLAB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
      torch.cuda.set_device(local_rank)
    elif torch.cuda.is_available():
      torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
    probe_collectives()
  return ShardInfo(rank, world_size, local_rank, envs_per_rank)


def probe_collectives() -> None:
  """Decide ONCE, and the same way on every rank, whether the backend has scatter / gather (an RCCL build may lack them): each
  is tried on a tiny tensor and the outcomes are agreed on with an all-reduce (MIN), so that no rank can end up in a fallback
  collective while its peers sit in the primary one.  gloo has both (host tensors)."""
  global _SCATTER_SUPPORTED, _GATHER_SUPPORTED, _PROBED
  _PROBED = True
  if not dist.is_initialized() or dist.get_backend() != "nccl":
    return
  dev = torch.device("cuda", torch.cuda.current_device())
  n, r = dist.get_world_size(), dist.get_rank()
  ok = torch.ones((2,), dtype=torch.int32, device=dev)
  try:
    out = torch.empty((1,), device=dev)
    dist.scatter(out, [torch.full((1,), float(k), device=dev) for k in range(n)] if r == 0 else None, src=0)
  except (RuntimeError, NotImplementedError):
    ok[0] = 0
  try:
    dist.gather(torch.zeros((1,), device=dev), [torch.empty((1,), device=dev) for _ in range(n)] if r == 0 else None, dst=0)
  except (RuntimeError, NotImplementedError):
    ok[1] = 0
  dist.all_reduce(ok, op=dist.ReduceOp.MIN)
  _SCATTER_SUPPORTED, _GATHER_SUPPORTED = bool(ok[0].item()), bool(ok[1].item())


def seed_for_rank(seed: int, info: ShardInfo) -> int:
  """Rank r uses seed + r."""
  return seed + info.rank


def device_index(info: ShardInfo) -> int:
  """CUDA device of this rank: local_rank, wrapped when ranks outnumber devices (gloo testing)."""
  n = torch.cuda.device_count()
  return info.local_rank % n if n else 0


_FORCE = False
_PROBED = False  # probe_collectives() has run in this process group (init_from_env does it; otherwise the first exchange does)
_SCATTER_SUPPORTED = True
_GATHER_SUPPORTED = True  # cleared when the backend turns out to have no gather (then: all-gather, rank dst keeps the result)
_GATHER_BUF: dict = {}  # receive buffers, reused across control steps (the result is valid until the next call)


def gather_rollout(info: ShardInfo, rows: torch.Tensor, dst: int = 0, to_all: bool = False, tag: int = 0) -> torch.Tensor | None:
  """Per-env rows ``(envs_per_rank, k)`` of every rank -> ``(global_envs, k)`` in rank order ON THE
  LEARNER (rank ``dst``); the other ranks get None.  ``to_all=True`` delivers it to every rank
  (all-gather: replicated learners).

  One fused buffer per control step ([obs | reward | done] columns) instead of one collective per
  tensor.  The gather is what north_star asks for ("all-gather back to the learner"): each rank
  sends its shard once, straight to the learner over its own xGMI link -- 1/world_size of the bytes
  an all-gather moves, and no ring through links that are per-pair anyway.
  """
  if info.world_size == 1 and not _FORCE:
    return rows
  if not _PROBED:  # a process group the caller initialised itself (torchrun user code): every rank reaches its first exchange together
    probe_collectives()
  rows = rows.contiguous()
  if rows.is_cuda and dist.get_backend() == "gloo":
    # gloo has no device-side gather: testing path (several ranks sharing one GPU), staged through the host
    out = gather_rollout(info, rows.cpu(), dst, to_all, tag)
    return None if out is None else out.to(rows.device)
  key = (rows.shape[1], rows.dtype, rows.device, tag)  # `tag`: callers with several exchanges in flight keep their results apart
  need = to_all or info.rank == dst
  out = _GATHER_BUF.get(key) if need else None
  if need and (out is None or out.shape[0] != info.global_envs):
    out = _GATHER_BUF[key] = torch.empty((info.global_envs, rows.shape[1]), dtype=rows.dtype, device=rows.device)
  if to_all or not _GATHER_SUPPORTED:
    if out is None:  # a backend without gather: every rank has to take part in the all-gather
      out = _GATHER_BUF[key] = torch.empty((info.global_envs, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(out, rows)
    return out if (to_all or info.rank == dst) else None
  dist.gather(rows, list(out.chunk(info.world_size)) if info.rank == dst else None, dst=dst)
  return out


def all_gather_rollout(info: ShardInfo, rows: torch.Tensor) -> torch.Tensor:
  """The rows of every rank on EVERY rank (replicated learners; round 1's exchange): ``gather_rollout(..., to_all=True)``."""
  out = gather_rollout(info, rows, to_all=True)
  assert out is not None
  return out


def scatter_actions(info: ShardInfo, actions_global: torch.Tensor | None, action_dim: int, device, src: int = 0) -> torch.Tensor:
  """Learner (rank ``src``) -> each rank's ``(envs_per_rank, action_dim)`` slice."""
  if info.world_size == 1 and not _FORCE:
    assert actions_global is not None
    return actions_global
  if not _PROBED:
    probe_collectives()
  out = torch.empty((info.envs_per_rank, action_dim), dtype=torch.float32, device=device)
  if dist.get_backend() == "nccl":
    # Scatter: every rank receives only its own slice (475 KB at 4096 x 29), sent by the learner over that rank's own xGMI
    # link -- 1 / world_size of what a broadcast of all the actions moves through every link.  A build of the backend
    # without scatter falls back to broadcast + slice (the same on every rank: the error is raised before anything is enqueued).
    if _SCATTER_SUPPORTED:  # decided once for all ranks by probe_collectives()
      chunks = list(actions_global.contiguous().chunk(info.world_size)) if info.rank == src else None
      dist.scatter(out, chunks, src=src)
      return out
    buf = actions_global if info.rank == src else torch.empty((info.global_envs, action_dim), dtype=torch.float32, device=device)
    dist.broadcast(buf, src=src)
    out.copy_(buf[info.env_slice])
  else:
    host = out.cpu() if out.is_cuda else out  # gloo scatters host tensors
    chunks = [c.cpu().contiguous() for c in actions_global.chunk(info.world_size)] if info.rank == src else None
    dist.scatter(host, chunks, src=src)
    if out.is_cuda:
      out.copy_(host)
  return out


def pingpong_steps(info: ShardInfo, nsteps: int, halves: list, learner_actions, action_dim: int, device, overlap: bool = True,
                   comm_stream=None) -> list:
  """`nsteps` control steps of a rank's worlds split into HALF BATCHES whose learner round trips are interleaved: while the rows
  of one half travel to the learner (rank 0), the learner decides that half's next actions and they travel back, the other
  half steps -- the learner-faithful way to hide the round trip (every half still acts on its own latest observation; nothing
  is stale).  `halves[h]` = (step_fn(actions) -> None, rows_fn() -> tensor (envs_per_half, k)); `learner_actions(h, k, rows_all)`
  is called on rank 0 only and returns the actions of ALL ranks' half h for control step k, (world_size * envs_per_half,
  action_dim), from the rows gathered after step k - 1 (None for k = 0).  With `overlap=False` the same dependency chain runs
  strictly in sequence (the reference for bit-equality: results are identical by construction, only the timing differs).
  On RCCL the exchange of a half is issued on `comm_stream` (default: a side stream), ordered against the physics by events, and
  (round 4) each half steps on a COMPUTE STREAM OF ITS OWN: a half batch is 2 waves per SIMD and the physics kernel is bound by
  each wave's own latency, so two half-batch launches one after the other take ~1.8 x a full launch (2048 worlds: 1.12 ms against
  1.25 ms for 4096, profiles/r03_v8/scenes.txt) -- the two launches have to share the chip at the same time for the pipeline to
  cost nothing; only then does hiding the exchange pay.  gloo (CPU tests) runs everything in issue order.  Returns the rows gathered after the last step per half (rank 0; None elsewhere)."""
  nh = len(halves)
  if info.envs_per_rank % nh:
    raise ValueError(f"pingpong_steps: envs_per_rank = {info.envs_per_rank} is not divisible by the {nh} part batches")
  sub = ShardInfo(info.rank, info.world_size, info.local_rank, info.envs_per_rank // nh)
  cuda = torch.cuda.is_available() and str(device).startswith("cuda") and dist.is_initialized() and dist.get_backend() == "nccl"
  main = torch.cuda.current_stream(device) if cuda else None
  side = (comm_stream or torch.cuda.Stream(device=device)) if (cuda and overlap) else main
  comp = [torch.cuda.Stream(device=device) for _ in range(nh)] if (cuda and overlap) else [main] * nh
  if cuda and overlap:
    for cs in comp:
      cs.wait_stream(main)  # whatever prepared the halves' state was enqueued on the caller's stream
  gathered: list = [None] * nh
  act: list = [None] * nh
  ready: list = [None] * nh  # event: the actions of half h have arrived (recorded on the exchange stream)

  def exchange(h: int, k: int, rows) -> None:
    """rows of half h after step k - 1 (None at k = 0) -> learner -> actions of half h for step k."""
    ctx = torch.cuda.stream(side) if cuda else _null()
    with ctx:
      if cuda and rows is not None:
        side.wait_event(rows[1])
      if rows is not None:
        gathered[h] = gather_rollout(sub, rows[0], tag=h)
      a_all = learner_actions(h, k, gathered[h]) if info.rank == 0 else None
      act[h] = scatter_actions(sub, a_all, action_dim, device)
      if cuda:
        ready[h] = torch.cuda.Event()
        ready[h].record(side)

  for h in range(nh):
    exchange(h, 0, None)
  for k in range(nsteps):
    for h in range(nh):
      step_fn, rows_fn = halves[h]
      with (torch.cuda.stream(comp[h]) if cuda else _null()):
        if cuda:
          comp[h].wait_event(ready[h])
          if act[h] is not None and act[h].is_cuda:
            act[h].record_stream(comp[h])  # allocated on the exchange stream, read by this half's launch
        step_fn(act[h])
        r = rows_fn()
        ev = None
        if cuda:
          ev = torch.cuda.Event()
          ev.record(comp[h])
          if side is not comp[h]:
            r.record_stream(side)  # allocated here, read by the gather on the exchange stream
      if k + 1 < nsteps:
        exchange(h, k + 1, (r, ev))
      else:  # the last rows still travel to the learner
        ctx = torch.cuda.stream(side) if cuda else _null()
        with ctx:
          if cuda:
            side.wait_event(ev)
          gathered[h] = gather_rollout(sub, r, tag=h)
  if cuda:
    main.wait_stream(side)
    for cs in comp:
      if cs is not main:
        main.wait_stream(cs)
  return gathered


class _null:
  def __enter__(self):
    return self

  def __exit__(self, *a):
    return False


def max_over_ranks(value: float, device) -> float:
  if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
    return value
  if dist.get_backend() == "gloo":
    device = "cpu"
  t = torch.tensor([value], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


def all_rank_values(value: float, device) -> list[float]:
  """The same scalar from every rank, in rank order (diagnostics: per-rank step times)."""
  if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
    return [value]
  if dist.get_backend() == "gloo":
    device = "cpu"
  out = torch.empty((dist.get_world_size(),), dtype=torch.float64, device=device)
  dist.all_gather_into_tensor(out, torch.tensor([value], dtype=torch.float64, device=device))
  return [float(v) for v in out.tolist()]


def barrier() -> None:
  if dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE):
    dist.barrier()
