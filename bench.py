"""Benchmark of the physics hot path: env-steps/s, Unitree G1 velocity-flat, 4096 envs per GPU.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one env (control) step of the reference's hot loop
(src/mjlab/envs/manager_based_rl_env.py:106-147) restricted to its physics: random action
-> 4 x [write ctrl, Simulation.step()] -> termination check + masked reset ->
Simulation.forward().  Inputs are resident in HBM before the timed region.  One process per
GPU; worlds are sharded across ranks (weak scaling, 4096 per GPU); for N > 1 the per-step
observation rows are all-gathered over RCCL like the north_star's obs/reward gather.

Prints ONE JSON line on rank 0 (contract in the task statement); extra objects:
  roofline     -- dominant kernel (Newton solve + integrate) vs the HBM roofline, using
                  SURVEY.md section 8(d)'s algorithmic bytes per world per physics step;
  cpu_baseline -- the CPU oracle (a port, NOT upstream mj_step) timed on this host's cores
                  on a bounded sample of the same workload.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from mjlab_amd import dist as mdist  # noqa: E402
from mjlab_amd import native, robots  # noqa: E402
from mjlab_amd.rollout import TRACKING_TASK_EVENTS, VELOCITY_TASK_EVENTS, PhysicsRollout, g1_action_scale, go1_action_scale, synthetic_motion  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402

# compulsory HBM traffic of the public mjData contract, fp32
# (rough: the G1 contract without the plane geom's pose -- terrain geoms are static, their poses are
# written once at construction and are not per-step traffic)
ALGO_BYTES_PER_WORLD_STEP = {"g1_velocity_flat": 10156, "g1_tracking_flat": 10716, "go1_velocity_flat": 5672, "g1_velocity_rough": 10108,
                             "go1_velocity_rough": 5624}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3  # vector fp32: 256 CUs x 4 SIMDs x 64 lanes x 2 flop / 2 cycles x 2.4 GHz (same guide)
# a forward() pass reads the same inputs and writes every derived field but not the integrated state (qpos, qvel, time)
FORWARD_LESS_BYTES = {"g1": (36 + 35 + 1) * 4, "go1": (19 + 18 + 1) * 4}


def cpu_baseline(scene: str, seed: int) -> dict:
  """Time the CPU oracle (test infrastructure, used here only as the reported baseline)."""
  from oracle.oracle import OracleSim

  model = robots.load_model(scene)
  cores = os.cpu_count() or 1
  nworld, env_steps = 64 * cores, 10
  ora = OracleSim(model, nworld, njmax=300, precision="f64")
  rng = np.random.default_rng(seed)
  jn = model.actuator_trnid[:, 0]
  default = model.key_qpos[0][model.jnt_qposadr[jn]]
  scale = g1_action_scale(model) if scene.startswith("g1") else go1_action_scale(model)
  best = None
  for nthread in sorted({cores, max(1, cores // 2)}, reverse=True):  # SMT siblings do not always help
    ora.reset(key=0)
    if hasattr(model, "terrain_origins"):  # spread the worlds over the terrain tiles
      o = np.asarray(model.terrain_origins)
      ora.qpos[:, :3] += o[rng.integers(0, min(6, o.shape[0]), nworld), np.arange(nworld) * o.shape[1] // nworld]
    ora.forward(nthread=nthread)
    t0 = time.perf_counter()
    for _ in range(env_steps):
      ora.ctrl[:] = default + scale * rng.uniform(-1, 1, size=(nworld, model.nu))
      ora.step(4, nthread=nthread)
      ora.forward(nthread=nthread)
    dt = time.perf_counter() - t0
    if best is None or nworld * env_steps / dt > best[0]:
      best = (nworld * env_steps / dt, nthread)
  return {
    "value": best[0],
    "unit": "env-steps/s",
    "cores": best[1],
    "kind": "port",
    "sample": f"{nworld} worlds x {env_steps} env-steps (4 substeps + 1 forward each), fp64 C oracle, {best[1]} pthreads "
    f"(best of {cores} and {max(1, cores // 2)}); CPU restatement, not upstream mj_step",
  }


def sharded_full_env(args, info, dev, model, tracking):
  """env-steps/s of the reference's own environment, one GraphedRlEnv per rank, with the learner exchange on the timed path
  (GraphedRlEnv.step_sharded).  Returns (value over all ranks, note); (None, reason) where the reference's source is absent."""
  import torch

  from mjlab_amd import dist as mdist

  sys.path.insert(0, str(ROOT / "tools"))
  import reference_env

  task = {"g1_velocity_flat": "Mjlab-Velocity-Flat-Unitree-G1", "go1_velocity_flat": "Mjlab-Velocity-Flat-Unitree-Go1",
          "g1_velocity_rough": "Mjlab-Velocity-Rough-Unitree-G1", "go1_velocity_rough": "Mjlab-Velocity-Rough-Unitree-Go1",
          "g1_tracking_flat": "Mjlab-Tracking-Flat-Unitree-G1"}.get(args.scene)
  have = torch.tensor([0.0 if (task is None or reference_env.locate_reference() is None) else 1.0])
  if mdist.max_over_ranks(1.0 - float(have), dev) > 0.0:  # (agreed on by every rank: nobody waits in a collective for a rank that skipped)
    return None, "not measured: the reference's source is not on every rank's machine (tools/stage_reference.sh stages it)"
  from mjlab_amd.graphed_env import GraphedRlEnv

  cfg_edit = None
  if tracking:
    import tempfile

    from mjlab_amd.rollout import write_motion_npz

    motion_path = str(Path(tempfile.mkdtemp()) / "motion.npz")
    write_motion_npz(motion_path, model, dev)

    def cfg_edit(cfg, _p=motion_path):
      cfg.commands.motion.motion_file = _p

  # the set-up that can fail on ONE rank (construction, capture, memory) is agreed on before anybody enters the timed collectives: a rank that
  # raised here used to skip ahead while the others waited for it in the next barrier -- now every rank reports "failed"
  err = None
  try:
    env = reference_env.make_env(task, num_envs=args.envs_per_gpu, device=dev, seed=mdist.seed_for_rank(args.seed, info), cfg_edit=cfg_edit)
    env.reset()
    genv = GraphedRlEnv(env, shard=info, capture=False)
  except Exception as e:  # noqa: BLE001
    err = f"{type(e).__name__}: {e}"
  if mdist.max_over_ranks(0.0 if err is None else 1.0, dev) > 0.0:
    return None, f"failed during set-up on {'this' if err else 'another'} rank" + (f": {err}" if err else "")
  genv.capture()  # (its warm-up steps hold the step's collectives: every rank is here)
  na = sum(env.action_manager.action_term_dim)
  gen = torch.Generator(device=dev)
  gen.manual_seed(args.seed)
  n = max(50, min(args.steps, 200))
  for k in range(20 + n):
    if k == 20:
      mdist.barrier()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
    a_all = 2.0 * torch.rand((info.global_envs, na), device=dev, generator=gen) - 1.0 if info.rank == 0 else None
    genv.step_sharded(a_all)
  torch.cuda.synchronize()
  mdist.barrier()
  t = mdist.max_over_ranks(time.perf_counter() - t0, dev)
  return info.global_envs * n / t, (f"{task}: GraphedRlEnv(env, shard) per rank x{info.world_size}, {n} timed steps after 20; per step: action scatter, the captured control step "
                                    f"({'two graphs around the all-reduce of the reset flag' if genv.graph_b is not None else 'one graph'}), all-reduce of the log sums"
                                    f"{' and the failure histogram' if tracking else ''}, gather of the observation groups + reward + dones to rank 0")


def main() -> None:
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=30)
  ap.add_argument("--envs-per-gpu", type=int, default=4096)
  ap.add_argument("--scene", default="g1_velocity_flat")
  ap.add_argument("--no-gather", action="store_true")
  ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly (no hipGraph at all)")
  ap.add_argument("--no-step-graph", action="store_true",
                  help="replay Simulation's per-call step/forward graphs like the reference (sim/sim.py:185,193) "
                  "instead of one hipGraph per control step")
  ap.add_argument("--torch-reset", action="store_true",
                  help="termination test + reset as a chain of torch ops (the reference's style) instead of the "
                  "fused mjlab_masked_reset launch")
  ap.add_argument("--no-fold", action="store_true",
                  help="recompute the position / collision / constraint stages in the step() that follows a "
                  "forward() even where qpos and qvel did not change (the reference's behaviour; results are bit-identical)")
  ap.add_argument("--masked-forward", action="store_true",
                  help="extension: forward() only on reset worlds (default: all worlds, like the reference)")
  ap.add_argument("--no-task-events", action="store_true",
                  help="leave out the velocity task's events (per-env foot friction, velocity pushes, 70 degree orientation "
                  "termination): round-1 workload (root-height termination, shared model)")
  ap.add_argument("--fuse", choices=["stage", "presolve", "step"], default="step",
                  help="launch structure: one kernel per stage / the four pre-solve stages fused / a whole substep per kernel")
  ap.add_argument("--substeps-per-call", type=int, default=4,
                  help="physics steps per Simulation.step() call: 1 = the reference's call pattern (ctrl write + step(), 4 times), "
                  "4 = one step(nsubstep=4) per control step (with --fuse step: one kernel launch for the 4 substeps)")
  ap.add_argument("--no-control-kernel", action="store_true",
                  help="separate library calls per phase of a control step (step, masked reset, forward, push) instead of one "
                  "mjlab_control_step launch (bit-identical results)")
  ap.add_argument("--balance-every", type=int, default=0,
                  help="re-deal the worlds over the SIMDs by expected cost every this many control steps (0 = never; "
                  "control kernel only; results unchanged)")
  ap.add_argument("--readback", action="store_true",
                  help="refresh EntityData's derived quantities (body / root poses and velocities, projected gravity, joint state) "
                  "in the control kernel's epilogue (mjlab_control_t.readback_on; control kernel only)")
  ap.add_argument("--no-pipeline", action="store_true",
                  help="N > 1: skip the second view in which each rank's worlds are two half batches whose learner round trips are "
                  "interleaved (mjlab_amd.dist.pingpong_steps; key `pipelined`).  `value` is always the plain synchronous exchange")
  ap.add_argument("--cone", choices=["pyramidal", "elliptic"], default="pyramidal",
                  help="friction cone of the compiled model (the tasks configure pyramidal: the headline workload; elliptic = the cone variants of the kernels, an experiment line)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-latency-bound", action="store_true", help="skip the one-wave-per-SIMD launch that measures roofline.latency")
  ap.add_argument("--no-big-batch", action="store_true", help="skip the 16 384-world leg (value_at_16384)")
  ap.add_argument("--no-full-env", action="store_true",
                  help="skip value_full_env (the reference's own ManagerBasedRlEnv of the same task stepped over this Simulation; "
                  "only measured where the reference source is reachable: MJLAB_REFERENCE_SRC / gpurun_ref, tools/stage_reference.sh)")
  ap.add_argument("--settle", type=int, default=200, help="untimed control steps of set-up before the warm-up steps (start-up transient of the rollout)")
  ap.add_argument("--seed", type=int, default=42)
  ap.add_argument("--exact-ls", action="store_true",
                  help="MuJoCo's exact iterative line search (SimulationCfg.ls_parallel=False) instead of the reference's setting, the grid search")
  args = ap.parse_args()

  info = mdist.init_from_env(args.envs_per_gpu)
  if info.world_size != args.gpus:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={info.world_size}; launch with torch.distributed.run")
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
  native.lib()  # fail loudly if the HIP extension is missing
  dev = f"cuda:{mdist.device_index(info)}"
  torch.cuda.set_device(mdist.device_index(info))

  model = robots.load_model(args.scene)
  if args.cone == "elliptic":
    import copy

    from mjlab_amd.mjcf import CONE_ELLIPTIC

    model = copy.deepcopy(model)
    model.opt.cone = CONE_ELLIPTIC
  sim = Simulation(args.envs_per_gpu, SimulationCfg(njmax=int(os.environ.get("MJLAB_BENCH_NJMAX", 300)), use_graph=not args.no_graph, fold_forward=not args.no_fold, fuse=args.fuse, ls_parallel=not args.exact_ls), model, dev)
  scale = g1_action_scale(model) if args.scene.startswith("g1") else go1_action_scale(model)
  robot = "g1" if args.scene.startswith("g1") else "go1"
  tracking = "tracking" in args.scene
  events = {} if args.no_task_events else dict((TRACKING_TASK_EVENTS if tracking else VELOCITY_TASK_EVENTS)[robot])
  if "motion_reset" in events:  # BASELINE config 4: resets to a random phase of a (synthetic) motion, the task's own terminations and events
    events["motion"] = synthetic_motion(model)
  roll = PhysicsRollout(sim, action_scale=scale, decimation=4, seed=mdist.seed_for_rank(args.seed, info),
                        masked_forward=args.masked_forward, fused_reset=not args.torch_reset,
                        min_height=-1.0e9 if "motion" in events else (0.3 if robot == "g1" else 0.15), substeps_per_call=args.substeps_per_call,
                        control_kernel=not args.no_control_kernel and args.fuse == "step" and not args.torch_reset, **events)
  if args.readback and roll.control_kernel:
    from mjlab_amd.entity_data import EntityReadback

    roll.readback = EntityReadback(sim)
  step_graph = not args.no_graph and not args.no_step_graph
  if step_graph:
    roll.capture_graph()
  exchange = (info.world_size > 1 or bool(os.environ.get("MJLAB_DIST_FORCE"))) and not args.no_gather
  nu = model.nu
  learner_gen = torch.Generator(device=dev)
  learner_gen.manual_seed(args.seed + 1000)

  step_no = [0]

  def env_step(with_rows: bool = False) -> None:
    """One control step.  N > 1: the learner on rank 0 decides the actions of ALL envs, every rank
    receives its slice (scatter over RCCL), steps its shard, and the [obs | ...] rows travel back to
    the learner (gather) -- the exchange SURVEY.md section 8e pairs with env sharding."""
    if exchange:
      a_all = torch.rand((info.global_envs, nu), device=dev, generator=learner_gen) * 2 - 1 if info.rank == 0 else None
      action = mdist.scatter_actions(info, a_all, nu, dev)
    else:
      action = roll.random_action(out=roll.action_buffer)
    if args.balance_every and roll.control_kernel:
      if step_no[0] % args.balance_every == 0:
        roll.balance_worlds()
      step_no[0] += 1
    roll.step(action)
    if exchange:
      mdist.gather_rollout(info, roll.observation_rows())
    elif with_rows:
      roll.observation_rows()

  # Part of the set-up, before the W warm-up steps of the contract: the rollout leaves its start-up transient (every robot
  # standing in the keyframe pose, no resets yet, first replays of the hipGraph, first refresh of the wave-priority
  # classes at control step 16) so that a short --warmup measures the same steady state as a long one.  Measured with
  # --steps 20 --warmup 2 on one box: 3.38 M (no settle), 3.66 M (100), 3.52 M (200), 3.46 M (400), 3.39 M (800) against
  # 3.56-3.57 M for every 200-step chunk of a 1000-step run: a 20-step window carries +-4 % of phase noise (pushes, reset
  # bursts); 200 is the default because a short window then does not read higher than the long-run mean.
  for _ in range(args.settle):
    env_step()
  for _ in range(args.warmup):
    env_step()
  mdist.barrier()
  torch.cuda.synchronize()
  # chunk boundaries for the spread of the rate (events on the launch stream: the timed loop itself is unchanged)
  nchunk = 5 if args.steps >= 10 else 1
  marks = [torch.cuda.Event(enable_timing=True) for _ in range(nchunk + 1)]
  bounds = [round(i * args.steps / nchunk) for i in range(nchunk + 1)]
  t0 = time.perf_counter()
  marks[0].record()
  for i in range(args.steps):
    env_step()
    if i + 1 in bounds[1:]:
      marks[bounds.index(i + 1)].record()
  torch.cuda.synchronize()
  mdist.barrier()
  my_elapsed = time.perf_counter() - t0
  elapsed = mdist.max_over_ranks(my_elapsed, dev)
  rank_ms = mdist.all_rank_values(my_elapsed / args.steps * 1e3, dev)
  chunk_rates = [args.envs_per_gpu * info.world_size * (bounds[i + 1] - bounds[i]) / (marks[i].elapsed_time(marks[i + 1]) * 1e-3)
                 for i in range(nchunk)]
  # second view: the same K steps with the per-env observation rows assembled every step (N = 1: what a
  # learner on the same GPU would read; N > 1 the rows are already part of the exchange above)
  value_with_rows = None
  if not exchange:
    for _ in range(3):  # untimed: the first assembly of the rows allocates its output (torch.cat)
      env_step(with_rows=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
      env_step(with_rows=True)
    torch.cuda.synchronize()
    value_with_rows = args.envs_per_gpu * args.steps / (time.perf_counter() - t1)
  # the exchange alone (N > 1), so that a scaling curve can be read: scatter of actions + gather of rows
  comm_ms, allreduce_us = None, None
  if exchange:
    rows = roll.observation_rows()
    mdist.barrier()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(20):
      a_all = torch.rand((info.global_envs, nu), device=dev, generator=learner_gen) * 2 - 1 if info.rank == 0 else None
      mdist.scatter_actions(info, a_all, nu, dev)
      mdist.gather_rollout(info, rows)
    torch.cuda.synchronize()
    comm_ms = mdist.max_over_ranks((time.perf_counter() - t2) / 20 * 1e3, dev)
    # the sharded full environment's mid-step collective (GraphedRlEnv._exchange_any: "did ANY environment of the global batch reset?", one
    # float all-reduced between the step's two graphs): 200 back-to-back calls on the launch stream, one synchronisation -- its cost per step
    flag = torch.zeros((), device=dev)
    for _ in range(10):
      torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
    mdist.barrier()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    for _ in range(200):
      torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
    torch.cuda.synchronize()
    allreduce_us = mdist.max_over_ranks((time.perf_counter() - t3) / 200 * 1e6, dev)

  # ---- N > 1, second view: the rank's worlds as TWO HALF BATCHES whose learner round trips are interleaved -- while the rows of
  # one half travel to the learner and its next actions travel back (side stream, ordered by events), the other half steps
  # (mjlab_amd.dist.pingpong_steps; nobody acts on a stale observation).  Reported next to `value`, never instead of it.
  pipelined = None
  if exchange and not args.no_pipeline and args.envs_per_gpu % 2 == 0:
    try:
      nh = args.envs_per_gpu // 2
      halves, hrolls = [], []
      for h in range(2):
        hs = Simulation(nh, SimulationCfg(njmax=int(os.environ.get("MJLAB_BENCH_NJMAX", 300)), fold_forward=not args.no_fold, fuse=args.fuse, ls_parallel=not args.exact_ls), model, dev)
        ev_h = {k: v for k, v in events.items()}
        hr = PhysicsRollout(hs, action_scale=scale, decimation=4, seed=mdist.seed_for_rank(args.seed, info) + 7919 * (h + 1), fused_reset=True,
                            min_height=-1.0e9 if "motion" in events else (0.3 if robot == "g1" else 0.15), substeps_per_call=args.substeps_per_call,
                            control_kernel=roll.control_kernel, **ev_h)
        if step_graph:
          hr.capture_graph()
        hrolls.append(hr)
        halves.append((hr.step, hr.observation_rows))
      gens = [torch.Generator(device=dev) for _ in range(2)]
      for h, g in enumerate(gens):
        g.manual_seed(args.seed + 2000 + h)

      def learner(h, k, rows_all):
        return torch.rand((info.world_size * nh, nu), device=dev, generator=gens[h]) * 2 - 1

      def timed(overlap: bool, n: int) -> float:
        mdist.pingpong_steps(info, max(2, args.warmup // 2), halves, learner, nu, dev, overlap=overlap)
        mdist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        mdist.pingpong_steps(info, n, halves, learner, nu, dev, overlap=overlap)
        torch.cuda.synchronize()
        mdist.barrier()
        return mdist.max_over_ranks((time.perf_counter() - t) / n * 1e3, dev)

      for hr in hrolls:  # the same start-up transient as the main rollout
        for _ in range(min(args.settle, 100)):
          hr.step(hr.random_action(out=hr.action_buffer))
      t_seq, t_pipe = timed(False, args.steps), timed(True, args.steps)
      torch.cuda.synchronize()
      t0h = time.perf_counter()
      for _ in range(args.steps):  # compute only: the two halves back to back, no exchange
        for hr in hrolls:
          hr.step(hr.random_action(out=hr.action_buffer))
      torch.cuda.synchronize()
      t_comp = mdist.max_over_ranks((time.perf_counter() - t0h) / args.steps * 1e3, dev)
      # compute only, the two halves on two streams: what the pipeline's physics costs when the half launches share the chip
      # (round 4: pingpong_steps steps each half on a stream of its own).  The proof the overlap needs: close to ms_per_step
      hstreams = [torch.cuda.Stream(device=dev) for _ in hrolls]
      for st in hstreams:
        st.wait_stream(torch.cuda.current_stream(dev))
      torch.cuda.synchronize()
      t0c = time.perf_counter()
      for _ in range(args.steps):
        for hr, st in zip(hrolls, hstreams):
          with torch.cuda.stream(st):
            hr.step(hr.random_action(out=hr.action_buffer))
      torch.cuda.synchronize()
      t_conc = mdist.max_over_ranks((time.perf_counter() - t0c) / args.steps * 1e3, dev)
      exch = max(t_seq - t_comp, 1e-9)
      pipelined = {"value": args.envs_per_gpu * info.world_size / (t_pipe * 1e-3), "ms_per_step": t_pipe, "ms_per_step_same_halves_sequential_exchange": t_seq,
                   "ms_per_step_halves_compute_only": t_comp, "ms_per_step_halves_concurrent_compute_only": t_conc, "exchange_overlap_frac": float(min(1.0, max(0.0, (t_seq - t_pipe) / exch))),
                   "note": "two half batches per rank, each stepping on a stream of its own (the two half launches share the chip); the exchange of one half overlaps the physics of the other (side stream + events); "
                   "exchange_overlap_frac = (sequential - pipelined) / (sequential - compute only)"}
      del hrolls, halves
    except Exception as e:  # noqa: BLE001
      pipelined = {"error": f"{type(e).__name__}: {e}"}

  # ---- dominant-kernel timing with HIP events on the launch stream.  With --fuse step the dominant
  # kernel is k_substep<NVP, true>: `substeps_per_call` whole physics steps of every world per launch; it is
  # bracketed here launch by launch on the same rollout (states keep evolving, resets included).  The
  # per-stage figures come from a second pass that runs the stage kernels one by one (informational).
  solve_ms, stage_ms, dom_ms, dom_name, dom_sub, lat, big = None, {}, None, None, 1, None, None
  if info.rank == 0:
    reps = max(5, min(args.steps, 25))
    if roll.control_kernel:
      acc = 0.0
      g, roll._graph = roll._graph, None  # eager launches for the bracketed pass
      for _ in range(reps):
        a = roll.random_action()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rnd_warm = torch.rand((4,), device=dev, generator=roll.gen)  # noqa: F841  (keeps the RNG launch out of the bracket below)
        e0.record()
        roll._step_eager(a)
        e1.record()
        torch.cuda.synchronize()
        acc += e0.elapsed_time(e1)
      roll._graph = g
      # the bracket holds torch.rand (one small launch) + the control-step kernel; the kernel's own
      # duration is what rocprofv3 reports (profiles/<tag>/kernel_stats.csv)
      dom_ms, dom_sub = acc / reps, roll.decimation + 1
      dom_name = f"k_control_step{'_cone' if args.cone == 'elliptic' else ''}<{min(x for x in (8, 16, 20, 24, 32, 36, 40, 48, 64) if x >= model.nv)}>"
    elif args.fuse == "step":
      acc, nl = 0.0, 0
      sim_graph = sim.use_graph
      sim.use_graph = False
      for _ in range(reps):
        sim.data.ctrl[:] = roll.default_joint + roll.random_action() * roll.action_scale
        evs = []
        for _ in range(roll.decimation // roll.substeps_per_call):
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          sim.step(roll.substeps_per_call)
          e1.record()
          evs.append((e0, e1))
        torch.cuda.synchronize()
        acc += sum(e0.elapsed_time(e1) for e0, e1 in evs)
        nl += len(evs)
      sim.use_graph = sim_graph
      dom_ms, dom_sub = acc / nl, roll.substeps_per_call
      dom_name = f"k_substep<{min(x for x in (8, 16, 20, 24, 32, 36, 40, 48, 64) if x >= model.nv)}, true>"
    # ---- the bound of the regime the kernel is in.  One world per wave and 4 waves per SIMD at 4096 worlds: the
    # launch ends when its slowest wave does, and a wave's lifetime is set by its own dependent chain (LDS / DPP / global round trips,
    # the sequential pivots of the factorization), not by the SIMD's issue rate.  The chain is measured, not modelled: the SAME control
    # kernel on a quarter of the worlds -- one wave per SIMD, nothing to share the issue port with -- of the same task, seed and
    # steady state.  critical_path = that launch time; frac = critical_path / measured: the share of the full launch a wave needs when
    # it has its SIMD to itself.  What is left (1 - frac) is what four waves cost each other (issue contention, LDS / L2 queueing).
    lat = None
    if roll.control_kernel and not args.no_latency_bound and args.envs_per_gpu % 4 == 0 and args.envs_per_gpu >= 1024:
      try:
        nq = args.envs_per_gpu // 4
        qs = Simulation(nq, SimulationCfg(njmax=int(os.environ.get("MJLAB_BENCH_NJMAX", 300)), use_graph=False, fold_forward=not args.no_fold, fuse=args.fuse, ls_parallel=not args.exact_ls), model, dev)
        qr = PhysicsRollout(qs, action_scale=scale, decimation=4, seed=mdist.seed_for_rank(args.seed, info), fused_reset=True,
                            min_height=-1.0e9 if "motion" in events else (0.3 if robot == "g1" else 0.15), substeps_per_call=args.substeps_per_call,
                            control_kernel=True, **events)
        for _ in range(args.settle + args.warmup):
          qr._step_eager(qr.random_action())
        accq = 0.0
        for _ in range(reps):
          a = qr.random_action()
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          rnd_warm = torch.rand((4,), device=dev, generator=qr.gen)  # noqa: F841
          e0.record()
          qr._step_eager(a)
          e1.record()
          torch.cuda.synchronize()
          accq += e0.elapsed_time(e1)
        mhz = 2400.0  # MI355X peak engine clock (/opt/skills/guides/MI355X_MICROARCH.md); the cycle figures are ms x this
        lat = {"critical_path_ms": accq / reps, "measured_ms": dom_ms, "frac": (accq / reps) / dom_ms,
               "critical_path_cycles": accq / reps * 1e-3 * mhz * 1e6, "measured_cycles": dom_ms * 1e-3 * mhz * 1e6, "clock_mhz": mhz,
               "waves_per_simd": {"critical_path": nq / 1024.0, "measured": args.envs_per_gpu / 1024.0},
               "how": f"same kernel, task, seed and steady state on {nq} worlds (one wave per SIMD): a wave's dependent chain without issue contention"}
        del qs, qr
      except Exception as e:  # noqa: BLE001
        lat = {"error": f"{type(e).__name__}: {e}"}
    # ---- the same rollout on 16 384 worlds: four launches' worth of waves, so that a finished wave's slot goes to
    # the next world and the launch's tail (a fifth of the 4096-world launch) is paid once per four batches.  `value` stays the 4096 line.
    if roll.control_kernel and not args.no_big_batch and info.world_size == 1 and args.envs_per_gpu == 4096:
      try:
        nbig = 16384
        bs = Simulation(nbig, SimulationCfg(njmax=int(os.environ.get("MJLAB_BENCH_NJMAX", 300)), fold_forward=not args.no_fold, fuse=args.fuse, ls_parallel=not args.exact_ls), model, dev)
        br = PhysicsRollout(bs, action_scale=scale, decimation=4, seed=mdist.seed_for_rank(args.seed, info), fused_reset=True,
                            min_height=-1.0e9 if "motion" in events else (0.3 if robot == "g1" else 0.15), substeps_per_call=args.substeps_per_call,
                            control_kernel=True, **events)
        if step_graph:
          br.capture_graph()
        for _ in range(min(args.settle, 100) + 5):
          br.step(br.random_action(out=br.action_buffer))
        torch.cuda.synchronize()
        tb = time.perf_counter()
        nbs = 20
        for _ in range(nbs):
          br.step(br.random_action(out=br.action_buffer))
        torch.cuda.synchronize()
        big = {"value": nbig * nbs / (time.perf_counter() - tb), "num_envs": nbig, "steps": nbs, "settle_steps": min(args.settle, 100) + 5}
        big["ms_per_step"] = nbig / big["value"] * 1e3
        del bs, br
      except Exception as e:  # noqa: BLE001
        big = {"error": f"{type(e).__name__}: {e}"}
    stages = [("position", 1), ("collision", 2), ("velocity", 4), ("constraint", 8), ("solve_integrate", 48)]
    acc = {k: 0.0 for k, _ in stages}
    nlaunch = 0
    for _ in range(reps):
      sim.data.ctrl[:] = roll.default_joint + roll.random_action() * roll.action_scale
      for _ in range(roll.decimation):
        evs = []
        for name, bits in stages:
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          sim.forward_stages(bits)
          e1.record()
          evs.append((name, e0, e1))
        torch.cuda.synchronize()
        for name, e0, e1 in evs:
          acc[name] += e0.elapsed_time(e1)
        nlaunch += 1
    stage_ms = {k: v / nlaunch for k, v in acc.items()}
    solve_ms = stage_ms["solve_integrate"]
    if dom_ms is None:
      dom_ms, dom_name = solve_ms, "k_solve_integrate"

  # ---- N > 1: the FULL environment sharded (reference env per rank behind GraphedRlEnv(env, shard=info): the tracking histogram and
  # the reset logging all-reduced, actions scattered from / observation groups + reward + dones gathered to the learner); every
  # rank takes part, rank 0 reports.  Only where the reference's source is staged (like value_full_env).
  full_env_sharded, full_env_sharded_note = None, None
  if not args.no_full_env and (info.world_size > 1 or mdist._FORCE):
    try:
      full_env_sharded, full_env_sharded_note = sharded_full_env(args, info, dev, model, tracking)
    except Exception as e:  # noqa: BLE001
      full_env_sharded_note = f"failed: {type(e).__name__}: {e}"

  if info.rank == 0:
    n_env = args.envs_per_gpu * info.world_size
    value = n_env * args.steps / elapsed
    algo = ALGO_BYTES_PER_WORLD_STEP.get(args.scene)
    traffic, valu_busy, prof = None, None, {}
    tfile = ROOT / "profiles" / "traffic.json"
    key = "substep" if args.fuse == "step" else "solve_integrate"
    if tfile.exists():
      try:
        prof = json.loads(tfile.read_text()).get(args.scene, {})
        traffic = prof.get(key + "_bytes_per_launch")
        valu_busy = prof.get(key + "_valu_busy")
      except Exception:  # noqa: BLE001
        traffic = None
    roof = None
    if algo and dom_ms:
      # units of one launch of the dominant kernel: physics steps (SURVEY 8(d)'s unit) and, in the control kernel, the
      # forward() pass after the resets, counted apart with its own (smaller) contract
      nstep = dom_sub - 1 if roll.control_kernel else dom_sub
      nfwd = 1 if roll.control_kernel else 0
      fwd_bytes = algo - FORWARD_LESS_BYTES[robot]
      algo_launch = (nstep * algo + nfwd * fwd_bytes) * args.envs_per_gpu
      achieved = algo_launch / (dom_ms * 1e-3) / 1e9
      flops_launch = prof.get(key + "_flops_per_launch")
      roof = {
        # "latency": the regime the kernel is in (see `latency` below).  achieved / peak / frac stay the HBM figures the contract asks for
        # (algorithmic bytes over the launch time against 8 TB/s) -- ~2 % by construction, SURVEY.md 8(d)'s consistency warning
        "bound": "latency" if lat and "frac" in lat else "hbm",
        "latency": lat,
        "kernel": dom_name,
        "units": {"physics_steps_per_launch": nstep, "forward_passes_per_launch": nfwd, "worlds": args.envs_per_gpu,
                  "bytes_per_world_step": algo, "bytes_per_world_forward": fwd_bytes},
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "frac_physics_steps_only": nstep * algo * args.envs_per_gpu / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        # NOT measured in this run: read from the committed rocprofv3 counter passes of the profile named in traffic_source
        # (FETCH_SIZE x 2.0 + WRITE_SIZE x 1.0, factors measured on known byte counts: profiles/calibration.json)
        "traffic": traffic,
        "traffic_source": prof.get("source"),
        # the bound that matters here (the HBM fraction is ~2 % by construction): share of the chip's VALU issue cycles the
        # dominant kernel uses -- SQ_INSTS_VALU x the issue cycles of its static instruction mix (same committed profile)
        "valu_busy": valu_busy,
        "valu_busy_2cycle_lower_bound": prof.get(key + "_valu_busy_2cycle_lower_bound"),
        # issued fp32 operations (static mix x SQ_INSTS_VALU + 2048 x MFMA of the committed profile) over THIS
        # run's kernel time, against the vector fp32 peak
        "flops": None if not flops_launch else {"achieved": flops_launch / (dom_ms * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                "frac": flops_launch / (dom_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, "flops_per_launch": flops_launch},
        "algorithmic_bytes_per_launch": algo_launch,
        "kernel_ms": dom_ms,
        "stage_ms": stage_ms,
        "note": "latency-bound by construction (SURVEY.md 8d): 4 waves per SIMD, one world per wave; lower HBM traffic is better",
      }
    # ---- the reference's OWN environment of the same task (ManagerBasedRlEnv, Scene, Entity, managers, MDP terms: unmodified
    # reference code, ~150 small torch kernels per control step around the physics) stepped over this Simulation: the full
    # env-steps/s SURVEY 8(d) asks for next to the physics-only `value`.  Only where the reference source is reachable
    # (tools/reference_env.py: MJLAB_REFERENCE_SRC / gpurun_ref staged by tools/stage_reference.sh); the driver's box has none.
    full_env, full_env_note, full_env_graphed, full_env_graphed_note = None, None, None, None
    task = {"g1_velocity_flat": "Mjlab-Velocity-Flat-Unitree-G1", "go1_velocity_flat": "Mjlab-Velocity-Flat-Unitree-Go1",
            "g1_velocity_rough": "Mjlab-Velocity-Rough-Unitree-G1", "go1_velocity_rough": "Mjlab-Velocity-Rough-Unitree-Go1",
            "g1_tracking_flat": "Mjlab-Tracking-Flat-Unitree-G1"}.get(args.scene)
    if not args.no_full_env and info.world_size == 1 and task is not None:
      sys.path.insert(0, str(ROOT / "tools"))
      import reference_env

      if reference_env.locate_reference() is None:
        full_env_note = "not measured: the reference's source is not on this machine (tools/stage_reference.sh stages it for one gpurun call)"
      else:
        try:
          cfg_edit = None
          if tracking:  # the task needs a motion file (none ships with the reference): the synthetic 10 s motion of the physics line, every
            # body's pose from this package's forward kinematics (mjlab_amd.rollout.write_motion_npz)
            import tempfile

            from mjlab_amd.rollout import write_motion_npz

            motion_path = str(Path(tempfile.mkdtemp()) / "motion.npz")
            write_motion_npz(motion_path, model, dev)

            def cfg_edit(cfg, _p=motion_path):
              cfg.commands.motion.motion_file = _p

          env = reference_env.make_env(task, num_envs=args.envs_per_gpu, device=dev, seed=args.seed, cfg_edit=cfg_edit)
          na = sum(env.action_manager.action_term_dim)
          gen = torch.Generator(device=dev)
          gen.manual_seed(args.seed)
          env.reset()
          nfull = max(20, min(args.steps, 100))
          for k in range(20 + nfull):
            if k == 20:
              torch.cuda.synchronize()
              tf = time.perf_counter()
            env.step(2.0 * torch.rand((args.envs_per_gpu, na), device=dev, generator=gen) - 1.0)
          torch.cuda.synchronize()
          full_env = args.envs_per_gpu * nfull / (time.perf_counter() - tf)
          full_env_note = f"{task}: the reference's ManagerBasedRlEnv.step over mjlab_amd.Simulation, {nfull} timed steps after 20, random policy"
        except Exception as e:  # noqa: BLE001
          full_env_note = f"failed: {type(e).__name__}: {e}"
        # the same environment object with its whole control step captured into ONE hipGraph (mjlab_amd/graphed_env.py: the
        # reference's action / termination / reward / observation managers as they are, resets / command resampling / pushes mask based)
        if full_env is not None:
          try:
            from mjlab_amd.graphed_env import GraphedRlEnv

            genv = GraphedRlEnv(env)
            ngr = max(50, min(args.steps, 200))
            for k in range(20 + ngr):
              if k == 20:
                torch.cuda.synchronize()
                tg = time.perf_counter()
              genv.step(2.0 * torch.rand((args.envs_per_gpu, na), device=dev, generator=gen) - 1.0)
            torch.cuda.synchronize()
            full_env_graphed = args.envs_per_gpu * ngr / (time.perf_counter() - tg)
            full_env_graphed_note = f"the same environment, GraphedRlEnv.step (one hipGraph per control step), {ngr} timed steps after 20; x{full_env_graphed / full_env:.1f} of value_full_env"
          except Exception as e:  # noqa: BLE001
            full_env_graphed_note = f"failed: {type(e).__name__}: {e}"
    cpu = None
    if not args.no_cpu_baseline and info.world_size == 1:  # reported at N = 1 only
      try:
        cpu = cpu_baseline(args.scene, args.seed)
      except Exception as e:  # noqa: BLE001
        cpu = {"error": str(e)}
    out = {
      "metric": f"env-steps/sec at num_envs={args.envs_per_gpu} per GPU, {'Unitree-G1' if args.scene.startswith('g1') else 'Unitree-Go1'} {'rough (box terrain)' if args.scene.endswith('rough') else 'flat'} "
      "(physics hot path: 4 substeps + 1 forward per env-step)",
      "value": value,
      "unit": "env-steps/s",
      "n_gpus": info.world_size,
      "steps": args.steps,
      "warmup": args.warmup,
      "settle_steps": args.settle,
      "ms_per_step": elapsed / args.steps * 1e3,
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,
      "dtype": "f32",
      "data": "synthetic (random actions, " + ("resets to random phases of a synthetic 10 s motion (SURVEY 8d(4)) with the task's pose / velocity / joint noise" if "motion" in events else "keyframe resets")
      + ("" if args.no_task_events else ", per-env foot friction U(0.3,1.2), velocity pushes every U(1,3) s") + "; compiled model from the reference MJCF)",
      "config": {
        "workload": f"{args.scene}: {args.envs_per_gpu} envs/GPU, timestep 0.005, decimation 4, Newton 10 it / 20 ls "
        + ("(ls_parallel: mujoco_warp's grid search, the reference's setting)" if sim.ls_parallel else "(exact iterative line search; ls_parallel off)")
        + f", implicitfast, {args.cone}, njmax 300"
        + ("" if args.no_task_events else ("; task events: DR friction / torso com / joint zero offsets, 6-component pushes, motion-phase resets, anchor height / orientation terminations, 10 s episodes"
                                           if "motion" in events else "; task events: DR friction (per-env geom_friction), pushes, bad_orientation 70 deg termination")),
        "global_envs": n_env,
        "parallelism": f"env-sharded x{info.world_size}" + (" + RCCL action scatter and obs gather to the learner (rank 0) every control step" if exchange else ""),
        "graph": "one hipGraph per control step" if step_graph else ("per-call step/forward hipGraphs" if sim.use_graph else "none"),
        "launches": {"stage": "one kernel per stage (5 per substep)", "presolve": "pre-solve stages fused (2 per substep)", "step": "one kernel per substep"}[args.fuse]
        + (", whole control step (action, 4 substeps, reset, forward" + (", EntityData read-back" if roll.readback is not None else "")
           + ", push) in ONE launch (mjlab_control_step)" if roll.control_kernel
           else (f", {args.substeps_per_call} substeps per Simulation.step() call" if args.substeps_per_call > 1 else "")),
        "forward_after_reset": "reset worlds only (extension)" if args.masked_forward else "all worlds (reference behaviour)",
        "forward_fold": "off" if args.no_fold else "step() after forward() skips the stages that depend on qpos/qvel only where both are unchanged (bit-exact)",
        "termination_and_reset": "torch ops" if args.torch_reset else "one fused launch (mjlab_masked_reset), mask based, no host sync",
      },
      "world_physics_steps_per_s": value * roll.decimation,
      "value_with_gather": value if exchange else value_with_rows,
      # the same task on 16 384 worlds of this one GPU (20 control steps): the contention-free rate per world; `value` above is the 4096 line
      "value_at_16384": big.get("value") if big else None,
      "value_at_16384_detail": big,
      "value_full_env": full_env,
      "value_full_env_note": full_env_note,
      "value_full_env_graphed": full_env_graphed,
      "value_full_env_graphed_note": full_env_graphed_note,
      "value_full_env_sharded": full_env_sharded,
      "value_full_env_sharded_note": full_env_sharded_note,
      "std_over_5": float(np.std(chunk_rates)) if nchunk == 5 else None,
      "chunk_values": chunk_rates,
      "per_rank_ms_per_step": rank_ms,
      "exchange_ms_per_step": comm_ms,
      "allreduce_4_bytes_us": allreduce_us,  # N > 1 (or MJLAB_DIST_FORCE): the sharded full environment's mid-step flag all-reduce, per call
      "pipelined": pipelined,
      "roofline": roof,
      "cpu_baseline": cpu,
    }
    print(json.dumps(out))
  if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
  main()
