"""Worker of tests/test_dist.py::test_sharded_tracking_environment_equals_the_single_process_batch (one process per rank, gloo, CPU).

Every rank builds BOTH the global batch -- the reference's own Mjlab-Tracking-Flat-Unitree-G1 environment with 2 n environments
behind GraphedRlEnv, a single process' view -- and its own shard of n environments behind ``GraphedRlEnv(env, shard=...)``.  The
shard starts from the global batch's slice; then both take the same 30 control steps (the learner's actions for all environments,
the same uniforms draw for draw: ``replicate_rng``), and after every step

  * this rank's observations / reward / dones / mjData equal the global batch's slice bit for bit,
  * the tracking task's failure statistics (``bin_failed_count``: all-reduced histogram) equal the global batch's bit for bit,
  * ``extras["log"]`` (all-reduced sums and counts) equals the global batch's to float rounding,
  * on the learner, the gathered rows equal the global batch's outputs for every environment."""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tools", ROOT / "tests"):
  sys.path.insert(0, str(p))


def slice_state(G, L, sl, NG):
  """Everything env.step reads, global batch -> this rank's shard (tests/_graphed_check.py::_sync with a slice)."""
  from mjlab_amd.graphed_env import _state_tensors

  from _graphed_check import _managers

  def cp(dst, src):
    dst.copy_(src[sl] if (src.dim() >= 1 and src.shape[0] == NG and dst.shape[0] != NG) else src)

  for k, t in G.sim._data.items():
    if t.shape[1:] == L.sim._data[k].shape[1:]:  # (the scene holds one site per environment: site arrays differ in width; derived, recomputed by the step)
      cp(L.sim._data[k], t)
  for k, t in G.sim._model_view.items():  # per-world model fields (domain randomisation)
    lt = L.sim._model_view[k]
    if t.dim() >= 1 and t.shape[0] == NG and t.stride(0) != 0 and t.shape[1:] == lt.shape[1:]:
      assert lt.stride(0) != 0, k
      lt.copy_(t[sl])
  for mg, ml in zip(_managers(G), _managers(L), strict=True):
    sa, sb = [], []
    _state_tensors(mg, None, set(), sa)
    _state_tensors(ml, None, set(), sb)
    assert [p for *_, p in sa] == [p for *_, p in sb]
    for (_, _, ta, _), (_, _, tb, _) in zip(sa, sb, strict=True):
      if ta.shape == tb.shape:
        tb.copy_(ta)
      else:
        assert ta.shape[0] == NG and tb.shape[0] == NG // WORLD, (ta.shape, tb.shape)
        tb.copy_(ta[sl])
  L.scene.env_origins.copy_(G.scene.env_origins[sl])  # (each process lays its environments out on a grid of its own: the shard takes the batch's places)
  L.episode_length_buf.copy_(G.episode_length_buf[sl])
  L._sim_step_counter, L.common_step_counter = G._sim_step_counter, G.common_step_counter


def main():
  global WORLD
  rank, WORLD, port, motion, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(WORLD), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
  import reference_env
  from _motion_fixture import write_full_motion
  from _oracle_simulation import OracleSimulation

  from mjlab_amd import dist as mdist
  from mjlab_amd.graphed_env import GraphedRlEnv

  n, steps = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (6, 30)
  NG = n * WORLD
  info = mdist.init_from_env(envs_per_rank=n, backend="gloo")
  if rank == 0:
    write_full_motion(motion)
  mdist.barrier()

  def make(num):
    def edit(cfg):
      cfg.commands.motion.motion_file = motion
      for group in ("policy", "critic"):
        getattr(cfg.observations, group).enable_corruption = False
      cfg.events.push_robot.interval_range_s = (0.1, 0.4)

    return reference_env.make_env("Mjlab-Tracking-Flat-Unitree-G1", num_envs=num, device="cpu", sim_cls=OracleSimulation, seed=7, cfg_edit=edit)

  torch.manual_seed(0)
  G, L = make(NG), make(n)
  G.reset(); L.reset()
  Gg = GraphedRlEnv(G, capture=False, shard=mdist.ShardInfo(0, 1, 0, NG), replicate_rng=True)
  Lg = GraphedRlEnv(L, capture=False, shard=info, replicate_rng=True)
  sl = info.env_slice
  zero = torch.zeros((NG, 29))
  Gg.step(zero.clone())  # (MotionCommand creates two metric entries on its first _update_metrics: one step each)
  Lg.step_sharded(zero.clone() if rank == 0 else None)
  cg, cl = G.command_manager.get_term("motion"), L.command_manager.get_term("motion")
  total = cg.motion.time_step_total
  gen = torch.Generator().manual_seed(5)
  stats = {"resets": 0, "failed": 0, "ended": 0, "steps_with_global_failures_on_one_rank_only": 0}
  for k in range(steps):
    slice_state(G, L, sl, NG)
    if k % 7 == 3:  # some motions run out: a resample inside _update_command, without a reset
      cg.time_steps[:2] = total - 2
      cl.time_steps.copy_(cg.time_steps[sl])
    action = (torch.rand((NG, 29), generator=gen) * 2 - 1) * (3.0 if k % 3 == 0 else 0.3)  # (large actions: falls -> failures for the sampler)
    torch.manual_seed(1000 + k)
    og, rg, tg, og_to, _ = Gg.step(action.clone())
    torch.manual_seed(1000 + k)
    (ol, rl, tl, ol_to, _), gathered = Lg.step_sharded(action.clone() if rank == 0 else None)
    # ---- this rank's slice of the global batch, bit for bit
    assert torch.equal(tl, tg[sl]) and torch.equal(ol_to, og_to[sl]) and torch.equal(rl, rg[sl]), k
    for grp in og:
      assert torch.equal(ol[grp], og[grp][sl]), (k, grp)
    for f in ("qpos", "qvel", "ctrl", "xpos"):
      assert torch.equal(getattr(L.sim.data, f), getattr(G.sim.data, f)[sl]), (k, f)
    assert torch.equal(cl.time_steps, cg.time_steps[sl]), k
    # ---- the global statistics: identical on every rank to the single process'
    assert torch.equal(cl.bin_failed_count, cg.bin_failed_count), (k, (cl.bin_failed_count - cg.bin_failed_count).abs().max())
    assert not bool(cl._current_bin_failed.any()) and not bool(cg._current_bin_failed.any())
    lg, ll = G.extras["log"], L.extras["log"]
    assert set(lg) == set(ll), (k, set(lg) ^ set(ll))
    for key, v in lg.items():
      a, b = float(v.float().mean()), float(ll[key].float().mean())
      assert abs(a - b) <= 1e-5 * (1.0 + abs(a)), (k, key, a, b)
    # ---- the learner holds every environment's rows
    assert (gathered is None) == (rank != 0)
    if rank == 0:
      gobs, grew, gterm, gto = gathered
      assert torch.equal(grew, rg) and torch.equal(gterm, tg) and torch.equal(gto, og_to), k
      for grp in og:
        assert torch.equal(gobs[grp], og[grp]), (k, grp)
    reset = tg | og_to
    stats["resets"] += int(reset.sum()); stats["failed"] += int(tg.sum()); stats["ended"] += int(((cg.time_steps) >= total - 1).sum())
    per_rank = [int(tg[r * n : (r + 1) * n].sum()) for r in range(WORLD)]
    stats["steps_with_global_failures_on_one_rank_only"] += int(sum(1 for c in per_rank if c > 0) == 1)
  stats["bin_failed_mass"] = float(cg.bin_failed_count.sum())
  if rank == 0:
    Path(out).write_text(json.dumps(stats))
  mdist.barrier()
  import torch.distributed as tdist

  tdist.destroy_process_group()


if __name__ == "__main__":
  main()
