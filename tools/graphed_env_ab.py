"""In-call A/B of GraphedRlEnv variants on one box (GPU box, reference staged): the same task captured with different options, timed
interleaved.   python tools/graphed_env_ab.py [num_envs] [steps] [task]"""
import sys
import tempfile
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import reference_env  # noqa: E402

from mjlab_amd.graphed_env import GraphedRlEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
task = sys.argv[3] if len(sys.argv) > 3 else "Mjlab-Velocity-Flat-Unitree-G1"
edit = None
if "Tracking" in task:
  from mjlab_amd import robots
  from mjlab_amd.rollout import write_motion_npz

  path = str(Path(tempfile.mkdtemp()) / "motion.npz")
  write_motion_npz(path, robots.load_model("g1_tracking_flat"), "cuda:0")

  def edit(cfg):
    cfg.commands.motion.motion_file = path

VARIANTS = {"torch restatements": dict(fused_terms=False), "fused terms (default)": dict(), "default without the relative-poses launch": dict(fused_relative_poses=False), "default without the motion-frame launch": dict(fused_motion_frame=False), "default without the metrics launch": dict(fused_motion_metrics=False),
            "no EntityData / term caches": dict(cache_entity_data=False), "forward() on the reset worlds only": dict(forward="reset_worlds"),
            "EntityData by the reference's own chains": dict(fused_entity_data=False)}
envs = {}
for name, kw in VARIANTS.items():
  if ("relative" in name or "motion-frame" in name or "metrics" in name) and "Tracking" not in task:
    continue
  env = reference_env.make_env(task, num_envs=n, device="cuda:0", cfg_edit=edit)
  env.reset()
  envs[name] = GraphedRlEnv(env, **kw)
a = torch.zeros((n, sum(env.action_manager.action_term_dim)), device="cuda:0")
gen = torch.Generator(device="cuda:0")
gen.manual_seed(1)
times = {k: [] for k in envs}
for rep in range(3):
  for name, g in envs.items():
    for _ in range(10):
      g.step(a)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
      g.step(2.0 * torch.rand(a.shape, device="cuda:0", generator=gen) - 1.0)
    torch.cuda.synchronize()
    times[name].append((time.perf_counter() - t) / steps)
for name, ts in times.items():
  best = min(ts)
  print(f"AB {task} {n} envs | {name:32s} | ms per step {' '.join(f'{x * 1e3:.3f}' for x in ts)} | best {n / best / 1e6:.3f} M env-steps/s")
