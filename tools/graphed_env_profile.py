"""Kernel census of one GraphedRlEnv control step (GPU box, reference staged): run under `rocprofv3 --kernel-trace --stats`.
  python tools/graphed_env_profile.py [num_envs] [steps] [task]     (task: a registered task id; default the G1 velocity-flat task)"""
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
task = sys.argv[3] if len(sys.argv) > 3 else "Mjlab-Velocity-Flat-Unitree-G1"
edit = None
if "Tracking" in task:
  from mjlab_amd import robots
  from mjlab_amd.rollout import write_motion_npz

  path = str(Path(tempfile.mkdtemp()) / "motion.npz")
  write_motion_npz(path, robots.load_model("g1_tracking_flat"), "cuda:0")

  def edit(cfg):
    cfg.commands.motion.motion_file = path

env = reference_env.make_env(task, num_envs=n, device="cuda:0", cfg_edit=edit)
env.reset()
g = GraphedRlEnv(env)
a = torch.zeros((n, sum(env.action_manager.action_term_dim)), device="cuda:0")
for _ in range(10):
  g.step(a)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
  g.step(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
print(f"GRAPHED {task} {n} envs: {dt * 1e3:.3f} ms per step, {n / dt:.0f} env-steps/s, graph nodes replayed per step: see the kernel trace / {steps + 10 + 3} bodies")
