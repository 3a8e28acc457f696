"""Kernel census of one GraphedRlEnv control step (GPU box, reference staged): run under `rocprofv3 --kernel-trace --stats`.
  python tools/graphed_env_profile.py [num_envs] [steps] [task] [random]     (task: a registered task id; default the G1 velocity-flat
  task; "random": a random policy instead of zero actions; many steps = a soak run: finiteness, overflow and memory are reported)"""
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
rnd = len(sys.argv) > 4 and sys.argv[4] == "random"
gen = torch.Generator(device="cuda:0")
gen.manual_seed(0)
for _ in range(steps):
  g.step(2.0 * torch.rand(a.shape, device="cuda:0", generator=gen) - 1.0 if rnd else a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
obs = g.step(a)[0]
torch.cuda.synchronize()
finite = all(bool(torch.isfinite(o).all()) for o in obs.values()) and bool(torch.isfinite(env.sim.data.qpos).all()) and bool(torch.isfinite(env.sim.data.qvel).all())
print(f"SOAK {steps + 11} steps: observations and state finite: {finite}; episode length min / max {int(env.episode_length_buf.min())} / {int(env.episode_length_buf.max())}; "
      f"overflow {env.sim.overflow_report()}; allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB")
print(f"GRAPHED {task} {n} envs: {dt * 1e3:.3f} ms per step, {n / dt:.0f} env-steps/s, graph nodes replayed per step: see the kernel trace / {steps + 10 + 3} bodies")
