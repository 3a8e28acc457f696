import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import reference_env
from mjlab_amd.graphed_env import GraphedRlEnv
n = 4096
env = reference_env.make_env("Mjlab-Velocity-Flat-Unitree-G1", num_envs=n, device="cuda:0")
env.reset()
g = GraphedRlEnv(env)
a = torch.zeros((n, 29), device="cuda:0")
for _ in range(20):
  g.step(a)
torch.cuda.synchronize()
# host time of replay alone (queue empty at start): one replay then sync
host, total = [], []
for _ in range(30):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  g.graph.replay()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  host.append(t1 - t0); total.append(t2 - t0)
print(f"HOST one replay from an idle queue: host call {1e3 * sum(host) / len(host):.3f} ms, until the GPU is done {1e3 * sum(total) / len(total):.3f} ms")
# back to back
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
  g.graph.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"HOST 100 replays back to back: host loop {1e3 * (t1 - t0) / 100:.3f} ms per replay, until done {1e3 * (t2 - t0) / 100:.3f} ms per replay")
