"""Kernel launch floor: time of a forward whose worlds are all masked out (GPU box)."""
import sys, ctypes
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots, native
from mjlab_amd.sim import Simulation, SimulationCfg
model = robots.load_model("g1_velocity_flat")
sim = Simulation(4096, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
mask = torch.zeros(4096, dtype=torch.bool, device="cuda")
for _ in range(5):
  sim.forward(mask)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
sim.data.world_mask.zero_()
e0.record()
for _ in range(n):
  native.check(sim._lib.mjlab_forward_masked(ctypes.byref(sim._m), ctypes.byref(sim._d), sim._stream()), "f")
e1.record()
torch.cuda.synchronize()
print("5 empty stage kernels: %.1f us per forward -> %.1f us per kernel" % (e0.elapsed_time(e1) / n * 1e3, e0.elapsed_time(e1) / n * 1e3 / 5))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
  for _ in range(10):
    native.check(sim._lib.mjlab_forward_masked(ctypes.byref(sim._m), ctypes.byref(sim._d), sim._stream()), "f")
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(50):
  g.replay()
e1.record(); torch.cuda.synchronize()
print("in a hipGraph: %.1f us per kernel" % (e0.elapsed_time(e1) / 50 / 50 * 1e3))
