"""Per-stage kernel time vs number of worlds (GPU box): tells issue-bound from latency-bound."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots  # noqa: E402
from mjlab_amd.rollout import PhysicsRollout, g1_action_scale  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "g1_velocity_flat"
model = robots.load_model(scene)
stages = [("position", 1), ("collision", 2), ("velocity", 4), ("constraint", 8), ("solve_integrate", 48)]
for nworld in (256, 1024, 2048, 4096, 8192, 16384):
  sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
  roll = PhysicsRollout(sim, action_scale=g1_action_scale(model) if scene.startswith("g1") else 0.25, seed=42, min_height=0.3 if scene.startswith("g1") else 0.15)
  for _ in range(30):
    roll.step(roll.random_action())
  acc = {k: 0.0 for k, _ in stages}
  n = 0
  for _ in range(5):
    sim.data.ctrl[:] = roll.default_joint + roll.random_action() * roll.action_scale
    for _ in range(4):
      evs = []
      for name, bits in stages:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sim.forward_stages(bits); e1.record()
        evs.append((name, e0, e1))
      torch.cuda.synchronize()
      for name, e0, e1 in evs:
        acc[name] += e0.elapsed_time(e1)
      n += 1
  tot = sum(acc.values()) / n
  print(f"nworld {nworld:6d}: " + "  ".join(f"{k} {v / n * 1e3:7.1f}us" for k, v in acc.items()) + f"  total {tot * 1e3:7.1f}us  -> {nworld / tot / 1e3:7.1f} M world-steps/s")
  del sim, roll
  torch.cuda.empty_cache()
