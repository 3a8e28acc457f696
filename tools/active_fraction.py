"""Fraction of constraint rows that are active (force > 0) at the solver's solution (GPU box)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots
from mjlab_amd.rollout import PhysicsRollout, g1_action_scale
from mjlab_amd.sim import Simulation, SimulationCfg
model = robots.load_model("g1_velocity_flat")
sim = Simulation(4096, SimulationCfg(njmax=300), model, "cuda:0")
roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=42)
for _ in range(40):
  roll.step(roll.random_action())
sim.step()
torch.cuda.synchronize()
d = sim.data
nefc = d.nefc.view(-1).float()
rows = torch.arange(sim.njmax, device="cuda")[None, :] < d.nefc.view(-1, 1)
act = ((d.efc_force > 0) & rows).sum(dim=1).float()
lim = ((d.efc_type == 3) & rows).sum(dim=1).float()
print("nefc mean %.1f  active mean %.1f  (%.0f%%)  limit rows mean %.1f" % (nefc.mean(), act.mean(), 100 * act.sum() / nefc.sum(), lim.mean()))
big = nefc > 48
print("worlds with nefc>48: %d; their nefc %.1f active %.1f" % (int(big.sum()), nefc[big].mean(), act[big].mean()))
