"""Distribution of solver work per world after a rollout (GPU box)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots  # noqa: E402
from mjlab_amd.rollout import PhysicsRollout, g1_action_scale  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402

model = robots.load_model("g1_velocity_flat")
sim = Simulation(4096, SimulationCfg(njmax=300), model, "cuda:0")
roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=42)
for _ in range(60):
  roll.step(roll.random_action())
sim.step()
torch.cuda.synchronize()
it = sim.data.solver_niter.cpu().numpy().ravel()
ne = sim.data.nefc.cpu().numpy().ravel()
nc = sim.data.ncon.cpu().numpy().ravel()
print("niter hist", np.bincount(it, minlength=11))
print("nefc  pct [0,25,50,75,90,99,100]", np.percentile(ne, [0, 25, 50, 75, 90, 99, 100]))
print("ncon  pct", np.percentile(nc, [0, 25, 50, 75, 90, 99, 100]))
print("mean niter", it.mean(), "mean nefc", ne.mean())
