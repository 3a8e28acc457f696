"""Why does forward() on all worlds cost twice as much in the tracking scene as in the velocity scene?  Newton iterations and rows after the
forward pass, over all worlds and over the worlds that just reset, and the pass timed alone (GPU box)."""
import sys, torch
sys.path.insert(0, '.')
from mjlab_amd import robots
from mjlab_amd.rollout import PhysicsRollout, g1_action_scale
from mjlab_amd.sim import Simulation, SimulationCfg
import bench
for scene in ("g1_velocity_flat", "g1_tracking_flat"):
  model = robots.load_model(scene)
  sim = Simulation(4096, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
  tracking = "tracking" in scene
  events = dict((bench.TRACKING_TASK_EVENTS if tracking else bench.VELOCITY_TASK_EVENTS)["g1"])
  if "motion_reset" in events:
    events["motion"] = bench.synthetic_motion(model)
  roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), decimation=4, seed=42, min_height=-1e9 if tracking else 0.3, control_kernel=False, fused_reset=not tracking, **events)
  for _ in range(150):
    roll.step(roll.random_action())
  torch.cuda.synchronize()
  # one more control step by hand: physics, reset, then time forward alone and look at its iterations
  stats = []
  for _ in range(20):
    rm = roll.step(roll.random_action())
    torch.cuda.synchronize()
    nit = sim.data.solver_niter.flatten().float(); nefc = sim.data.nefc.flatten().float()
    rm = rm.bool() if isinstance(rm, torch.Tensor) else None
    stats.append((nit.mean().item(), nit.max().item(), nefc.mean().item(), nefc.max().item(), (rm.float().mean().item() if rm is not None else -1),
                  (nit[rm].mean().item() if rm is not None and rm.any() else -1), (nefc[rm].mean().item() if rm is not None and rm.any() else -1)))
  import numpy as np
  s = np.array(stats)
  print(scene, "after forward: niter mean %.2f max %.0f | nefc mean %.1f max %.0f | reset frac %.4f | reset worlds: niter %.2f nefc %.1f" % tuple(s.mean(axis=0)))
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  for name, fn in (("forward", sim.forward), ("step(4)", lambda: sim.step(4))):
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("  ", name, "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
