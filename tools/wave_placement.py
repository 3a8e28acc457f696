"""Where the workgroups of one k_control_step launch run (profiling build, GPU box): HW_ID / XCC_ID per blockIdx.
Answers which workgroups share a SIMD -- what `mjlab_control_t.world_order` needs to know to keep two expensive worlds
off the same SIMD.

  MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so python tools/wave_placement.py
"""
import sys
from collections import Counter
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots  # noqa: E402
from mjlab_amd.rollout import VELOCITY_TASK_EVENTS, PhysicsRollout, g1_action_scale  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402

model = robots.load_model("g1_velocity_flat")
sim = Simulation(4096, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=42, substeps_per_call=4, control_kernel=True, **VELOCITY_TASK_EVENTS["g1"])
maps = []
for k in range(4):
  for _ in range(5):
    roll.step(roll.random_action())
  torch.cuda.synchronize()
  p = sim.data.profile.cpu().numpy()
  hw, xcc = p[:, 61].astype(np.int64), p[:, 62].astype(np.int64)
  simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
  maps.append(np.stack([xcc, se, sh, cu, simd], axis=1))
m = maps[-1]
print("launch-to-launch identical placement:", [bool((maps[i] == m).all()) for i in range(3)])
key = [tuple(r) for r in m]
cnt = Counter(key)
print("distinct (xcc, se, sh, cu, simd):", len(cnt), " waves per SIMD min / max:", min(cnt.values()), max(cnt.values()))
print("distinct xcc:", sorted(set(m[:, 0])), " se:", sorted(set(m[:, 1])), " sh:", sorted(set(m[:, 2])), " cu:", sorted(set(m[:, 3])))
print("first 24 workgroups (xcc, se, sh, cu, simd):", key[:24])
first = {}
for b, kk in enumerate(key):
  first.setdefault(kk, []).append(b)
some = list(first.items())[:6]
print("workgroups sharing a SIMD (examples):", [v for _, v in some])
strides = Counter()
for v in first.values():
  for a, b in zip(v, v[1:]):
    strides[b - a] += 1
print("blockIdx differences between SIMD mates:", strides.most_common(8))
np.save(Path(__file__).resolve().parents[1] / "gpurun_out" / "wave_placement.npy", m) if (Path(__file__).resolve().parents[1] / "gpurun_out").is_dir() else None
