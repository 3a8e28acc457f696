"""Where the launches of one GraphedRlEnv control step come from, by OWNER (runs on the CPU over the oracle: no GPU needed).

A hipGraph node is a kernel launch, and outside the physics a launch is one dispatched torch operator that is not a view.  This tool
runs the step body uncaptured (what tests/test_graphed_env.py teacher-forces against the reference's eager step) under a
``TorchDispatchMode`` and attributes every non-view operator to the innermost stack frame that belongs to a term function, a manager,
``EntityData``, ``MotionCommand`` or this package -- the launch-count table VERDICT round 5 (item 2) asks for before anything else is
fused: which part of the ~600 nodes of the tracking task's graph is reachable from the boundary (``EntityData`` quantities: row f1)
and which part is the reference's own term arithmetic (``envs/mdp``, ``tasks/*/mdp``: SURVEY section 2 OUT OF SCOPE).

On the GPU the terms this package restates as ONE HIP launch (reset events, command resampling, the reward accumulation, masked
fills / sums, the entity read-back) run as torch twins here: their operators are listed under their owner and counted as one launch
per call in the "on the device" column.

  python tools/graphed_env_census.py [task] [num_envs]
  python tools/graphed_env_census.py [task] [num_envs] cuda:0     (GPU box, reference staged: the step body uncaptured over the HIP simulation
                                                                  with the fused terms ON -- every HIP launch of this package is counted
                                                                  where mjlab_amd.native.check reports it, every other launch is a torch
                                                                  operator; scripted math helpers fuse further under the graph's profile)
"""
from __future__ import annotations

import collections
import sys
import tempfile
from pathlib import Path

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "tests"))

VIEW_OPS = {"view", "_unsafe_view", "select", "slice", "as_strided", "expand", "unsqueeze", "squeeze", "permute", "transpose", "t", "detach", "alias",
            "unbind", "split", "split_with_sizes", "reshape", "_reshape_alias", "unfold", "diagonal", "narrow", "view_as_real", "view_as_complex",
            "lift_fresh", "is_same_size", "sym_size", "sym_stride", "sym_numel", "_local_scalar_dense", "item", "real", "imag", "movedim", "chunk",
            "expand_as", "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided"}


def owner_of_stack() -> tuple[str, str]:
  """(group, function) of the innermost frame that names an owner."""
  f = sys._getframe(2)
  best = None
  while f is not None:
    fn = f.f_code.co_filename
    name = f.f_code.co_name
    if "/mjlab/tasks/" in fn and "/mdp/" in fn:
      return ("task term (tasks/*/mdp)" if "commands.py" not in fn else "MotionCommand (tasks/tracking/mdp/commands.py)"), f"{Path(fn).name}:{name}"
    if "/mjlab/envs/mdp/" in fn:
      return "env term (envs/mdp)", f"{Path(fn).name}:{name}"
    if "/mjlab/entity/data.py" in fn:
      return "EntityData (entity/data.py)", name
    if "isaaclab/utils/math.py" in fn and best is None:
      best = ("math helper called from: ", name)
    if "/mjlab/managers/" in fn:
      return "manager (managers/*.py)", f"{Path(fn).name}:{name}"
    if "/mjlab_amd/env_core.py" in fn or "/mjlab_amd/env_terms.py" in fn:
      return "restated piece, ONE launch on the device (env_core / env_terms)", name
    if "/mjlab_amd/entity_data.py" in fn:
      return "entity read-back, ONE launch on the device", name
    if "/mjlab_amd/graphed_env.py" in fn:
      return "GraphedRlEnv (mask-based bookkeeping)", name
    if "_oracle_simulation" in fn or "/oracle/" in fn or "/mjlab_amd/sim" in fn:
      return "physics (one launch on the device)", name
    f = f.f_back
  return "other", "?"


class Census(TorchDispatchMode):
  def __init__(self):
    super().__init__()
    self.ops = collections.Counter()
    self.by_owner = collections.defaultdict(collections.Counter)
    self.who = collections.defaultdict(collections.Counter)  # operator -> (group: function) for the data-movement operators

  def __torch_dispatch__(self, func, types, args=(), kwargs=None):
    name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
    if name not in VIEW_OPS:
      grp, fn = owner_of_stack()
      self.by_owner[grp][fn] += 1
      self.ops[name] += 1
      if name in ("copy_", "clone", "index", "index_put_", "cat", "stack", "repeat", "_to_copy", "index_select", "gather", "where", "masked_fill_", "fill_", "zero_"):
        self.who[name][f"{grp.split(' (')[0]}: {fn}"] += 1
    return func(*args, **(kwargs or {}))


def main():
  task = sys.argv[1] if len(sys.argv) > 1 else "Mjlab-Tracking-Flat-Unitree-G1"
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
  device = sys.argv[3] if len(sys.argv) > 3 else "cpu"
  import reference_env

  from mjlab_amd.graphed_env import GraphedRlEnv

  edit = None
  if "Tracking" in task:
    from _motion_fixture import write_full_motion

    path = str(Path(tempfile.mkdtemp()) / "motion.npz")
    write_full_motion(path)

    def edit(cfg):
      cfg.commands.motion.motion_file = path

  hip = collections.Counter()
  if device == "cpu":
    from _oracle_simulation import OracleSimulation

    env = reference_env.make_env(task, num_envs=n, device="cpu", sim_cls=OracleSimulation, seed=7, cfg_edit=edit)
  else:
    from mjlab_amd import native

    env = reference_env.make_env(task, num_envs=n, device=device, seed=7, cfg_edit=edit)
    orig_check = native.check

    def counting_check(rc, what):  # every HIP launch of this package reports through native.check(rc, name)
      hip[what] += 1
      return orig_check(rc, what)

    native.check = counting_check
  env.reset()
  g = GraphedRlEnv(env, capture=False)
  na = sum(env.action_manager.action_term_dim)
  for _ in range(3):
    g.step(torch.rand(n, na, device=device) * 2 - 1)
  hip.clear()
  c = Census()
  steps = 4
  with c:
    for _ in range(steps):
      g.step(torch.rand(n, na, device=device) * 2 - 1)
  total = sum(sum(v.values()) for v in c.by_owner.values())
  print(f"{task}: {total / steps:.0f} non-view operators per control step on {device} (uncaptured body, {n} envs, mean of {steps} steps)")
  print(f"{'owner':70s} {'ops/step':>9s} {'launches/step on the device':>28s}")
  dev_total = 0.0
  for grp, fns in sorted(c.by_owner.items(), key=lambda kv: -sum(kv[1].values())):
    ops = sum(fns.values()) / steps
    dev = len(fns) * 1.0 if "ONE launch" in grp else (1.0 if grp.startswith("physics") else ops)
    if "ONE launch" in grp:
      dev = sum(1 for _ in fns)  # one launch per restated function called (each is called once per step or once per phase)
    dev_total += dev
    print(f"{grp:70s} {ops:9.1f} {dev:28.1f}")
    for fn, k in sorted(fns.items(), key=lambda kv: -kv[1])[:40]:
      print(f"    {fn:66s} {k / steps:9.1f}")
  print(f"{'TOTAL (estimate of graph nodes per step)':70s} {total / steps:9.1f} {dev_total:28.1f}")
  if hip:
    print(f"HIP launches of this package per step ({sum(hip.values()) / steps:.1f}):", ", ".join(f"{k} {v / steps:.1f}" for k, v in hip.most_common()))
  for opn in ("copy_", "clone", "index", "cat", "repeat", "where"):
    if c.who[opn]:
      print(f"{opn} ({sum(c.who[opn].values()) / steps:.0f} per step):", "; ".join(f"{k} {v / steps:.1f}" for k, v in c.who[opn].most_common(14)))
  print("most frequent operators:", ", ".join(f"{k} {v / steps:.0f}" for k, v in c.ops.most_common(14)))


if __name__ == "__main__":
  main()
