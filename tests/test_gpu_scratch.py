"""No kernel's result may depend on what the previous kernel left in scratch (the private segment).

Round 6 root cause of the "memory aperture violation" of round 5: hipcc placed a spill STORE in the exit block of a
divergent loop ahead of the `s_or_b64 exec, exec, sN` that gives the lanes back, so it wrote nothing, and the reload returned whatever
the slot held -- zeros in a fresh process, another kernel's spills after a kernel with a different frame had run.  The static side is
tests/test_code_object.py (tools/exec_zero_check.py finds that instruction pattern in the code objects); this is the dynamic side:
every launch structure of every padded size is run once as it is and once with `mjlab_poison_scratch` in front of EVERY launch, and
the two runs must agree bit for bit.  A kernel that reloads a slot it never wrote turns the poison (NaN / non-canonical address)
into a fault or a different result here, whatever ran before it in the process.  Replaces the calls of reference
src/mjlab/sim/sim.py:136,189-195 like every other test of the step.
"""

import copy
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

from make_golden import golden_inputs, models  # noqa: E402

pytestmark = pytest.mark.gpu

FIELDS = ("qpos", "qvel", "qacc", "qacc_warmstart", "xpos", "cvel", "qfrc_constraint", "sensordata", "actuator_force")


def _run(name, fuse, cone, lsp, poison, nworld=64, seed=3):
  import torch

  from mjlab_amd import mjcf, native
  from mjlab_amd.sim import Simulation, SimulationCfg

  L = native.lib()
  model = copy.deepcopy(models()[name])
  if cone:
    model.opt.cone = mjcf.CONE_ELLIPTIC
  qpos, qvel, ctrl = golden_inputs(model, nworld, seed)
  sim = Simulation(nworld, SimulationCfg(njmax=300, fuse=fuse, ls_parallel=lsp, use_graph=False), model, "cuda:0")
  st = torch.cuda.current_stream().cuda_stream

  def dirty():
    if poison:
      native.check(L.mjlab_poison_scratch(nworld, st), "mjlab_poison_scratch")

  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
  sim.data.qacc_warmstart.zero_()
  dirty(); sim.forward()
  for _ in range(2):
    dirty(); sim.step()
  dirty(); sim.step(4)
  dirty(); sim.forward()
  torch.cuda.synchronize()
  return {f: getattr(sim.data, f).clone() for f in FIELDS}


SCENES = ["box", "go1_velocity_flat", "mixed", "g1_velocity_flat"]  # padded sizes 8, 20, 32, 36


@pytest.mark.parametrize("cone", [False, True], ids=["pyramid", "elliptic"])
@pytest.mark.parametrize("fuse", ["step", "stage"])
@pytest.mark.parametrize("name", SCENES)
def test_results_do_not_depend_on_stale_scratch(name, fuse, cone):
  import torch

  for lsp in (True, False):
    clean = _run(name, fuse, cone, lsp, poison=False)
    dirty = _run(name, fuse, cone, lsp, poison=True)
    for f in FIELDS:
      assert bool(torch.isfinite(dirty[f]).all()), (name, fuse, cone, lsp, f)
      assert torch.equal(clean[f], dirty[f]), (name, fuse, cone, lsp, f, float((clean[f] - dirty[f]).abs().max()))


def test_control_step_kernel_does_not_depend_on_stale_scratch():
  """The headline kernel (k_control_step<36>: 16 spilled VGPRs) and its cone variant: a rollout with task events, resets included,
  with poison in front of every control step, equals the clean one bit for bit."""
  import torch

  from mjlab_amd import mjcf, native
  from mjlab_amd.rollout import VELOCITY_TASK_EVENTS, PhysicsRollout
  from mjlab_amd.sim import Simulation, SimulationCfg

  L = native.lib()
  for cone in (False, True):
    out = {}
    for poison in (False, True):
      model = copy.deepcopy(models()["g1_velocity_flat"])
      if cone:
        model.opt.cone = mjcf.CONE_ELLIPTIC
      sim = Simulation(256, SimulationCfg(njmax=300, fuse="step"), model, "cuda:0")
      roll = PhysicsRollout(sim, action_scale=0.25, seed=11, min_height=0.3, control_kernel=True, **VELOCITY_TASK_EVENTS["g1"])
      st = torch.cuda.current_stream().cuda_stream
      resets = 0
      for _ in range(12):
        if poison:
          native.check(L.mjlab_poison_scratch(256, st), "mjlab_poison_scratch")
        resets += int(roll.step(roll.random_action()).sum())
      torch.cuda.synchronize()
      out[poison] = {f: getattr(sim.data, f).clone() for f in FIELDS}
      assert bool(torch.isfinite(sim.data.qpos).all())
    for f in FIELDS:
      assert torch.equal(out[False][f], out[True][f]), (cone, f)
