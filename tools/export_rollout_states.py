"""Export the ROLLOUT STATES the parity gate compares on (GPU box): 256 worlds per scene after 250 control steps of random
actions (falls, self-collisions, resets; per-world friction / torso com / joint zero offsets like tests/test_gpu_parity_gate.py),
as small fp32 fixtures tests/golden/rollout_states_<scene>.npz: qpos, qvel, ctrl, qacc_warmstart + the per-world model fields
(dr_<field>).  tools/dump_mjwarp_reference.py feeds exactly these states to the pinned upstream engine; tests/test_golden.py
feeds them to the oracle and the HIP path.

  python tools/export_rollout_states.py [out_dir]      # default gpurun_out/rollout_states (copy to tests/golden/ and commit)
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from parity_report import randomize_model  # noqa: E402

from mjlab_amd import robots  # noqa: E402
from mjlab_amd.rollout import TRACKING_TASK_EVENTS, PhysicsRollout, g1_action_scale, go1_action_scale, synthetic_motion  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402
from oracle.oracle import OracleSim  # noqa: E402

SCENES = {"g1_velocity_flat": ("geom_friction",), "g1_tracking_flat": ("geom_friction", "body_ipos", "qpos0"), "go1_velocity_flat": ("geom_friction",)}
N, STEPS, SEED = 256, 250, 123


def main(out_dir: str = "gpurun_out/rollout_states") -> None:
  out = ROOT / out_dir
  out.mkdir(parents=True, exist_ok=True)
  for scene, expand in SCENES.items():
    model = robots.load_model(scene)
    njmax = 300 if "velocity" in scene else 250
    sim = Simulation(N, SimulationCfg(njmax=njmax, use_graph=False), model, "cuda:0")
    ora = OracleSim(model, N, njmax=njmax, precision="f64")  # receives the same per-world fields (randomize_model writes both)
    randomize_model(sim, ora, model, expand, SEED + 1)
    scale = g1_action_scale(model) if scene.startswith("g1") else go1_action_scale(model)
    if scene == "g1_tracking_flat":
      ev = TRACKING_TASK_EVENTS["g1"]
      roll = PhysicsRollout(sim, action_scale=scale, seed=SEED, min_height=-1.0e9, fused_reset=False, motion=synthetic_motion(model),
                            motion_reset=ev["motion_reset"], push=ev["push"], episode_length_s=ev["episode_length_s"])
    else:
      roll = PhysicsRollout(sim, action_scale=scale, seed=SEED, min_height=0.3 if scene.startswith("g1") else 0.15)
    nreset = 0
    for _ in range(STEPS):
      nreset += int(roll.step(roll.random_action()).sum())
    sim.data.ctrl[:] = roll.default_joint + roll.random_action() * roll.action_scale
    torch.cuda.synchronize()
    blob = {f: getattr(sim.data, f).cpu().numpy().astype(np.float32) for f in ("qpos", "qvel", "ctrl", "qacc_warmstart")}
    for f in expand:
      blob["dr_" + f] = getattr(sim.model, f).cpu().numpy().astype(np.float32)
    blob["meta"] = np.array([N, STEPS, SEED, nreset])
    np.savez_compressed(out / f"rollout_states_{scene}.npz", **blob)
    print(scene, "resets on the way", nreset, {k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
  main(*sys.argv[1:])
