"""Parity GATE: HIP path vs the CPU oracle over the state distribution of rollouts.

`tools/parity_report.py::scene_report` rolls 1024 worlds per scene on the GPU with random actions
(falls, self-collisions and resets included), hands the reached states to the oracle and compares
one forward() and one step().  Two horizons (25 and 250 control steps), the fp64 and the fp32 build
of the oracle, per-world model randomisation (friction everywhere; torso com and joint zero offsets
on the tracking scene, like the reference's startup events).

Tolerances (relative, per world: max |gpu - oracle| / max |oracle|), from north_star's "1e-5 rel
fp32" and the measured distributions (profiles/r02_*/parity_report*.txt; the literals are the
measured worst case x3 or the north_star figure, whichever is larger):

  counts / sensordata      identical in >= 99 % of the worlds (a contact sitting exactly on its
                           margin may flip between fp32 and fp64); everything below is over the
                           worlds with identical counts
  kinematics, qM           max <= 1e-6 .. 2e-6
  velocity-stage outputs   max <= 1e-5 .. 4e-5 (flat) -- qfrc_smooth carries the PD actuator force
                           kp (q_des - q), a difference of two fp32 numbers
  efc_J                    max <= 1e-5
  efc_D / efc_aref         functions of penetration / 1 mm (solimp width): a 1e-7 position error is
                           amplified 1e3-fold; gated at p99
  qacc, qfrc_constraint    p99 <= 1e-5, max <= 5e-5 (flat)
  one step later           qpos max <= 1e-6, qvel max <= 1e-5 (flat)

On the rough scenes the robots are ~100 m from the origin, where fp32 world coordinates resolve
7.6 um: kinematics stay relative-exact, but contact depths (and with them efc_D, qacc) lose three
digits -- separate, looser literals below, with the same structure.
"""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))

pytestmark = pytest.mark.gpu

N = 1024
FLAT = {
  "same_frac": 0.99,
  "kin_max": 1e-6, "qM_max": 2e-6,
  "vel_max": 4e-5,
  "efc_J_max": 1e-5, "efc_pos_max": 2e-4, "efc_D_p99": 2e-3, "efc_aref_p99": 1e-3,
  "qacc_p99": 1e-5, "qacc_max": 5e-5,
  "step_qpos_max": 1e-6, "step_qvel_max": 1e-5,
}  # fmt: skip
ROUGH = {
  "same_frac": 0.98,
  "kin_max": 1e-6, "qM_max": 2e-6,
  "vel_max": 4e-5,
  "efc_J_max": 5e-5, "efc_pos_max": 5e-3, "efc_D_p99": 2e-2, "efc_aref_p99": 1e-2,
  "qacc_p99": 5e-4, "qacc_max": 5e-3,
  "step_qpos_max": 5e-6, "step_qvel_max": 5e-4,
}  # fmt: skip

CASES = [
  # scene, control steps, oracle precision, expanded model fields
  ("go1_velocity_flat", 25, "f64", ("geom_friction",)),
  ("go1_velocity_flat", 250, "f64", ("geom_friction",)),
  ("g1_velocity_flat", 25, "f64", ("geom_friction",)),
  ("g1_velocity_flat", 250, "f64", ("geom_friction",)),
  ("g1_velocity_flat", 250, "f32", ("geom_friction",)),
  ("g1_tracking_flat", 25, "f64", ("geom_friction", "body_ipos", "qpos0")),
  ("g1_tracking_flat", 250, "f64", ("geom_friction", "body_ipos", "qpos0")),
  ("g1_tracking_flat", 250, "f32", ("geom_friction", "body_ipos", "qpos0")),
  ("g1_velocity_rough", 25, "f64", ("geom_friction",)),
  ("g1_velocity_rough", 250, "f64", ("geom_friction",)),
  ("go1_velocity_rough", 25, "f64", ("geom_friction",)),
  ("go1_velocity_rough", 250, "f64", ("geom_friction",)),
]


def _check(r, tol):
  from parity_report import KIN, VEL, format_report

  print(format_report(r))
  n, same = r["n"], r["same_counts"]
  assert same >= tol["same_frac"] * n, f"identical counts in only {same} of {n} worlds"
  assert r["same_sensordata"] >= same, "sensordata differs in a world with identical contact / row counts"
  assert r["overflow_gpu"] == r["overflow_oracle"] == 0
  f = r["fields"]
  for k in KIN:
    assert f[k][2] <= (tol["qM_max"] if k == "qM" else tol["kin_max"]), (k, f[k])
  for k in VEL:
    assert f[k][2] <= tol["vel_max"], (k, f[k])
  assert f["efc_J"][2] <= tol["efc_J_max"], f["efc_J"]
  assert f["efc_pos"][2] <= tol["efc_pos_max"], f["efc_pos"]
  assert f["efc_D"][1] <= tol["efc_D_p99"], f["efc_D"]
  assert f["efc_aref"][1] <= tol["efc_aref_p99"], f["efc_aref"]
  for k in ("qacc", "qfrc_constraint"):
    assert f[k][1] <= tol["qacc_p99"] and f[k][2] <= tol["qacc_max"], (k, f[k])
  assert f["step_qpos"][2] <= tol["step_qpos_max"], f["step_qpos"]
  assert f["step_qvel"][2] <= tol["step_qvel_max"], f["step_qvel"]
  # the Newton iteration does the same amount of work on both sides
  assert abs(r["niter_gpu"][0] - r["niter_oracle"][0]) < 0.25, (r["niter_gpu"], r["niter_oracle"])


@pytest.mark.parametrize("scene,steps,precision,expand", CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}" for c in CASES])
def test_rollout_state_parity(scene, steps, precision, expand):
  from parity_report import scene_report

  rough = scene.endswith("rough")
  r = scene_report(scene, N, steps, precision, expand=expand, spread=3.5 if rough else None)
  _check(r, ROUGH if rough else FLAT)
  if rough:
    # the compared states really are on the stairs, not on the flat spawn platforms
    assert r["worlds_with_terrain_contact"] >= 0.8 * N
    assert r["worlds_with_edge_contact"] >= 0.1 * N, r["worlds_with_edge_contact"]


def test_literal_termination_switch_matches_the_literal_oracle():
  """MJLAB_OPT_LITERAL_TERMINATION: MuJoCo's tolerance / gtol rules only, on both sides.  The fp32
  solver then runs into the iteration caps more often (it cannot resolve 1e-8), but lands on the
  same solution to the same tolerance."""
  from parity_report import scene_report

  r = scene_report("g1_velocity_flat", 512, 25, "f64", expand=(), flags={"literal_termination": True})
  f = r["fields"]
  assert r["same_counts"] >= 0.99 * r["n"]
  assert f["qacc"][1] <= 3e-5 and f["qacc"][2] <= 2e-4, f["qacc"]
  assert r["niter_gpu"][0] >= r["niter_oracle"][0] - 0.1  # noise floors off: never fewer iterations than fp64


def test_warmstart_at_advance_switch():
  """MJLAB_OPT_WARMSTART_AT_ADVANCE: forward() leaves qacc_warmstart alone, step() saves qacc."""
  import torch

  from mjlab_amd import robots
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model("go1_velocity_flat")
  for at_advance in (False, True):
    sim = Simulation(16, SimulationCfg(njmax=300, warmstart_at_advance=at_advance), model, "cuda:0")
    sim.data.qpos[:, 2] -= 0.05
    sim.data.qacc_warmstart[:] = 0.125
    sim.forward()
    torch.cuda.synchronize()
    untouched = bool((sim.data.qacc_warmstart == 0.125).all())
    assert untouched == at_advance
    sim.step()
    torch.cuda.synchronize()
    assert torch.equal(sim.data.qacc_warmstart, sim.data.qacc)
