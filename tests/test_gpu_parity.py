"""HIP path vs the CPU oracle on seeded inputs, through the Simulation boundary / C ABI.

Tolerances: the device computes in fp32, the oracle in fp64; north_star asks for 1e-5 relative.
Every literal below is the worst relative error measured on the GPU (recorded per assertion in
gpurun_out/parity_margins.txt by teardown_module; profiles/r02_v1/parity_margins.txt) times 3,
rounded up to 1 / 2 / 5 x 10^k: kinematics and the mass matrix 1e-6, velocity-stage outputs 2e-6,
efc_J 1e-6, qacc / qfrc_constraint 1e-5 .. 2e-5 on these seeded states (efc_D / efc_aref keep 1e-4:
they are functions of penetration / 1 mm, which amplifies a 1e-7 position error 1e3-fold).  The
distribution over rollout states is gated separately (tests/test_gpu_parity_gate.py).
"""

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))

pytestmark = pytest.mark.gpu

from make_golden import golden_inputs, models  # noqa: E402

from oracle.oracle import OracleSim  # noqa: E402

KIN = ("xpos", "xquat", "xmat", "xipos", "ximat", "xanchor", "xaxis", "geom_xpos", "geom_xmat", "site_xpos", "site_xmat",
       "subtree_com", "cinert", "cdof", "qM")  # fmt: skip  (qLD: the factor never leaves the solve kernel's LDS)
VEL = ("cvel", "cdof_dot", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "actuator_force", "qfrc_smooth")


_MARGINS: dict = {}  # (test id, line) -> largest relative error seen; written to gpurun_out/parity_margins.txt


def _rel(a, b):
  a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
  r = np.abs(a - b).max() / max(1e-6, np.abs(b).max()) if b.size else 0.0
  import inspect
  import os

  key = (os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], inspect.stack()[1].lineno)
  _MARGINS[key] = max(_MARGINS.get(key, 0.0), float(r))
  return r


def _elem(a, b, field):
  """Element-wise metric: the WORST element's |a - b| / (atol(field) + 1e-5 |b|); <= 1 means every
  element is within north_star's 1e-5 relative plus the field's absolute floor (tools/parity_report.py::ATOL) -- a small
  component next to a large one cannot hide behind the array's largest value, as it can in `_rel`."""
  sys.path.insert(0, str(ROOT / "tools"))
  from parity_report import ATOL, RTOL

  a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
  return float((np.abs(a - b) / (ATOL[field] + RTOL * np.abs(b))).max()) if b.size else 0.0


def teardown_module(module):
  """The tolerances in this file are literals; the margins they were met with are recorded next to
  the run (gpurun_out/, when present) so that the literals can be kept at measured x 3."""
  out = ROOT / "gpurun_out"
  if out.is_dir() and _MARGINS:
    with open(out / "parity_margins.txt", "w") as f:
      for (test, line), v in sorted(_MARGINS.items()):
        f.write(f"{v:10.3e}  {test}:{line}\n")


def _pair(name, nworld=8, seed=11, graph=False, njmax=300, lsp=False):
  """`lsp`: the line search, the SAME one on both sides (True = mujoco_warp's grid search: what the reference configures and what
  bench.py / every default Simulation run; False = MuJoCo's exact search)."""
  import torch

  from mjlab_amd.sim import Simulation, SimulationCfg

  model = models()[name]
  qpos, qvel, ctrl = golden_inputs(model, nworld, seed)
  sim = Simulation(nworld, SimulationCfg(njmax=njmax, use_graph=graph, ls_parallel=lsp), model, "cuda:0")
  assert sim.ls_parallel == lsp
  ora = OracleSim(model, nworld, njmax=njmax, precision="f64", ls_parallel=lsp)
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    getattr(ora, f)[:] = v
  return sim, ora, model


def _np(t):
  import torch

  torch.cuda.synchronize()
  return t.cpu().numpy()


@pytest.mark.parametrize("lsp", [False, True], ids=["exact_ls", "grid_ls"])
@pytest.mark.parametrize("name", ["box", "mixed", "go1_velocity_flat", "g1_velocity_flat", "g1_tracking_flat"])
def test_forward_all_fields(name, lsp):
  sim, ora, model = _pair(name, lsp=lsp)
  sim.forward()
  ora.forward()
  assert np.array_equal(_np(sim.data.ncon).ravel(), ora.ncon.ravel())
  assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  for f in KIN:
    assert _rel(_np(getattr(sim.data, f)), getattr(ora, f)) < 2e-6, f
  for f in VEL:
    assert _rel(_np(getattr(sim.data, f)), getattr(ora, f)) < 2e-06, f
  # constraint rows: same order by construction (pair order, limits first)
  nv = model.nv
  for w in range(sim.num_envs):
    n = int(ora.nefc[w, 0])
    Jg = _np(sim.data.efc_J)[w].reshape(-1, nv)[:n]
    assert _rel(Jg, ora.efc_J[w].reshape(-1, nv)[:n]) < 1e-06
    for f in ("efc_D", "efc_aref", "efc_pos"):
      assert _rel(_np(getattr(sim.data, f))[w, :n], getattr(ora, f)[w, :n]) < 1e-4, f
  assert _rel(_np(sim.data.qacc_smooth), ora.qacc_smooth) < 2e-05
  assert _rel(_np(sim.data.qacc), ora.qacc) < 2e-05
  assert _rel(_np(sim.data.qfrc_constraint), ora.qfrc_constraint) < 1e-05
  assert np.array_equal(_np(sim.data.sensordata), ora.sensordata.astype(np.float32))
  # element by element: every element of the kinematic and velocity-stage fields is within 1e-5 relative + the field's floor
  # (measured on the rollout-state gate: worst element 0.44 of the bound); the solutions of the ill-conditioned systems carry
  # the same ABSOLUTE noise in their small components as in their large ones (gate: up to ~10 x the bound in 1 % of the worlds)
  for f in ("xpos", "xquat", "xipos", "subtree_com", "geom_xpos", "site_xpos", "qM", "cvel", "qfrc_bias", "actuator_force", "qfrc_smooth"):
    assert _elem(_np(getattr(sim.data, f)), getattr(ora, f), f) <= 1.0, f
  for f in ("qacc_smooth", "qacc", "qfrc_constraint"):
    assert _elem(_np(getattr(sim.data, f)), getattr(ora, f), f) <= 30.0, f


@pytest.mark.parametrize("lsp", [False, True], ids=["exact_ls", "grid_ls"])
@pytest.mark.parametrize("name", ["pendulum", "box", "mixed", "go1_velocity_flat", "g1_velocity_flat"])
def test_rollout_state(name, lsp):
  if name == "pendulum":
    import torch

    from mjlab_amd import robots
    from mjlab_amd.sim import Simulation, SimulationCfg

    model = robots.pendulum_model()
    sim = Simulation(3, SimulationCfg(ls_parallel=lsp), model, "cuda:0")
    ora = OracleSim(model, 3, ls_parallel=lsp)
    q0 = np.array([[0.5], [-1.0], [2.5]])
    sim.data.qpos[:] = torch.from_numpy(q0.astype(np.float32)).cuda()
    ora.qpos[:] = q0
    nstep, tq, tv = 200, 1e-4, 1e-3
  else:
    sim, ora, model = _pair(name, lsp=lsp)
    nstep, tq, tv = 10, 5e-6, 2e-5
  for _ in range(nstep):
    sim.step()
  ora.step(nstep)
  assert _rel(_np(sim.data.qpos), ora.qpos) < tq
  assert _rel(_np(sim.data.qvel), ora.qvel) < tv
  assert _np(sim.data.time) == pytest.approx(ora.time.ravel(), rel=1e-5)


def test_graph_replay_equals_eager():
  a, _, _ = _pair("g1_velocity_flat", graph=False)
  b, _, _ = _pair("g1_velocity_flat", graph=True)
  assert b.step_graph is not None
  for _ in range(5):
    a.step()
    b.step()
  a.forward()
  b.forward()
  for f in ("qpos", "qvel", "qacc", "xpos", "sensordata"):
    assert np.array_equal(_np(getattr(a.data, f)), _np(getattr(b.data, f))), f


def test_single_world_and_odd_world_counts():
  for nworld in (1, 3, 67):
    sim, ora, _ = _pair("go1_velocity_flat", nworld=nworld)
    sim.step()
    ora.step()
    assert _rel(_np(sim.data.qpos), ora.qpos) < 1e-06


def test_row_capacity_overflow_is_consistent():
  # njmax smaller than the rows the state wants: both sides must drop the same contacts
  sim, ora, _ = _pair("g1_velocity_flat", njmax=24)
  sim.forward()
  ora.forward()
  assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  assert _np(sim.data.nefc).max() <= 24
  assert _rel(_np(sim.data.qacc), ora.qacc) < 2e-05
  # ... and both say so (data.overflow, MJLAB_OVF_NJMAX), instead of dropping rows silently
  assert np.array_equal(_np(sim.data.overflow).ravel(), ora.overflow.ravel())
  assert (_np(sim.data.overflow).ravel() & 2).any()
  with pytest.warns(UserWarning, match="capacity overflow"):
    assert sim.overflow_report()["njmax"] > 0


def test_no_contact_state_and_zero_ctrl():
  import torch

  sim, ora, model = _pair("g1_velocity_flat")
  sim.data.qpos[:, 2] += 2.0
  ora.qpos[:, 2] += 2.0
  sim.data.ctrl[:] = 0
  ora.ctrl[:] = 0
  sim.step()
  ora.step()
  assert int(_np(sim.data.ncon).max()) == 0
  assert _rel(_np(sim.data.qvel), ora.qvel) < 1e-05
  torch.cuda.synchronize()


def test_xfrc_and_qfrc_applied():
  import torch

  sim, ora, model = _pair("go1_velocity_flat")
  rng = np.random.default_rng(5)
  xf = np.zeros((sim.num_envs, model.nbody, 6))
  xf[:, 2] = rng.normal(0, 20, (sim.num_envs, 6))
  xf[::2, 7, :3] = rng.normal(0, 10, (sim.num_envs // 2, 3))
  qf = rng.normal(0, 2, (sim.num_envs, model.nv))
  sim.data.xfrc_applied[:] = torch.from_numpy(xf.astype(np.float32)).cuda()
  sim.data.qfrc_applied[:] = torch.from_numpy(qf.astype(np.float32)).cuda()
  ora.xfrc_applied[:] = xf
  ora.qfrc_applied[:] = qf
  sim.forward()
  ora.forward()
  assert _rel(_np(sim.data.qfrc_smooth), ora.qfrc_smooth) < 1e-06
  assert _rel(_np(sim.data.qacc_smooth), ora.qacc_smooth) < 1e-06


def test_expand_model_fields_per_world_friction():
  import torch

  sim, ora, model = _pair("g1_velocity_flat")
  with pytest.raises(ValueError, match="Fields not found"):
    sim.expand_model_fields(["not_a_field"])
  before = sim.model.geom_friction
  assert before.shape == (sim.num_envs, model.ngeom, 3) and before.stride(0) == 0
  sim.expand_model_fields(["geom_friction", "body_ipos", "qpos0"])
  fr = sim.model.geom_friction
  assert fr.stride(0) == model.ngeom * 3
  assert torch.equal(fr[5], torch.from_numpy(model.geom_friction.astype(np.float32)).cuda())
  rng = np.random.default_rng(9)
  feet = [i for i, n in enumerate(model.names["geom"]) if "foot" in n and n.endswith("collision")]
  vals = rng.uniform(0.3, 1.2, (sim.num_envs, len(feet)))
  ptr = fr.data_ptr()
  env_grid = torch.arange(sim.num_envs).cuda()[:, None]
  fr[env_grid, torch.tensor(feet).cuda()[None, :], 0] = torch.from_numpy(vals.astype(np.float32)).cuda()
  assert sim.model.geom_friction.data_ptr() == ptr
  ofr = ora.expand_model_field("geom_friction")
  ofr[:, feet, 0] = vals
  sim.create_graph()
  for _ in range(3):
    sim.step()
  ora.step(3)
  assert _rel(_np(sim.data.qvel), ora.qvel) < 5e-06
  assert _rel(_np(sim.data.qpos), ora.qpos) < 1e-06
  # and friction really matters: a world with different friction diverges from world-shared friction
  sim2, _, _ = _pair("g1_velocity_flat")
  for _ in range(3):
    sim2.step()
  assert _rel(_np(sim2.data.qvel), _np(sim.data.qvel)) > 1e-4


def test_bridge_contract():
  import torch

  sim, _, model = _pair("go1_velocity_flat", nworld=4)
  assert sim.data.nworld == 4
  assert sim.data.qpos.shape == (4, model.nq) and sim.data.xpos.shape == (4, model.nbody, 3)
  assert sim.data.xmat.shape == (4, model.nbody, 3, 3) and sim.data.geom_xmat.shape == (4, model.ngeom, 3, 3)
  assert sim.data.cvel.shape == (4, model.nbody, 6) and sim.data.time.shape == (4,)
  assert sim.model.geom_bodyid.shape == (model.ngeom,) and sim.model.jnt_range.shape == (4, model.njnt, 2)
  with pytest.raises(AttributeError, match="read-only"):
    sim.data.qpos = torch.zeros(1)
  with pytest.raises(AttributeError):
    sim.model.body_mass = torch.zeros(1)
  ptr = sim.data.qpos.data_ptr()
  sim.data.qpos[1:3, 0] = 0.5
  assert sim.data.qpos.data_ptr() == ptr and float(sim.data.qpos[2, 0]) == 0.5
  assert float((sim.data.qpos * 2.0)[1, 0]) == 1.0  # plain torch ops work


def test_rollout_driver_runs_and_resets():
  import torch

  from mjlab_amd import robots
  from mjlab_amd.rollout import PhysicsRollout, g1_action_scale
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model("g1_velocity_flat")
  sim = Simulation(256, SimulationCfg(njmax=300), model, "cuda:0")
  roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=1)
  nreset = 0
  for _ in range(60):
    nreset += int(roll.step(roll.random_action()).sum())
  torch.cuda.synchronize()
  assert torch.isfinite(sim.data.qpos).all()
  assert float(sim.data.qpos[:, 2].min()) > 0.2  # fallen robots were reset
  assert roll.observation_rows().shape == (256, 99)
  assert nreset > 0


def test_device_selftest_wave_primitives():
  import torch

  from mjlab_amd import native

  native.check(native.lib().mjlab_selftest(torch.cuda.current_stream().cuda_stream), "mjlab_selftest")


def test_every_domain_randomization_field_is_honoured_per_world():
  """All fields the reference can randomise per world (FIELD_SPECS, envs/mdp/events.py:184-209):
  expand, perturb each world differently, and compare with the oracle given the same per-world
  values -- i.e. the kernels really read `field + w * stride` for every one of them."""
  import torch

  sim, ora, model = _pair("g1_velocity_flat", nworld=6)
  fields = ["dof_armature", "dof_frictionloss", "dof_damping", "jnt_range", "jnt_stiffness", "body_mass", "body_ipos",
            "body_iquat", "body_inertia", "body_pos", "body_quat", "geom_friction", "geom_pos", "geom_quat", "geom_rgba",
            "site_pos", "site_quat", "qpos0"]  # fmt: skip
  sim.expand_model_fields(fields)
  rng = np.random.default_rng(17)
  n = sim.num_envs

  def unit(q):
    return q / np.linalg.norm(q, axis=-1, keepdims=True)

  for f in fields:
    base = np.asarray(getattr(model, f), dtype=np.float64)
    t = getattr(sim.model, f)
    assert t.shape == (n, *base.shape) and t.stride(0) == base.size, f
    if f == "geom_rgba":
      continue  # not consumed by the physics
    b = np.broadcast_to(base, (n, *base.shape)).copy()
    if f in ("body_iquat", "body_quat", "geom_quat", "site_quat"):
      new = unit(b + rng.normal(0, 0.02, b.shape))
      if f == "body_quat":
        new[:, :3] = b[:, :3]  # leave world / terrain / floating root frames alone
    elif f in ("body_mass", "body_inertia", "dof_armature"):
      new = b * rng.uniform(0.85, 1.15, b.shape)
    elif f in ("dof_damping", "jnt_stiffness"):
      new = b + rng.uniform(0.0, 0.5, b.shape)
      if f == "dof_damping":
        new[:, :6] = 0.0
      else:
        new[:, 0] = 0.0  # free joint
    elif f == "jnt_range":
      new = b + rng.uniform(-0.05, 0.05, b.shape)
    elif f == "dof_frictionloss":  # friction-loss rows on about half of the joint dofs, different ones per world
      new = rng.uniform(0.05, 0.4, b.shape) * (rng.random(b.shape) < 0.5)
      new[:, :6] = 0.0
    elif f == "geom_friction":
      new = b * rng.uniform(0.5, 1.5, b.shape)
    elif f == "qpos0":
      new = b.copy()
      new[:, 7:] += rng.normal(0, 0.02, (n, base.size - 7))
    else:  # positions
      new = b + rng.normal(0, 0.005, b.shape)
      if f == "body_pos":
        new[:, :3] = b[:, :3]
    t[:] = torch.from_numpy(new.astype(np.float32)).cuda()
    ora.expand_model_field(f)[:] = new
  sim.create_graph()
  sim.forward()
  ora.forward()
  assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  assert np.array_equal(_np(sim.data.nf).ravel(), ora.nf.ravel()) and len(set(ora.nf.ravel().tolist())) > 1
  for f in ("xpos", "xipos", "geom_xpos", "site_xpos", "subtree_com", "qM", "qfrc_bias", "qfrc_passive", "qfrc_smooth"):
    assert _rel(_np(getattr(sim.data, f)), getattr(ora, f)) < 1e-06, f
  # different worlds really got different models
  assert float(sim.data.qM[0].sub(sim.data.qM[1]).abs().max()) > 1e-3
  for _ in range(2):
    sim.step()
  ora.step(2)
  assert _rel(_np(sim.data.qpos), ora.qpos) < 1e-06
  assert _rel(_np(sim.data.qvel), ora.qvel) < 1e-05
  assert sim.data.act.shape == (n, 0)


@pytest.mark.parametrize("njmax", [300, 100])
def test_capacity_paths_with_hundreds_of_contacts(njmax):
  """A large geom margin turns most of the 502 candidate pairs into contacts: several 64-contact
  chunks in the constraint build, the sequential replay of the row-capacity rule, contact and
  row capacities reached, line-search rows beyond the register-cached ones."""
  import copy

  import torch

  from mjlab_amd.sim import Simulation, SimulationCfg

  model = copy.deepcopy(models()["g1_velocity_flat"])
  model.geom_margin = np.full_like(model.geom_margin, 0.25)
  nworld = 4
  qpos, qvel, ctrl = golden_inputs(model, nworld, 13)
  sim = Simulation(nworld, SimulationCfg(njmax=njmax, use_graph=False), model, "cuda:0")
  ora = OracleSim(model, nworld, njmax=njmax, precision="f64")
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    getattr(ora, f)[:] = v
  sim.forward()
  ora.forward()
  ncon, nefc = ora.ncon.ravel(), ora.nefc.ravel()
  assert ncon.max() >= min(129, sim.nconmax) and nefc.max() > min(njmax, 200) - 8, (ncon, nefc)  # the scenario does what it says
  assert np.array_equal(_np(sim.data.ncon).ravel(), ncon)
  assert np.array_equal(_np(sim.data.nefc).ravel(), nefc)
  assert _np(sim.data.nefc).max() <= njmax
  nv = model.nv
  for w in range(nworld):
    nc, n = int(ncon[w]), int(nefc[w])
    assert np.array_equal(_np(sim.data.contact_efc_address)[w, :nc], ora.contact_efc_address[w, :nc])
    assert np.array_equal(_np(sim.data.efc_type)[w, :n], ora.efc_type[w, :n])
    Jg = _np(sim.data.efc_J)[w].reshape(-1, nv)[:n]
    assert _rel(Jg, ora.efc_J[w].reshape(-1, nv)[:n]) < 2e-06
    for f in ("efc_D", "efc_aref", "efc_pos", "efc_margin"):
      assert _rel(_np(getattr(sim.data, f))[w, :n], getattr(ora, f)[w, :n]) < 2e-4, f
  assert np.array_equal(_np(sim.data.sensordata), ora.sensordata.astype(np.float32))
  # the solve itself: both sides stop at the 10-iteration cap on a 300-row problem, so only the
  # quality of the iterate is compared, not the iterate
  assert torch.isfinite(sim.data.qacc).all()
  assert _rel(_np(sim.data.qacc_smooth), ora.qacc_smooth) < 2e-05
  d = sim.data
  M, qa, qs = d.qM.double(), d.qacc.double(), d.qfrc_smooth.double()
  J = d.efc_J.view(nworld, njmax, nv).double()
  rows = torch.arange(njmax, device="cuda")[None, :] < d.nefc.view(-1, 1)
  jar = torch.where(rows, torch.einsum("wrv,wv->wr", J, qa) - d.efc_aref.double(), torch.zeros_like(d.efc_aref.double()))
  cost_gpu = 0.5 * torch.einsum("wi,wij,wj->w", qa - d.qacc_smooth.double(), M, qa - d.qacc_smooth.double()) + 0.5 * (
    torch.where(rows, d.efc_D.double(), torch.zeros_like(jar)) * jar.clamp_max(0) ** 2).sum(dim=1)
  # oracle's cost at its own iterate, same formula in numpy
  for w in range(nworld):
    n = int(nefc[w])
    Mo = ora.qM[w].reshape(nv, nv)
    da = ora.qacc[w] - ora.qacc_smooth[w]
    jo = np.minimum(ora.efc_J[w].reshape(-1, nv)[:n] @ ora.qacc[w] - ora.efc_aref[w, :n], 0)
    cost_o = 0.5 * da @ Mo @ da + 0.5 * np.sum(ora.efc_D[w, :n] * jo * jo)
    assert float(cost_gpu[w]) <= cost_o * 1.05 + 1e-6, (w, float(cost_gpu[w]), cost_o)


@pytest.mark.parametrize("variant", ["impratio", "direct_solref", "margin_gap", "solimp_power", "solimp_flat", "euler_damped", "springs", "frictionloss"])
def test_parameter_branches_match_oracle(variant):
  """Less-travelled branches of the constraint parameter code (impratio scaling of the pyramid
  regulariser, negative solref = direct stiffness/damping, geom margin/gap, solimp power != 2,
  Euler with implicit joint damping) on the model that has every primitive and joint type."""
  import copy

  import torch

  from mjlab_amd import mjcf
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = copy.deepcopy(models()["mixed"])
  if variant == "impratio":
    model.opt.impratio = 4.0
  elif variant == "direct_solref":
    model.geom_solref = np.tile(np.array([-800.0, -40.0]), (model.ngeom, 1))
    model.jnt_solref = np.tile(np.array([-500.0, -30.0]), (model.njnt, 1))
  elif variant == "margin_gap":
    model.geom_margin = np.full_like(model.geom_margin, 0.02)
    model.geom_gap = np.full_like(model.geom_gap, 0.005)
  elif variant == "solimp_power":
    model.geom_solimp = np.tile(np.array([0.8, 0.97, 0.01, 0.3, 3.0]), (model.ngeom, 1))
  elif variant == "solimp_flat":  # width 0: mj getimpedance's flat case, imp = 0.5 (dmin + dmax) (round 6)
    model.geom_solimp = np.tile(np.array([0.8, 0.95, 0.0, 0.5, 2.0]), (model.ngeom, 1))
  elif variant == "euler_damped":
    model.opt.integrator = mjcf.INT_EULER
    model.dof_damping = np.where(np.arange(model.nv) >= model.nv - 2, 0.8, 0.0)
  elif variant == "springs":  # joint springs pull towards mjModel.qpos_spring (springref), which is NOT qpos0
    hinge_or_slide = np.asarray(model.jnt_type) != mjcf.JNT_FREE
    model.jnt_stiffness = np.where(hinge_or_slide, 25.0, 0.0)
    model.qpos_spring = np.asarray(model.qpos_spring, dtype=np.float64).copy()
    model.qpos_spring[np.asarray(model.jnt_qposadr)[hinge_or_slide]] += 0.04
  elif variant == "frictionloss":  # friction-loss rows (Huber cost): free-body dofs, a brake that holds, one that slips
    fl = np.zeros(model.nv)
    fl[:6] = 0.05
    fl[-2:] = [3.0, 0.3]
    model.dof_frictionloss = fl
    model.dof_solref = np.asarray(model.dof_solref, dtype=np.float64).copy()
    model.dof_solimp = np.asarray(model.dof_solimp, dtype=np.float64).copy()
    model.dof_solref[-1] = [0.05, 0.7]
    model.dof_solimp[-2] = [0.8, 0.9, 0.001, 0.5, 2.0]
  nworld = 8
  qpos, qvel, ctrl = golden_inputs(model, nworld, 23)
  sim = Simulation(nworld, SimulationCfg(njmax=64, use_graph=False), model, "cuda:0")
  ora = OracleSim(model, nworld, njmax=64, precision="f64")
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    getattr(ora, f)[:] = v
  sim.forward()
  ora.forward()
  assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  assert ora.nefc.max() >= 3
  for w in range(nworld):
    n = int(ora.nefc[w, 0])
    for f in ("efc_D", "efc_aref", "efc_pos", "efc_margin"):
      assert _rel(_np(getattr(sim.data, f))[w, :n], getattr(ora, f)[w, :n]) < 2e-05, (variant, f)
    assert np.array_equal(_np(sim.data.efc_type)[w, :n], ora.efc_type[w, :n]) and np.array_equal(_np(sim.data.efc_id)[w, :n], ora.efc_id[w, :n])
  assert np.array_equal(_np(sim.data.nf).ravel(), ora.nf.ravel())
  if variant == "frictionloss":
    assert int(ora.nf[0, 0]) == 8
    assert np.array_equal(_np(sim.data.efc_frictionloss)[:, :8], ora.efc_frictionloss[:, :8].astype(np.float32))
    f_o = ora.efc_force[:, :8]
    assert _rel(_np(sim.data.efc_force)[:, :8], f_o) < 2e-05
    lim = ora.efc_frictionloss[:, :8]
    assert (np.abs(f_o) >= lim - 1e-12).any() and (np.abs(f_o) < 0.99 * lim).any()  # rows in the linear and in the quadratic zone
  assert _rel(_np(sim.data.qacc), ora.qacc) < 5e-06
  for _ in range(5):
    sim.step()
  ora.step(5)
  assert _rel(_np(sim.data.qpos), ora.qpos) < 1e-06
  assert _rel(_np(sim.data.qvel), ora.qvel) < 5e-06


def test_friction_loss_on_every_joint_of_the_g1_tracks_the_oracle():
  """dof_frictionloss on all 29 joints (what randomize_field("dof_frictionloss") produces): 29 friction-loss rows ahead of
  the limit and contact rows, rows in both zones, over a short rollout; and with a row capacity too small for them."""
  import copy

  import torch

  from mjlab_amd.sim import Simulation, SimulationCfg

  model = copy.deepcopy(models()["g1_velocity_flat"])
  rng = np.random.default_rng(5)
  fl = np.zeros(model.nv)
  fl[6:] = rng.uniform(0.05, 2.0, model.nv - 6)
  model.dof_frictionloss = fl
  nworld = 16
  qpos, qvel, ctrl = golden_inputs(model, nworld, 31)
  sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
  ora = OracleSim(model, nworld, njmax=300, precision="f64")
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    getattr(ora, f)[:] = v
  sim.forward()
  ora.forward()
  assert (ora.nf == 29).all() and np.array_equal(_np(sim.data.nf).ravel(), ora.nf.ravel()) and np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  nv = model.nv
  for w in range(nworld):
    n = int(ora.nefc[w, 0])
    assert np.array_equal(_np(sim.data.efc_type)[w, :n], ora.efc_type[w, :n])
    assert _rel(_np(sim.data.efc_J)[w].reshape(-1, nv)[:n], ora.efc_J[w].reshape(-1, nv)[:n]) < 5e-06
    for f in ("efc_D", "efc_aref"):
      assert _rel(_np(getattr(sim.data, f))[w, :n], getattr(ora, f)[w, :n]) < 1e-04, f
  f_o, lim = ora.efc_force[:, :29], ora.efc_frictionloss[:, :29]
  assert (np.abs(f_o) >= lim - 1e-12).mean() > 0.05 and (np.abs(f_o) < 0.99 * lim).mean() > 0.05
  assert _rel(_np(sim.data.qacc), ora.qacc) < 2e-05
  assert _rel(_np(sim.data.efc_force)[:, :29], f_o) < 5e-05
  assert _rel(_np(sim.data.qfrc_constraint), ora.qfrc_constraint) < 2e-05
  # Rollout.  With 29 more rows some worlds end their Newton iteration at the cap of 10, where the iterate depends on
  # rounding: the fp32 and the fp64 build of the restatement themselves differ by 7e-5 in qpos after these 10 steps
  # (2e-6 without the friction-loss rows), so the device is held to the fp32 build tightly and to fp64 at that level.
  ora32 = OracleSim(model, nworld, njmax=300, precision="f32")
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(ora32, f)[:] = v
  for _ in range(10):
    sim.step()
  ora.step(10)
  ora32.step(10)
  assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  assert _rel(_np(sim.data.qpos), ora32.qpos) < 2e-06
  assert _rel(_np(sim.data.qvel), ora32.qvel) < 1e-05
  assert _rel(_np(sim.data.qpos), ora.qpos) < 2e-04
  assert _rel(_np(sim.data.qvel), ora.qvel) < 2e-03
  # capacity: 29 friction-loss rows into njmax = 20 -> the first 20 dofs get theirs, the flag is raised, both sides agree
  sim = Simulation(2, SimulationCfg(njmax=20, use_graph=False), model, "cuda:0")
  ora = OracleSim(model, 2, njmax=20, precision="f64")
  for f, v in (("qpos", qpos[:2]), ("qvel", qvel[:2]), ("ctrl", ctrl[:2])):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    getattr(ora, f)[:] = v
  sim.forward()
  ora.forward()
  from mjlab_amd import _abi

  assert (ora.nf == 20).all() and np.array_equal(_np(sim.data.nf).ravel(), ora.nf.ravel()) and np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  assert np.array_equal(_np(sim.data.efc_id)[:, :20], ora.efc_id[:, :20])
  assert (_np(sim.data.overflow).ravel() & _abi.OVF_NJMAX).all() and (ora.overflow.ravel() & _abi.OVF_NJMAX).all()
  assert _rel(_np(sim.data.qacc), ora.qacc) < 2e-05


@pytest.mark.parametrize("name,iterations", [("mixed", 100), ("g1_velocity_flat", 10), ("g1_velocity_flat", 100), ("go1_velocity_flat", 50)])
def test_cg_solver_tracks_the_oracle_and_converges_to_newton(name, iterations):
  """MujocoCfg(solver="cg") (reference sim/sim.py:56): the Polak-Ribiere CG of mj_solPrimal, preconditioned by M.
  Device vs the restatement's CG on the same states and over a short rollout (fp32 and fp64 take different numbers of
  iterations near the tolerance, so the iterate is compared where both converged and the COST everywhere), and CG with
  enough iterations lands on the Newton solution."""
  import copy

  import torch

  from mjlab_amd import mjcf
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = copy.deepcopy(models()[name])
  model.opt.solver = mjcf.SOL_CG
  model.opt.iterations = iterations
  nworld = 16
  qpos, qvel, ctrl = golden_inputs(model, nworld, 41)
  sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
  ora = OracleSim(model, nworld, njmax=300, precision="f64")
  ora32 = OracleSim(model, nworld, njmax=300, precision="f32")
  for o in (ora, ora32):
    for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
      getattr(o, f)[:] = v
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
  sim.data.qacc_warmstart.zero_()  # the constructor's forward() left its own solution there; a capped CG depends on where it starts
  sim.forward()
  ora.forward()
  ora32.forward()
  it_g, it_o = _np(sim.data.solver_niter).ravel(), ora.solver_niter.ravel()
  assert it_g.max() <= iterations and (it_g > 0).any()
  if iterations >= 50:  # converged on both sides: same solution, and it is Newton's
    # linear convergence + the fp32 noise floor of the termination test: the fp32 build of the restatement is itself
    # 1e-3 .. 2e-3 from the fp64 one here (Newton: 1e-5)
    assert _rel(_np(sim.data.qacc), ora.qacc) < 1e-02
    newton = copy.deepcopy(model)
    newton.opt.solver = mjcf.SOL_NEWTON
    on = OracleSim(newton, nworld, njmax=300, precision="f64")
    for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
      getattr(on, f)[:] = v
    on.forward()
    assert _rel(_np(sim.data.qacc), on.qacc) < 1e-02
    assert abs(float(np.median(it_g)) - float(np.median(ora32.solver_niter))) <= max(6, iterations // 5)
  else:  # both sides stop at the cap in most worlds: same iterate as the fp32 build of the restatement
    assert (it_o == iterations).mean() > 0.3
    assert _rel(_np(sim.data.qacc), ora32.qacc) < 2e-03  # 6e-5 after one iteration, growing along the CG path
    return  # ten CG iterations leave the contacts unresolved: a rollout from there amplifies any difference
  for _ in range(5):
    sim.step()
  ora.step(5)
  ora32.step(5)
  assert _rel(_np(sim.data.qpos), ora32.qpos) < 3e-03  # five steps of a solver that stops on an fp32 noise floor: 1.2e-3 measured
  assert _rel(_np(sim.data.qpos), ora.qpos) < 2e-03


@pytest.mark.parametrize("solver", ["newton", "cg"])
def test_every_solver_instantiation_with_friction_loss_and_hundreds_of_rows(solver):
  """The four instantiations of the solver code (row arrays in LDS / in global memory for worlds with more rows than LDS
  holds, Newton / CG) with friction-loss rows present: a large geom margin gives 150-300 rows per world (the global-memory
  instantiation), the unmodified margin the usual 20-60.  Compared with the fp32 build of the restatement by the primal
  cost reached (the iterates of capped solves differ) and, for Newton, by the iterate."""
  import copy

  import torch

  from mjlab_amd import mjcf
  from mjlab_amd.sim import Simulation, SimulationCfg

  for margin in (None, 0.25):
    model = copy.deepcopy(models()["g1_velocity_flat"])
    if margin is not None:
      model.geom_margin = np.full_like(model.geom_margin, margin)
    fl = np.zeros(model.nv)
    fl[6:] = np.linspace(0.05, 1.0, model.nv - 6)
    model.dof_frictionloss = fl
    model.opt.solver = mjcf.SOL_CG if solver == "cg" else mjcf.SOL_NEWTON
    model.opt.iterations = 60 if solver == "cg" else 10
    nworld = 8
    qpos, qvel, ctrl = golden_inputs(model, nworld, 13)
    sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
    ora = OracleSim(model, nworld, njmax=300, precision="f32")
    for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
      getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
      getattr(ora, f)[:] = v
    sim.data.qacc_warmstart.zero_()
    sim.forward()
    ora.forward()
    nefc = ora.nefc.ravel()
    assert np.array_equal(_np(sim.data.nefc).ravel(), nefc) and np.array_equal(_np(sim.data.nf).ravel(), ora.nf.ravel())
    assert (nefc > 128).all() if margin is not None else (nefc <= 128).sum() >= 4  # both instantiations, also within one launch
    assert torch.isfinite(sim.data.qacc).all()
    nv, nf = model.nv, int(ora.nf[0, 0])

    def cost(qacc, w):  # primal cost with the Huber zones of the friction-loss rows, from the oracle's own arrays (fp64 arithmetic)
      n = int(nefc[w])
      M, J = ora.qM[w].reshape(nv, nv).astype(np.float64), ora.efc_J[w].reshape(-1, nv)[:n].astype(np.float64)
      D, fl_ = ora.efc_D[w, :n].astype(np.float64), ora.efc_frictionloss[w, :n].astype(np.float64)
      da = qacc.astype(np.float64) - ora.qacc_smooth[w]
      x = J @ qacc.astype(np.float64) - ora.efc_aref[w, :n]
      c = 0.5 * da @ M @ da
      for r in range(n):
        if r < nf:
          rf = fl_[r] / D[r]
          c += 0.5 * D[r] * x[r] ** 2 if abs(x[r]) < rf else fl_[r] * (abs(x[r]) - 0.5 * rf)
        elif x[r] < 0:
          c += 0.5 * D[r] * x[r] ** 2
      return c

    g = _np(sim.data.qacc)
    for w in range(nworld):
      cg_, co_ = cost(g[w], w), cost(ora.qacc[w], w)
      assert cg_ <= co_ * 1.02 + 1e-6, (solver, margin, w, cg_, co_)
    if solver == "newton" and margin is None:
      assert _rel(g, ora.qacc) < 1e-04


MULTI_JOINT_XML = """
<mujoco model="multi_joint">
  <compiler angle="radian" autolimits="true"/>
  <option timestep="0.002"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 0.01"/>
    <body name="cart" pos="0 0 0.3">
      <inertial pos="0 0 0" mass="2" diaginertia="0.02 0.02 0.02"/>
      <joint name="sx" type="slide" axis="1 0 0"/>
      <joint name="sy" type="slide" axis="0 1 0"/>
      <geom name="cart_geom" type="sphere" size="0.1"/>
      <body name="pole" pos="0 0 0.1" quat="0.9689124 0.2474040 0 0">
        <inertial pos="0 0 0.3" mass="0.5" diaginertia="0.02 0.02 0.001"/>
        <joint name="rx" type="hinge" axis="1 0 0" pos="0 0 0"/>
        <joint name="ry" type="hinge" axis="0 1 0" pos="0 0 0.05" range="-0.4 0.4"/>
        <geom name="pole_geom" type="capsule" size="0.03" fromto="0 0 0.05 0 0 0.6"/>
        <body name="tip" pos="0 0 0.6">
          <inertial pos="0 0 0" mass="0.2" diaginertia="0.001 0.001 0.001"/>
          <joint name="tz" type="slide" axis="0 0 1" range="-0.1 0.1"/>
          <joint name="rz" type="hinge" axis="0 0 1"/>
          <geom name="tip_geom" type="sphere" size="0.05"/>
        </body>
      </body>
    </body>
  </worldbody>
</mujoco>
"""


def test_bodies_with_several_joints():
  """Bodies carrying two joints each (slide+slide, hinge+hinge with different anchors, slide+hinge):
  the per-body joint loops of the kinematics and the chain sums of the velocity stage."""
  import torch

  from mjlab_amd import mjcf
  from mjlab_amd.sim import Simulation, SimulationCfg

  spec = mjcf.Spec.from_string(MULTI_JOINT_XML)
  spec.option.integrator = mjcf.INT_IMPLICITFAST
  model = spec.compile()
  assert model.nv == 6 and list(model.body_jntnum) == [0, 2, 2, 2]
  nw = 64
  rng = np.random.default_rng(0)
  qpos = rng.normal(scale=0.6, size=(nw, model.nq))
  qpos[:, 2] = rng.uniform(-3.0, 3.0, nw)  # swing the pole all the way round: it meets the floor
  qvel = rng.normal(scale=1.0, size=(nw, model.nv))
  sim = Simulation(nw, SimulationCfg(), model, "cuda:0")
  ora = OracleSim(model, nw)
  sim.data.qpos[:] = torch.from_numpy(qpos.astype(np.float32)).cuda()
  sim.data.qvel[:] = torch.from_numpy(qvel.astype(np.float32)).cuda()
  ora.qpos[:], ora.qvel[:] = qpos, qvel
  sim.forward()
  ora.forward()
  assert ora.ncon.sum() > 10 and ora.nefc.max() > 4
  assert np.array_equal(_np(sim.data.ncon).ravel(), ora.ncon.ravel())
  assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  for f in KIN:
    assert _rel(_np(getattr(sim.data, f)), getattr(ora, f)) < 1e-06, f
  for f in VEL:
    assert _rel(_np(getattr(sim.data, f)), getattr(ora, f)) < 1e-06, f
  assert _rel(_np(sim.data.qacc_smooth), ora.qacc_smooth) < 5e-06
  assert _rel(_np(sim.data.qacc), ora.qacc) < 2e-05
  for _ in range(20):
    sim.step()
  ora.step(20)
  assert _rel(_np(sim.data.qpos), ora.qpos) < 5e-06
  assert _rel(_np(sim.data.qvel), ora.qvel) < 1e-05


def test_more_geoms_sites_and_actuators_than_lanes():
  """More than 128 moving geoms, more than 64 sites and more than 64 actuators: the rounds beyond
  what the stage kernels pre-load into registers (in-place loads) give the same results."""
  import torch

  from mjlab_amd import mjcf
  from mjlab_amd.mjcf import SpecActuator
  from mjlab_amd.sim import Simulation, SimulationCfg

  spec = mjcf.Spec.from_string(MULTI_JOINT_XML)
  spec.option.integrator = mjcf.INT_IMPLICITFAST
  rng = np.random.default_rng(3)
  bodies = [spec.body(n) for n in ("cart", "pole", "tip")]
  for i in range(150):  # decoration geoms: never collide, but their poses are outputs
    q = rng.normal(size=4)
    spec.add_geom(bodies[i % 3], f"deco_{i}", mjcf.GEOM_SPHERE, (0.01,), pos=rng.normal(scale=0.2, size=3), quat=q / np.linalg.norm(q), contype=0, conaffinity=0)
  for i in range(70):
    q = rng.normal(size=4)
    spec.add_site(bodies[i % 3], f"site_{i}", pos=rng.normal(scale=0.2, size=3), quat=q / np.linalg.norm(q))
  joints = ["sx", "sy", "rx", "ry", "tz", "rz"]
  for i in range(70):  # several actuators per joint
    lim = i % 3 != 0
    spec.actuators.append(SpecActuator(name=f"act_{i}", joint=joints[i % 6], gainprm0=1.0 + 0.1 * i, biasprm=(0.01 * i, -(1.0 + 0.1 * i), -0.05),
                                       forcerange=(-2.0, 2.0) if lim else None, ctrlrange=(-0.5, 0.5) if i % 2 else None, gear=1.0 + 0.01 * i))
  model = spec.compile()
  assert model.ngeom - model.nstaticgeom > 128 and model.nsite > 64 and model.nu > 64
  nw = 16
  qpos = rng.normal(scale=0.4, size=(nw, model.nq))
  qvel = rng.normal(scale=0.5, size=(nw, model.nv))
  ctrl = rng.normal(scale=0.6, size=(nw, model.nu))
  sim = Simulation(nw, SimulationCfg(), model, "cuda:0")
  ora = OracleSim(model, nw)
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    getattr(ora, f)[:] = v
  sim.forward()
  ora.forward()
  for f in ("geom_xpos", "geom_xmat", "site_xpos", "site_xmat"):
    assert _rel(_np(getattr(sim.data, f)), getattr(ora, f)) < 2e-6, f
  for f in ("actuator_force", "qfrc_actuator", "qfrc_smooth"):
    assert _rel(_np(getattr(sim.data, f)), getattr(ora, f)) < 1e-06, f
  assert _rel(_np(sim.data.qacc), ora.qacc) < 5e-05


def test_ls_parallel_is_executed_and_matches_the_restatement(monkeypatch):
  """SimulationCfg.ls_parallel=True (the reference's default, sim/sim.py:89,111) runs mujoco_warp's parallel grid search --
  `ls_iterations` log-spaced steps, lowest cost wins -- on the device; the CPU restatement carries the same rule
  (OracleSim(ls_parallel=True)).  A grid search moves every iterate by one of 20 discrete step sizes, so where fp32 and fp64
  pick different candidates the iterates part; compared are (a) the first Newton iterate from identical inputs (iterations = 1:
  same candidate in every world or the test says how many differ) and (b) the converged solve, which both searches bring to the
  same minimiser within the iteration cap in most worlds."""
  import torch

  from mjlab_amd.sim import Simulation, SimulationCfg

  import copy

  base = models()["g1_velocity_flat"]
  nworld = 64
  qpos, qvel, ctrl = golden_inputs(base, nworld, 5)
  res = {}
  for iters in (1, 10):
    model = copy.deepcopy(base)
    model.opt.iterations = iters
    for par in (True, False):
      sim = Simulation(nworld, SimulationCfg(njmax=300, ls_parallel=par, use_graph=False), model, "cuda:0")
      assert sim.ls_parallel == par
      ora = OracleSim(model, nworld, njmax=300, precision="f64", ls_parallel=par)
      for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
        getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
        getattr(ora, f)[:] = v
      sim.forward()
      ora.forward(nthread=8)
      g, o = _np(sim.data.qacc).astype(np.float64), ora.qacc
      err = np.abs(g - o).max(axis=1) / np.maximum(np.abs(o).max(axis=1), 1e-6)
      res[(iters, par)] = (err, _np(sim.data.solver_niter).ravel().copy(), ora.solver_niter.ravel().copy(), g)
  e1p, n1g, n1o, q1p = res[(1, True)]
  e1e, _, _, q1e = res[(1, False)]
  # the grid search really ran: after one iteration its iterate differs from the exact search's in most worlds
  assert (np.abs(q1p - q1e).max(axis=1) > 1e-4 * np.abs(q1e).max(axis=1)).mean() > 0.5
  # (a single Newton step from a cold start is not compared with the restatement: it carries the fp32 rounding of the first search
  # direction, 1e-3 and more in ill-conditioned worlds under EITHER search; the converged solve below is what parity means)
  print(f"ls_parallel, 1 iteration: qacc err vs restatement median {np.median(e1p):.2e} (exact search: median {np.median(e1e):.2e})")
  e10p, n10g, n10o, _ = res[(10, True)]
  print(f"ls_parallel, 10 iterations: qacc err median {np.median(e10p):.2e} p90 {np.percentile(e10p, 90):.2e} max {e10p.max():.2e}; "
        f"iterations gpu {n10g.mean():.2f} oracle {n10o.mean():.2f} (exact search: {res[(10, False)][1].mean():.2f})")
  assert np.median(e10p) < 2e-5 and np.percentile(e10p, 90) < 1e-3
  assert abs(n10g.mean() - n10o.mean()) < 1.0


def test_local_frame_keeps_full_precision_far_from_the_origin():
  """The stages compute positions in the world's local frame (xorigin = the floating base's position rounded to whole
  metres; include/mjlab_fields.h): a robot standing 100 m out -- where the reference's environments put most of their 4096
  robots (env_spacing x a 64 x 64 grid) and where fp32 world coordinates resolve 7.6 um -- is solved as accurately as one at
  the origin.  Both sides get the SAME fp32-rounded state; the restatement is fp64 in plain world coordinates."""
  import torch

  from mjlab_amd.sim import Simulation, SimulationCfg

  for name in ("g1_velocity_flat", "go1_velocity_flat"):
    model = models()[name]
    nworld = 64
    qpos, qvel, ctrl = golden_inputs(model, nworld, 7)
    errs = {}
    for shift in ((0.0, 0.0), (103.37, -87.81)):
      q = qpos.copy()
      q[:, 0] += shift[0]
      q[:, 1] += shift[1]
      q32 = q.astype(np.float32)
      sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
      ora = OracleSim(model, nworld, njmax=300, precision="f64")
      for f, v in (("qpos", q32), ("qvel", qvel.astype(np.float32)), ("ctrl", ctrl.astype(np.float32))):
        getattr(sim.data, f)[:] = torch.from_numpy(v).cuda()
        getattr(ora, f)[:] = v.astype(np.float64)
      sim.forward()
      ora.forward(nthread=8)
      assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel()) and np.array_equal(_np(sim.data.ncon).ravel(), ora.ncon.ravel())
      org = _np(sim.data.xorigin)
      assert np.array_equal(org, np.rint(q32[:, :3])), "xorigin = the floating base's position rounded to whole metres"
      per_world = lambda a, b: np.abs(a.reshape(nworld, -1).astype(np.float64) - b.reshape(nworld, -1)).max(axis=1) / np.maximum(np.abs(b.reshape(nworld, -1)).max(axis=1), 1e-6)  # noqa: E731
      nefc = ora.nefc.ravel()

      def rows(a, width):  # row arrays: only the first nefc rows of a world are defined
        a = np.asarray(a, np.float64).reshape(nworld, -1, width).copy()
        for w_ in range(nworld):
          a[w_, nefc[w_] :] = 0.0
        return a

      errs[shift] = {f: per_world(_np(getattr(sim.data, f)), getattr(ora, f)) for f in ("qacc", "qacc_smooth", "qM", "qfrc_bias")}
      errs[shift]["efc_aref"] = per_world(rows(_np(sim.data.efc_aref), 1), rows(ora.efc_aref, 1))
      errs[shift]["efc_J"] = per_world(rows(_np(sim.data.efc_J), model.nv), rows(ora.efc_J, model.nv))
      # the public arrays are world coordinates (one rounding at 100 m: 4e-8 relative)
      for f in ("xpos", "xipos", "geom_xpos", "subtree_com", "site_xpos", "xanchor"):
        assert _rel(_np(getattr(sim.data, f)), getattr(ora, f)) < 1e-6, f
      ncon = ora.ncon.ravel()
      cp_g, cp_o = _np(sim.data.contact_pos).reshape(nworld, -1, 3), ora.contact_pos.reshape(nworld, -1, 3)
      for w in range(nworld):
        assert np.abs(cp_g[w, : ncon[w]] - cp_o[w, : ncon[w]]).max(initial=0.0) < 1e-5
    near, far = errs[(0.0, 0.0)], errs[(103.37, -87.81)]
    for f in near:
      print(f"{name} {f:12s} median / max relative error per world: at the origin {np.median(near[f]):.2e} / {near[f].max():.2e}, 100 m out {np.median(far[f]):.2e} / {far[f].max():.2e}")
      assert np.median(far[f]) < 2.0 * np.median(near[f]) + 1e-7, f
      assert far[f].max() < 3.0 * near[f].max() + 1e-6, f


@pytest.mark.parametrize("name", ["mixed", "go1_velocity_flat", "g1_velocity_flat"])
def test_pgs_solver_tracks_the_restatement_and_converges_to_newton(name):
  """MujocoCfg(solver="pgs") (reference sim/sim.py:56; north_star: "the PGS/Newton constraint solver"): mj_solPGS with scalar rows
  as a stage kernel (stage_pgs.h; Simulation switches to one kernel per stage for it).  Device vs the restatement's PGS after a
  few sweeps from the same warm start (Gauss-Seidel in fp32 follows the fp64 path row by row: 1e-4), identical force bounds, and
  with enough sweeps both land on the Newton solution (the dual optimum is the primal one).  Then a short rollout."""
  import copy

  import torch

  from mjlab_amd import mjcf
  from mjlab_amd.sim import Simulation, SimulationCfg

  base = copy.deepcopy(models()[name])
  if name == "go1_velocity_flat":  # friction-loss rows (two-sided bounds) in the mix
    base.dof_frictionloss = np.asarray(base.dof_frictionloss, dtype=np.float64).copy()
    base.dof_frictionloss[6:] = 0.2
  nworld = 16
  qpos, qvel, ctrl = golden_inputs(base, nworld, 41)
  per_world = lambda a, b: np.abs(a.astype(np.float64) - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-6)  # noqa: E731
  for iterations in (4, 600):
    model = copy.deepcopy(base)
    model.opt.solver, model.opt.iterations = mjcf.SOL_PGS, iterations
    sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
    assert sim.fuse == "stage"
    ora = OracleSim(model, nworld, njmax=300, precision="f64", flags=_abi_flags_frictionloss())
    for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
      getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
      getattr(ora, f)[:] = v.astype(np.float32)
    sim.data.qacc_warmstart.zero_()  # the constructor's forward() left its own solution there
    sim.forward()
    ora.forward(nthread=8)
    assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
    it_g, it_o = _np(sim.data.solver_niter).ravel(), ora.solver_niter.ravel()
    assert it_g.max() <= iterations and (it_g[ora.nefc.ravel() > 0] >= 1).all()
    fg = _np(sim.data.efc_force)
    for w in range(nworld):
      n, nf = int(ora.nefc[w, 0]), int(ora.nf[w, 0])
      assert (fg[w, nf:n] >= 0).all() and (np.abs(fg[w, :nf]) <= _np(sim.data.efc_frictionloss)[w, :nf] + 1e-6).all()
    err = per_world(_np(sim.data.qacc), ora.qacc)
    if iterations == 4:
      assert np.abs(it_g - it_o).max() <= 1
      assert err.max() < 5e-4, err  # four sweeps of the same row-by-row path in fp32 (the fp32 build of the restatement: 1.4e-5)
    else:
      newton = copy.deepcopy(base)
      newton.opt.iterations = 100
      on = OracleSim(newton, nworld, njmax=300, precision="f64", flags=_abi_flags_frictionloss())
      for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
        getattr(on, f)[:] = v.astype(np.float32)
      on.forward(nthread=8)
      print(f"{name}: PGS, {iterations} sweeps: device vs restatement median {np.median(err):.2e} max {err.max():.2e}; device vs Newton max {per_world(_np(sim.data.qacc), on.qacc).max():.2e}; "
            f"sweeps device {it_g.mean():.1f} restatement {it_o.mean():.1f}")
      # a linearly converging method stopped by an improvement test that fp32 resolves to ~1e-6 of the cost
      assert np.median(per_world(_np(sim.data.qacc), on.qacc)) < 2e-3 and per_world(_np(sim.data.qacc), on.qacc).max() < 5e-2
      for _ in range(5):
        sim.step()
      ora.step(5, nthread=8)
      assert np.isfinite(_np(sim.data.qpos)).all()
      assert np.median(per_world(_np(sim.data.qpos), ora.qpos)) < 1e-3


def _abi_flags_frictionloss():
  from mjlab_amd import _abi

  return _abi.OPT_FRICTIONLOSS


def test_nan_guard_dumps_the_device_ring(tmp_path):
  """reference sim/sim.py:129,191 + utils/nan_guard.py: cfg fields, `sim.nan_guard.watch`, one dump with the
  pre-step states of the last steps; the history ring stays on the device until the dump."""
  import torch

  from mjlab_amd.sim import NanGuardCfg, Simulation, SimulationCfg

  model = models()["go1_velocity_flat"]
  cfg = SimulationCfg(njmax=100, ls_parallel=False, nan_guard=NanGuardCfg(enabled=True, buffer_size=4, output_dir=str(tmp_path), max_envs_to_capture=3))
  sim = Simulation(8, cfg, model, "cuda:0")
  assert sim.nan_guard.enabled and sim.nan_guard._ring is None
  sim.data.qpos[:] = torch.tensor(model.key_qpos[0], dtype=torch.float32, device="cuda")
  for _ in range(6):
    sim.step()
  assert sim.nan_guard._ring.is_cuda and not list(tmp_path.iterdir())
  sim.data.qvel[5, 2] = float("nan")
  sim.step()
  dumps = sorted(tmp_path.glob("nan_dump_*.npz"))
  assert len(dumps) == 1
  z = np.load(dumps[0], allow_pickle=True)
  meta = z["_metadata"].item()
  assert 5 in meta["nan_env_ids"] and meta["num_envs_captured"] == 3 and meta["state_size"] == model.nq + model.nv
  keys = sorted(k for k in z.files if k.startswith("states_step_"))
  assert keys == [f"states_step_{i:06d}" for i in (3, 4, 5, 6)]
  assert z[keys[-1]].shape == (3, model.nq + model.nv) and np.isfinite(z[keys[-1]]).all()
  sim.step()  # one dump per run
  assert len(list(tmp_path.glob("nan_dump_*.npz"))) == 1
  assert (tmp_path / meta["model_file"]).exists()
  # disabled guard: nothing allocated, watch() is a no-op context
  sim2 = Simulation(2, SimulationCfg(njmax=100, ls_parallel=False), model, "cuda:0")
  with sim2.nan_guard.watch(sim2.data):
    pass
  assert not sim2.nan_guard.enabled


def _snake_model(nlink: int):
  """Free base + a chain of `nlink` hinge links (alternating axes, limited, PD-actuated) lying on the plane:
  nv = 6 + nlink, so that every padded size the solve / substep / control kernels are instantiated for is hit."""
  from mjlab_amd import mjcf
  from mjlab_amd.robots import ActuatorCfg, apply_actuators

  body = ""
  for i in range(nlink):
    axis = "0 1 0" if i % 2 else "0 0 1"
    body += (f'<body name="l{i}" pos="0.08 0 0"><inertial pos="0.04 0 0" mass="0.2" diaginertia="0.0002 0.0006 0.0006"/>'
             f'<joint name="j{i}" type="hinge" axis="{axis}" range="-0.6 0.6" damping="0.05"/>'
             f'<geom name="g{i}" type="capsule" size="0.025" fromto="0 0 0 0.08 0 0" contype="1" conaffinity="0"/>')
  xml = (f'<mujoco model="snake"><compiler angle="radian" autolimits="true"/><option timestep="0.004"/><worldbody>'
         f'<geom name="floor" type="plane" size="0 0 0.01" contype="0" conaffinity="1"/>'
         f'<body name="base" pos="0 0 0.03"><inertial pos="0 0 0" mass="1" diaginertia="0.004 0.004 0.004"/><freejoint name="root"/>'
         f'<geom name="gb" type="sphere" size="0.03" contype="1" conaffinity="0"/>{body}{"</body>" * nlink}</body></worldbody></mujoco>')
  spec = mjcf.Spec.from_string(xml)
  spec.option.integrator = mjcf.INT_IMPLICITFAST
  apply_actuators(spec, (ActuatorCfg([f"j{i}" for i in range(nlink)], effort_limit=5.0, stiffness=8.0, damping=0.4, armature=0.005),))
  return spec.compile()


@pytest.mark.parametrize("nlink,nvp", [(2, 8), (9, 16), (13, 20), (17, 24), (25, 32), (29, 36), (33, 40), (41, 48), (57, 64)])
def test_every_padded_size_of_the_solve_kernels(nlink, nvp):
  """One model per template instantiation (nv = 6 + nlink -> NVP), through all three launch structures that
  carry the solve stage: forward + steps vs the oracle, and the fused kernels bit-identical to the per-stage ones."""
  import torch

  from mjlab_amd.csrc_sizes import solve_nvp
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = _snake_model(nlink)
  assert solve_nvp(model.nv) == nvp
  nw = 16
  rng = np.random.default_rng(nlink)
  qpos = np.tile(model.qpos0, (nw, 1))
  # chains beyond ~45 links lying on the floor give an fp32 Newton Hessian the fp32 build of the restatement cannot
  # factor either (cond(M) 1e5 x contact stiffness): the long ones hover, their rows are joint limits (yaw beyond +-0.6)
  lifted = nlink > 40
  qpos[:, 2] = (0.3 if lifted else 0.024) + rng.uniform(0, 0.01, nw)
  amp = np.where(np.arange(nlink) % 2, 0.01, 0.66 if lifted else 0.3)  # yaw joints bend the chain in the plane, pitch joints barely lift it
  qpos[:, 7:] = rng.uniform(-1, 1, (nw, nlink)) * amp
  qvel = rng.normal(0, 0.3, (nw, model.nv))
  ctrl = rng.uniform(-0.4, 0.4, (nw, model.nu))
  ora = OracleSim(model, nw, njmax=480)
  ora.qpos[:], ora.qvel[:], ora.ctrl[:] = qpos, qvel, ctrl
  ora.forward()
  out = {}
  for fuse in ("stage", "step"):
    sim = Simulation(nw, SimulationCfg(njmax=480, fuse=fuse), model, "cuda:0")
    for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
      getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    sim.forward()
    if fuse == "stage":
      assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel()) and ora.nefc.mean() > (1 if lifted else 4) and ora.nefc.max() < 480
      assert _rel(_np(sim.data.qM), ora.qM) < 2e-6
      assert _rel(_np(sim.data.qacc), ora.qacc) < (2e-4 if nlink < 30 else 5e-3)  # long chains: cond(M) 1e4 .. 1e5
    sim.step()
    sim.step(3)
    out[fuse] = {f: _np(getattr(sim.data, f)).copy() for f in ("qpos", "qvel", "qacc", "efc_force", "xpos")}
  ora.step(4)
  assert _rel(out["stage"]["qpos"], ora.qpos) < (2e-5 if nlink < 30 else 2e-4) and _rel(out["stage"]["qvel"], ora.qvel) < (2e-3 if nlink < 30 else 2e-2)
  for f in out["stage"]:
    assert np.array_equal(out["stage"][f], out["step"][f]), f
