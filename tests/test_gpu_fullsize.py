"""Properties of the HIP path at BASELINE.json's full size (4096 worlds) that need no oracle run
of that size: determinism, batch / permutation independence, the equations of motion and the
KKT structure of the constraint solve evaluated on the device's own outputs, momentum in free
flight, isolation of a diverged world, stage-split equivalence.  All through the Simulation
boundary / C ABI.
"""

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))

pytestmark = pytest.mark.gpu

NWORLD = 4096


def _rollout_state(name, nworld=NWORLD, seed=3, steps=12):
  """A spread of realistic states: keyframe resets + `steps` control steps of random actions."""
  import torch

  from mjlab_amd import robots
  from mjlab_amd.rollout import PhysicsRollout, g1_action_scale, go1_action_scale
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model(name)
  sim = Simulation(nworld, SimulationCfg(njmax=300 if "velocity" in name else 250), model, "cuda:0")
  scale = g1_action_scale(model) if name.startswith("g1") else go1_action_scale(model)
  roll = PhysicsRollout(sim, action_scale=scale, seed=seed, min_height=0.3 if name.startswith("g1") else 0.15)
  for _ in range(steps):
    roll.step(roll.random_action())
  torch.cuda.synchronize()
  return sim, roll, model


def _clone_into(dst, src, fields=("qpos", "qvel", "ctrl", "qacc_warmstart", "qfrc_applied", "xfrc_applied", "time"), index=None):
  for f in fields:
    v = getattr(src.data, f)
    getattr(dst.data, f)[:] = v if index is None else v[index]


def _fresh(model, nworld, njmax=300, graph=True):
  from mjlab_amd.sim import Simulation, SimulationCfg

  return Simulation(nworld, SimulationCfg(njmax=njmax, use_graph=graph), model, "cuda:0")


OUT = ("qpos", "qvel", "qacc", "xpos", "xquat", "cvel", "subtree_com", "sensordata", "actuator_force", "qfrc_constraint")


def test_deterministic_and_batch_independent():
  import torch

  sim, _, model = _rollout_state("g1_velocity_flat")
  a, b = _fresh(model, NWORLD), _fresh(model, NWORLD)
  _clone_into(a, sim)
  _clone_into(b, sim)
  # a small batch holding a slice of the same worlds, and a permuted full batch
  idx = torch.arange(100, 108, device="cuda")
  small = _fresh(model, 8)
  _clone_into(small, sim, index=idx)
  perm = torch.randperm(NWORLD, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
  p = _fresh(model, NWORLD)
  _clone_into(p, sim, index=perm)
  for s in (a, b, small, p):
    for _ in range(4):
      s.step()
    s.forward()
  torch.cuda.synchronize()
  for f in OUT:
    A = getattr(a.data, f)
    assert torch.equal(A, getattr(b.data, f)), f"run-to-run: {f}"
    assert torch.equal(A[idx], getattr(small.data, f)), f"batch size: {f}"
    assert torch.equal(A[perm], getattr(p.data, f)), f"world order: {f}"
  assert torch.isfinite(a.data.qpos).all()


@pytest.mark.parametrize("name", ["g1_velocity_flat", "go1_velocity_flat", "g1_tracking_flat"])
def test_equations_of_motion_and_kkt(name):
  """M qacc = qfrc_smooth + J^T f;  f = -D min(0, J qacc - aref) >= 0  on every world."""
  import torch

  sim, _, model = _rollout_state(name, steps=8)
  sim.forward()
  torch.cuda.synchronize()
  d = sim.data
  nv, njm = model.nv, sim.njmax
  M = d.qM.double()
  qacc, qs, fc = d.qacc.double(), d.qfrc_smooth.double(), d.qfrc_constraint.double()
  J = d.efc_J.view(NWORLD, njm, nv).double()
  f = d.efc_force.double()
  rows = torch.arange(njm, device="cuda")[None, :] < d.nefc.view(-1, 1)
  f = torch.where(rows, f, torch.zeros_like(f))
  J = torch.where(rows[:, :, None], J, torch.zeros_like(J))
  # (1) J^T f is what the solver reports as the constraint force
  jtf = torch.einsum("wrv,wr->wv", J, f)
  scale = fc.abs().amax(dim=1, keepdim=True).clamp_min(1.0)
  assert float(((jtf - fc).abs() / scale).max()) < 1e-4
  # (2) equations of motion hold at the solver's iterate to its tolerance: the residual is the
  # Newton gradient, which the solver drives below tolerance / scale except where the
  # 10-iteration cap binds -> require it small relative to the forces on 99% of the worlds
  res = torch.einsum("wij,wj->wi", M, qacc) - qs - fc
  rel = res.abs().amax(dim=1) / (qs.abs().amax(dim=1) + fc.abs().amax(dim=1)).clamp_min(1.0)
  assert float(torch.quantile(rel, 0.99)) < 1e-3
  assert float(rel.max()) < 5e-2
  # (3) unilateral: forces are non-negative, and positive only on rows with J qacc - aref < 0
  assert float(f.min()) >= 0.0
  jar = torch.einsum("wrv,wv->wr", J, qacc) - torch.where(rows, d.efc_aref.double(), torch.zeros_like(f))
  Dm = torch.where(rows, d.efc_D.double(), torch.zeros_like(f))
  f_expect = -Dm * jar.clamp_max(0.0)
  fs = f.amax(dim=1, keepdim=True).clamp_min(1.0)
  assert float(((f - f_expect).abs() / fs).max()) < 2e-3
  # (4) bookkeeping
  assert int(d.nefc.max()) <= njm and int(d.ncon.max()) <= sim.nconmax
  assert int(d.solver_niter.max()) <= model.opt.iterations
  assert torch.isfinite(d.qacc).all()
  sd = d.sensordata
  assert float(sd.min()) >= 0.0 and torch.equal(sd, sd.round())  # contact counts


def test_free_flight_momentum():
  """No ground contact: internal (actuator / limit / damping / self-contact) forces cannot change the linear momentum;
  per step it changes by exactly m*g*h (the implicit damping term has no entry on the base)."""
  import torch

  from mjlab_amd import robots

  # rigid-ish flight: random joint pose held by the position actuators, random base twist.
  # (With fast joint motion the semi-implicit update itself changes momentum by
  # O(h^2 * dA/dt * qacc); that is the integrator's property, not what is tested here.)
  model = robots.load_model("g1_velocity_flat")
  n = 512
  s = _fresh(model, n, graph=False)
  g = torch.Generator(device="cuda").manual_seed(5)
  q = torch.tensor(model.key_qpos[0], dtype=torch.float32, device="cuda").repeat(n, 1)
  q[:, 2] += 3.0
  q[:, 7:] += 0.2 * torch.randn((n, model.nq - 7), device="cuda", generator=g)
  quat = torch.randn((n, 4), device="cuda", generator=g)
  q[:, 3:7] = quat / quat.norm(dim=1, keepdim=True)
  s.data.qpos[:] = q
  s.data.qvel[:] = 0
  s.data.qvel[:, :6] = torch.randn((n, 6), device="cuda", generator=g)
  s.data.ctrl[:] = q[:, 7:]
  root = int(np.nonzero(model.body_parentid == 0)[0][-1])  # last child of the world = robot root
  mass = torch.tensor(model.body_mass, dtype=torch.float64, device="cuda")
  robot = torch.zeros(model.nbody, dtype=torch.bool, device="cuda")
  robot[root : root + int(model.body_subtreenum[root])] = True

  def com_velocity():
    s.forward()
    d = s.data
    c = d.cvel.double()  # [angular, linear] at subtree_com[root]
    r = d.xipos.double() - d.subtree_com[:, root : root + 1].double()
    v = c[..., 3:] + torch.cross(c[..., :3], r, dim=-1)
    mm = (mass * robot)[None, :, None]
    return (mm * v).sum(dim=1) / mm.sum()

  vel = []
  for _ in range(5):
    vel.append(com_velocity())
    s.step()
  torch.cuda.synchronize()
  assert float(s.data.qpos[:, 2].min()) > 2.0  # nowhere near the ground (self-contacts are internal too)
  gh = torch.tensor(model.opt.gravity, dtype=torch.float64, device="cuda") * model.opt.timestep
  for k in range(4):
    err = (vel[k + 1] - vel[k] - gh).abs().amax(dim=1)
    # exact up to the integrator's own O(h^2 dA/dt qacc) term, which is only large in the few
    # worlds where the random pose starts deep inside a joint limit
    assert float(err.median()) < 5e-5 and float(torch.quantile(err, 0.9)) < 1e-3 and float(err.max()) < 0.2 * abs(float(gh[2])), k
  # instantaneous statement without integrator terms: at rest relative to the base (no joint or
  # angular velocity) the centre of mass accelerates with exactly g, whatever the internal forces
  s.data.qpos[:] = q
  s.data.qvel[:] = 0
  s.data.qvel[:, :3] = torch.randn((n, 3), device="cuda", generator=g)
  s.forward()
  d = s.data
  a_com = torch.einsum("wij,wj->wi", d.qM[:, :3, :].double(), d.qacc.double()) / float(mass[robot].sum())
  grav = torch.tensor(model.opt.gravity, dtype=torch.float64, device="cuda")
  assert float((a_com - grav).abs().max()) < 5e-4
  assert float(d.qfrc_constraint[:, :3].abs().max()) < 1e-3  # limits / self-contacts do not push the base


def test_diverged_world_is_isolated():
  import torch

  sim, _, model = _rollout_state("g1_velocity_flat", nworld=256, steps=4)
  a, b = _fresh(model, 256), _fresh(model, 256)
  _clone_into(a, sim)
  _clone_into(b, sim)
  b.data.qpos[17, 9] = float("nan")
  b.data.qvel[200, :] = float("inf")
  for s in (a, b):
    for _ in range(3):
      s.step()
  torch.cuda.synchronize()
  keep = torch.ones(256, dtype=torch.bool, device="cuda")
  keep[17] = keep[200] = False
  for f in ("qpos", "qvel", "qacc", "xpos"):
    assert torch.equal(getattr(a.data, f)[keep], getattr(b.data, f)[keep]), f


@pytest.mark.parametrize("scene", ["g1_velocity_rough", "go1_velocity_rough"])
def test_garbage_states_on_the_terrain_neither_fault_nor_leak(scene):
  """Worlds whose state is NaN / infinite / astronomically far away (what a diverged policy produces before the NaN guard
  or the termination test sees it) must not take the launch down -- the terrain grid walk turns positions into cell
  indices -- nor touch their neighbours: through the stage kernels and through the whole-control-step kernel."""
  import torch

  from mjlab_amd import robots
  from mjlab_amd.rollout import VELOCITY_TASK_EVENTS, PhysicsRollout, g1_action_scale, go1_action_scale
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model(scene)
  robot = "go1" if scene.startswith("go1") else "g1"
  scale = go1_action_scale(model) if robot == "go1" else g1_action_scale(model)
  n = 256
  bad = {3: float("nan"), 40: float("inf"), 41: -float("inf"), 77: 3.0e38, 130: -1.0e30, 200: 1.0e9}
  sims = []
  for poison in (False, True):
    sim = Simulation(n, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
    roll = PhysicsRollout(sim, action_scale=scale, seed=3, substeps_per_call=4, control_kernel=True, episode_length_s=1e6, **VELOCITY_TASK_EVENTS[robot])
    gen = torch.Generator(device="cuda").manual_seed(1)
    for k in range(6):
      if poison and k == 2:
        for w, v in bad.items():
          sim.data.qpos[w, :3] = v  # the floating base's position
        sim.data.qvel[9, :] = float("nan")
      act = torch.rand((n, model.nu), device="cuda", generator=gen) * 2 - 1
      roll.step(act)
      if poison and k == 3:  # and once through the per-stage kernels
        sim2 = Simulation(n, SimulationCfg(njmax=300, use_graph=False, fuse="stage"), model, "cuda:0")
        for f in ("qpos", "qvel", "ctrl"):
          getattr(sim2.data, f)[:] = getattr(sim.data, f)
        for w, v in bad.items():
          sim2.data.qpos[w, :3] = v
        sim2.step()
        sim2.forward()
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    sims.append(sim)
  keep = torch.ones(n, dtype=torch.bool, device="cuda")
  keep[list(bad) + [9]] = False
  for f in ("qpos", "qvel", "qacc", "xpos", "sensordata"):
    assert torch.equal(getattr(sims[0].data, f)[keep], getattr(sims[1].data, f)[keep]), f
  assert torch.isfinite(sims[1].data.qpos[keep]).all()


def test_forward_is_idempotent_and_stage_split_equals_step():
  import torch

  from mjlab_amd import native

  sim, _, model = _rollout_state("g1_velocity_flat", nworld=512, steps=4)
  a, b = _fresh(model, 512, graph=False), _fresh(model, 512, graph=False)
  _clone_into(a, sim)
  _clone_into(b, sim)
  q0, v0, t0 = a.data.qpos.clone(), a.data.qvel.clone(), a.data.time.clone()
  a.forward()
  x1 = {f: getattr(a.data, f).clone() for f in OUT}
  a.forward()
  for f in OUT:
    if f in ("qacc", "qfrc_constraint"):
      # the second forward starts the Newton solve from the first one's result (forward refreshes
      # qacc_warmstart, as mj_fwdConstraint does), so it stops at a marginally different iterate
      ref = x1[f].abs().amax().clamp_min(1.0)
      assert float((x1[f] - getattr(a.data, f)).abs().max() / ref) < 1e-3, f
    else:
      assert torch.equal(x1[f], getattr(a.data, f)), f
  assert torch.equal(q0, a.data.qpos) and torch.equal(v0, a.data.qvel) and torch.equal(t0, a.data.time)
  # forward stages one by one, then integrate only  ==  one step (from the same state:
  # forward() also refreshes qacc_warmstart, like mj_fwdConstraint does)
  _clone_into(a, sim)
  for bits in (native.STAGE_POSITION, native.STAGE_COLLISION, native.STAGE_VELOCITY, native.STAGE_CONSTRAINT,
               native.STAGE_SOLVE, native.STAGE_INTEGRATE):
    a.forward_stages(bits)
  b.step()
  torch.cuda.synchronize()
  for f in ("qpos", "qvel", "qacc", "time", "qacc_warmstart"):
    assert torch.equal(getattr(a.data, f), getattr(b.data, f)), f
  assert float(b.data.time[0]) == pytest.approx(float(t0[0]) + model.opt.timestep, rel=1e-6)


def test_many_contact_state_matches_oracle():
  """Robot lying on the ground: self-collisions + many ground contacts (rows near the
  capacity the tasks configure)."""
  import torch

  from make_golden import models

  from mjlab_amd.sim import Simulation, SimulationCfg
  from oracle.oracle import OracleSim

  model = models()["g1_velocity_flat"]
  nworld = 6
  rng = np.random.default_rng(21)
  qpos = np.tile(model.key_qpos[0], (nworld, 1))
  qpos[:, 2] = 0.12 + rng.uniform(0, 0.03, nworld)  # pelvis just above the floor
  ang = rng.uniform(0, 2 * np.pi, nworld)
  qpos[:, 3], qpos[:, 4], qpos[:, 5], qpos[:, 6] = np.cos(np.pi / 4), np.sin(np.pi / 4) * np.cos(ang), np.sin(np.pi / 4) * np.sin(ang), 0
  qpos[:, 7:] += rng.normal(0, 0.3, (nworld, model.nq - 7))
  qvel = rng.normal(0, 0.2, (nworld, model.nv))
  ctrl = qpos[:, 7:]
  sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
  ora = OracleSim(model, nworld, njmax=300, precision="f64")
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    getattr(ora, f)[:] = v
  sim.forward()
  ora.forward()
  torch.cuda.synchronize()
  assert np.array_equal(sim.data.ncon.cpu().numpy().ravel(), ora.ncon.ravel())
  assert np.array_equal(sim.data.nefc.cpu().numpy().ravel(), ora.nefc.ravel())
  assert ora.ncon.max() >= 12  # the state really is contact rich
  nv = model.nv
  for w in range(nworld):
    n = int(ora.nefc[w, 0])
    Jg = sim.data.efc_J.cpu().numpy()[w].reshape(-1, nv)[:n]
    Jo = ora.efc_J[w].reshape(-1, nv)[:n]
    assert np.abs(Jg - Jo).max() / max(1e-6, np.abs(Jo).max()) < 1e-5
    for f in ("efc_D", "efc_aref"):
      g, o = getattr(sim.data, f).cpu().numpy()[w, :n], getattr(ora, f)[w, :n]
      assert np.abs(g - o).max() / max(1e-6, np.abs(o).max()) < 2e-4, f
  assert np.array_equal(sim.data.sensordata.cpu().numpy(), ora.sensordata.astype(np.float32))


def test_masked_forward_touches_only_marked_worlds():
  import torch

  sim, _, model = _rollout_state("g1_velocity_flat", nworld=512, steps=4)
  a, b = _fresh(model, 512, graph=False), _fresh(model, 512, graph=False)
  _clone_into(a, sim)
  _clone_into(b, sim)
  a.forward()
  b.forward()
  # move every world, then recompute only the marked ones on `b`, all of them on `a`
  for s in (a, b):
    s.data.qpos[:, 7:] += 0.05
    s.data.qpos[:, 2] += 0.01
  before = {f: getattr(b.data, f).clone() for f in OUT}
  mask = torch.zeros(512, dtype=torch.bool, device="cuda")
  mask[::3] = True
  a.forward()
  b.forward(mask)
  torch.cuda.synchronize()
  for f in OUT:
    if f in ("qpos", "qvel"):
      continue
    assert torch.equal(getattr(a.data, f)[mask], getattr(b.data, f)[mask]), f"marked: {f}"
    assert torch.equal(before[f][~mask], getattr(b.data, f)[~mask]), f"unmarked: {f}"


def test_whole_step_graph_equals_eager_rollout():
  """One hipGraph replay per control step (action -> 4 substeps -> termination -> masked reset ->
  forward) gives bitwise the same physics as the eager sequence of the same calls."""
  import torch

  from mjlab_amd import robots
  from mjlab_amd.rollout import PhysicsRollout, g1_action_scale
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model("g1_velocity_flat")
  rolls = []
  for graph in (False, True):
    sim = Simulation(256, SimulationCfg(njmax=300), model, "cuda:0")
    # no termination inside the window, so the reset sampler's random numbers are never used
    roll = PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=7, episode_length_s=1e6, min_height=-10.0)
    if graph:
      state = {f: getattr(sim.data, f).clone() for f in ("qpos", "qvel", "ctrl", "qacc_warmstart")}
      elen = roll.episode_length.clone()
      roll.capture_graph()  # the capture warm-up advances the state once: restore it
      for f, v in state.items():
        getattr(sim.data, f)[:] = v
      roll.episode_length.copy_(elen)
      sim.forward()
    rolls.append(roll)
  gen = torch.Generator(device="cuda").manual_seed(123)
  for _ in range(6):
    act = torch.rand((256, model.nu), device="cuda", generator=gen) * 2 - 1
    r0 = rolls[0].step(act)
    r1 = rolls[1].step(act)
    assert not bool(r0.any()) and not bool(r1.any())
  torch.cuda.synchronize()
  assert rolls[1]._graph is not None
  for f in ("qpos", "qvel", "qacc", "xpos", "sensordata"):
    assert torch.equal(getattr(rolls[0].sim.data, f), getattr(rolls[1].sim.data, f)), f


def test_wave_priorities_do_not_change_results():
  """`data.sched_thr` (score quantiles written by `update_priority_thresholds`) only ranks the waves of a SIMD against each
  other (s_setprio): a rollout with the refresh, one without it and one with absurd thresholds are bitwise the same."""
  import torch

  from mjlab_amd import robots
  from mjlab_amd.rollout import VELOCITY_TASK_EVENTS, PhysicsRollout, go1_action_scale
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model("go1_velocity_flat")
  rolls = []
  for mode in ("refresh", "static", "absurd"):
    sim = Simulation(1024, SimulationCfg(njmax=100), model, "cuda:0")
    sim.priority_refresh = mode == "refresh"
    if mode == "absurd":
      sim.data.sched_thr.view(-1)[:3] = torch.tensor([1, 2, 3], dtype=torch.int32, device="cuda")  # every world in the top class
    rolls.append(PhysicsRollout(sim, action_scale=go1_action_scale(model), seed=5, substeps_per_call=4, control_kernel=True, **VELOCITY_TASK_EVENTS["go1"]))
  gen = torch.Generator(device="cuda").manual_seed(9)
  for _ in range(40):
    act = torch.rand((1024, model.nu), device="cuda", generator=gen) * 2 - 1
    for r in rolls:
      r.step(act)
  torch.cuda.synchronize()
  thr = rolls[0].sim.data.sched_thr.view(-1)[:3].tolist()
  assert 0 < thr[0] <= thr[1] <= thr[2]  # refreshed at control steps 16 and 32 from the batch's own scores
  assert rolls[1].sim.data.sched_thr.view(-1)[:3].tolist() == [0, 0, 0]
  for f in ("qpos", "qvel", "qacc", "efc_force", "sensordata", "solver_niter"):
    for r in rolls[1:]:
      assert torch.equal(getattr(rolls[0].sim.data, f), getattr(r.sim.data, f)), f


def test_fused_masked_reset_equals_torch_chain():
  """mjlab_masked_reset vs the same termination + reset logic as torch ops (the reference's
  style): identical reset decisions and untouched worlds; reset worlds carry a valid sample of
  the reset distribution (velocity_env_cfg.py:136-144)."""
  import torch

  from mjlab_amd import robots
  from mjlab_amd.rollout import PhysicsRollout, g1_action_scale
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model("g1_velocity_flat")
  rolls = []
  for fused in (False, True):
    sim = Simulation(1024, SimulationCfg(njmax=300), model, "cuda:0")
    rolls.append(PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=11, fused_reset=fused, episode_length_s=0.6))
  a, b = rolls
  b.sim.data.qpos[:] = a.sim.data.qpos
  b.episode_length.copy_(a.episode_length)
  gen = torch.Generator(device="cuda").manual_seed(5)
  key = torch.tensor(model.key_qpos[0], dtype=torch.float32, device="cuda")
  total = 0
  for k in range(12):
    act = torch.rand((1024, model.nu), device="cuda", generator=gen) * 2 - 1
    if k == 5:  # a diverged world and a fallen one
      for r in (a, b):
        r.sim.data.qpos[3, 10] = float("nan")
        r.sim.data.qpos[9, 2] = 0.1
    ra, rb = a.step(act).bool(), b.step(act).bool()  # fused reset: int32 mask (non-zero = reset); torch path: bool
    assert torch.equal(ra, rb), k
    keep = ~ra
    for f in ("qpos", "qvel", "qacc_warmstart"):
      assert torch.equal(getattr(a.sim.data, f)[keep], getattr(b.sim.data, f)[keep]), (k, f)
    assert torch.equal(a.episode_length, b.episode_length)
    if bool(rb.any()):
      q = b.sim.data.qpos[rb]
      assert torch.equal(q[:, 7:], key[7:].expand_as(q[:, 7:])) and torch.equal(q[:, 2], key[2].expand_as(q[:, 2]))
      assert float((q[:, :2] - key[:2]).abs().max()) <= 0.5
      assert float(q[:, 4:6].abs().max()) == 0.0 and float((q[:, 3:7].norm(dim=1) - 1).abs().max()) < 1e-6
      assert float(b.sim.data.qvel[rb].abs().max()) == 0.0
      # the reset worlds are brought back in sync so that the two rollouts stay comparable
      a.sim.data.qpos[ra] = b.sim.data.qpos[rb]
      a.sim.forward()
      b.sim.forward()
    total += int(rb.sum())
  assert total >= 2 + 1024 // 30  # the two forced resets plus time-outs of the 0.6 s episodes


@pytest.mark.parametrize("name,stand_steps,min_z", [("g1_velocity_flat", 160, 0.55), ("go1_velocity_flat", 2000, 0.2)])
def test_long_horizon_stability(name, stand_steps, min_z):
  """Physical plausibility over a long horizon, 4096 worlds from slightly different poses with the
  position actuators commanded to the keyframe pose.  The quadruped is statically stable and must
  settle and stay up for the whole 10 s; the humanoid has no balance controller, so it must stay
  up for the first 0.8 s and then -- wherever it ends -- come to rest on the ground without
  anything diverging (2000 physics steps)."""
  import torch

  from mjlab_amd import robots
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model(name)
  sim = Simulation(NWORLD, SimulationCfg(njmax=300), model, "cuda:0")
  g = torch.Generator(device="cuda").manual_seed(1)
  key = torch.tensor(model.key_qpos[0], dtype=torch.float32, device="cuda")
  q = key.repeat(NWORLD, 1)
  q[:, 7:] += 0.03 * torch.randn((NWORLD, model.nq - 7), device="cuda", generator=g)
  q[:, :2] += torch.rand((NWORLD, 2), device="cuda", generator=g) - 0.5
  sim.data.qpos[:] = q
  sim.data.ctrl[:] = key[7:]
  root = int(np.nonzero(model.body_parentid == 0)[0][-1])
  z = {}
  for k in range(1, 2001):
    sim.step()
    if k in (stand_steps, 1500, 2000):
      z[k] = sim.data.qpos[:, 2].clone()
      if k == stand_steps:
        up = sim.data.xmat[:, root, 2, 2].clone()
  torch.cuda.synchronize()
  assert torch.isfinite(sim.data.qpos).all() and torch.isfinite(sim.data.qvel).all()
  assert float(z[stand_steps].min()) > min_z, float(z[stand_steps].min())
  assert float(up.median()) > 0.97 and float(up.min()) > 0.8  # torso upright while standing
  assert float(z[2000].min()) > 0.03  # nothing sinks through the floor
  assert float((z[2000] - z[1500]).abs().median()) < 5e-3  # at rest at the end (standing or lying)
  assert float(sim.data.qvel.abs().median()) < 0.05
  assert float(sim.data.time[0]) == pytest.approx(2000 * model.opt.timestep, rel=1e-4)


def test_forward_folded_into_next_step_is_bit_exact():
  """step() right after forward() skips the position / collision / constraint-build stages of every
  world whose qpos and qvel are bit-identical to what forward() saw (include/mjlab_amd.h);
  results equal a full recomputation bit for bit; a changed state or a touched per-world model
  field makes the world recompute."""
  import torch

  from mjlab_amd.sim import Simulation, SimulationCfg

  src, _, model = _rollout_state("g1_velocity_flat", nworld=512, steps=6)
  out = {}
  for fold in (True, False):
    s = Simulation(512, SimulationCfg(njmax=300, use_graph=fold, fold_forward=fold), model, "cuda:0")
    s.expand_model_fields(["geom_friction"])
    s.create_graph()
    _clone_into(s, src)
    s.forward()
    assert int(s.data.fold_valid.sum()) == 512
    s.data.ctrl[:] += 0.02          # a new action, like the env writes before stepping
    s.data.qvel[:100, 7] += 0.01    # worlds 0..99: the state itself changed after forward()
    s.step()
    s.step()
    torch.cuda.synchronize()
    assert int(s.data.fold_valid.sum()) == 0
    out[fold] = {f: getattr(s.data, f).clone() for f in OUT + ("efc_J", "efc_aref", "nefc", "ncon")}
    if fold:
      _clone_into(s, src)
      s.forward()
      s.step()
      torch.cuda.synchronize()
      reuse = s.data.fold_reuse.clone()
      assert int(reuse.sum()) == 512  # nothing but ctrl could have changed: all worlds reuse
      _clone_into(s, src)
      s.forward()
      s.data.qvel[:100, 7] += 0.01
      s.step()
      torch.cuda.synchronize()
      assert int(s.data.fold_reuse[:100].sum()) == 0 and int(s.data.fold_reuse[100:].sum()) == 412
      _clone_into(s, src)
      s.forward()
      s.model.geom_friction[100:150, :, 0] *= 1.0  # touching an expanded field is enough
      assert int(s.data.fold_valid.sum()) == 0
      s.step()
      torch.cuda.synchronize()
      assert int(s.data.fold_reuse.sum()) == 0
  for f in out[True]:
    assert torch.equal(out[True][f], out[False][f]), f
  # the switch is per Simulation (mjlab_option_t.flags), not process wide: two live sims do not interfere
  a = Simulation(64, SimulationCfg(njmax=300, fold_forward=True), model, "cuda:0")
  b = Simulation(64, SimulationCfg(njmax=300, fold_forward=False), model, "cuda:0")
  for s in (a, b):
    s.forward()
    s.step()
  torch.cuda.synchronize()
  assert int(a.data.fold_reuse.sum()) == 64 and int(b.data.fold_reuse.sum()) == 0
  # a one-world Simulation hands out writable model views without any expansion: touching one invalidates
  c = Simulation(1, SimulationCfg(njmax=300), model, "cuda:0")
  c.forward()
  assert int(c.data.fold_valid.sum()) == 1
  c.model.geom_friction[0, :, 0] *= 0.5
  assert int(c.data.fold_valid.sum()) == 0


@pytest.mark.parametrize("scene", ["g1_velocity_flat", "go1_velocity_rough"])
def test_fused_launch_structures_are_bit_identical(scene):
  """SimulationCfg.fuse: the four pre-solve stages in one kernel ("presolve") or a whole substep in one
  kernel ("step") run the same stage bodies as the one-kernel-per-stage pipeline: every output must be
  bit-identical over a rollout with resets, forward() folds and masked forwards."""
  import torch

  from mjlab_amd import robots
  from mjlab_amd.rollout import PhysicsRollout
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model(scene)
  fields = OUT + ("efc_J", "efc_aref", "efc_D", "nefc", "ncon", "xpos", "cvel", "qM", "sensordata", "qfrc_smooth", "contact_pos")
  out = {}
  # (launch structure, substeps per step() call; 0 = the whole control step as one mjlab_control_step launch)
  variants = (("stage", 1), ("presolve", 1), ("step", 1), ("stage", 4), ("step", 4), ("step", 0))
  robot = "g1" if scene.startswith("g1") else "go1"
  from mjlab_amd.rollout import VELOCITY_TASK_EVENTS

  for fuse, nsub in variants:
    s = Simulation(256, SimulationCfg(njmax=300, fuse=fuse), model, "cuda:0")
    roll = PhysicsRollout(s, action_scale=0.25, seed=9, min_height=0.3 if scene.startswith("g1") else 0.15, substeps_per_call=max(nsub, 1),
                          control_kernel=nsub == 0, **VELOCITY_TASK_EVENTS[robot])
    for k in range(12):
      roll.step(roll.random_action())
    s.forward(torch.arange(256, device="cuda") % 3 == 0)  # masked forward through the same launch structure
    s.step()
    torch.cuda.synchronize()
    out[(fuse, nsub)] = {f: getattr(s.data, f).clone() for f in fields}
  for v in variants[1:]:
    for f in fields:
      assert torch.equal(out[variants[0]][f], out[v][f]), (v, f)


def test_tracking_task_resets_in_the_control_kernel_equal_the_torch_chain():
  """BASELINE config 4 under its own events (mjlab_amd.rollout.TRACKING_TASK_EVENTS): a world that resets is put on a random
  phase of the motion with the cfg's pose / velocity / joint noise, the way MotionCommand._resample_command writes it
  (reference tasks/tracking/mdp/commands.py:299-363), and the anchor terminations (terminations.py:27-53) run against the
  motion frame of the world's phase.  The one-launch control kernel (mjlab_motion_reset_t) against the chain of separate calls
  + torch ops on the same uniforms: identical reset decisions and phases, untouched worlds bit-identical, reset states equal to
  rounding (sincos / fused multiply-adds differ in the last bit between the two), and what the cfg describes."""
  import torch

  from mjlab_amd import robots
  from mjlab_amd.rollout import TRACKING_TASK_EVENTS, PhysicsRollout, g1_action_scale, synthetic_motion
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = robots.load_model("g1_tracking_flat")
  ev = dict(TRACKING_TASK_EVENTS["g1"])
  motion = synthetic_motion(model)
  rolls = []
  for ck in (False, True):
    sim = Simulation(512, SimulationCfg(njmax=250, fuse="step"), model, "cuda:0")
    rolls.append(PhysicsRollout(sim, action_scale=g1_action_scale(model), seed=21, min_height=-1.0e9, fused_reset=ck, control_kernel=ck,
                                substeps_per_call=4 if ck else 1, motion=motion, **ev))
  a, b = rolls
  assert torch.equal(a.sim.data.qpos, b.sim.data.qpos) and torch.equal(a.motion["time_steps"], b.motion["time_steps"])
  q0 = a.sim.data.qpos.clone()
  # the initial states ARE reset states: pelvis within the pose noise of the motion's root, joints within the noise of a frame
  assert float((q0[:, 2] - 0.76).abs().max()) <= 0.0101 and float(q0[:, :2].abs().max()) <= 0.0501
  tab = torch.as_tensor(motion["joint_pos"], device="cuda")
  assert float((q0[:, 7:] - tab[a.motion["time_steps"].long()]).abs().max()) <= 0.1001
  assert float(a.sim.data.qvel[:, :3].abs().max()) > 0.1  # the velocity noise of the reset reached qvel
  gen = torch.Generator(device="cuda").manual_seed(5)
  total = 0
  for k in range(60):
    act = torch.rand((512, model.nu), device="cuda", generator=gen) * 2 - 1
    ra, rb = a.step(act).bool(), b.step(act).bool()
    assert torch.equal(ra, rb), k
    assert torch.equal(a.motion["time_steps"], b.motion["time_steps"]) and torch.equal(a.episode_length, b.episode_length)
    keep = ~ra
    for f in ("qpos", "qvel", "qacc_warmstart"):
      assert torch.equal(getattr(a.sim.data, f)[keep], getattr(b.sim.data, f)[keep]), (k, f)
      if bool(ra.any()) and f != "qacc_warmstart":  # (the warm start of a reset world is the forward()'s qacc of a state that differs in the last bit)
        assert float((getattr(a.sim.data, f)[ra] - getattr(b.sim.data, f)[ra]).abs().max()) < 3e-6, (k, f)
    if bool(ra.any()):
      q = b.sim.data.qpos[rb]
      assert float((q[:, 2] - 0.76).abs().max()) <= 0.0101 and float((q[:, 3:7].norm(dim=1) - 1).abs().max()) < 1e-6
      assert float((q[:, 7:] - tab[b.motion["time_steps"].long()[rb]]).abs().max()) <= 0.1001
      # the reset worlds are brought back in sync so that the two rollouts stay comparable
      for f in ("qpos", "qvel", "qacc_warmstart"):  # (the warm start too: the forward() below starts its Newton iteration from it)
        getattr(a.sim.data, f)[ra] = getattr(b.sim.data, f)[rb]
      a.sim.forward()
      b.sim.forward()
    total += int(ra.sum())
  assert total > 50, total  # random actions throw the robots off the motion: anchor terminations fire
  assert int(a.motion["time_steps"].max()) < motion["joint_pos"].shape[0]


def test_solver_optimality_conditions_at_full_size():
  """Size-independent properties of the constraint solve at BASELINE's full size (4096 G1 worlds on rollout
  states, no oracle involved): the published constraint force is J^T efc_force; efc_force is the penalty law of
  the published row residuals (f = -D min(0, J qacc - aref), never negative); and the Newton iteration stopped
  at a stationary point of MuJoCo's convex cost: the gradient M qacc - qfrc_smooth - J^T f, scaled as the
  solver scales it (1 / (meaninertia nv)), is at rounding-noise level in all but the few worlds that ran into
  the 10-iteration cap."""
  import torch

  src, _, model = _rollout_state("g1_velocity_flat", nworld=4096, steps=12)
  sim = src
  sim.forward()
  torch.cuda.synchronize()
  d = sim.data
  n, nv, njmax = 4096, model.nv, sim.njmax
  J = d.efc_J.view(n, njmax, nv)
  rows = torch.arange(njmax, device="cuda")[None, :] < d.nefc.view(n, 1)
  f = torch.where(rows, d.efc_force, torch.zeros_like(d.efc_force))
  assert float(f.min()) >= 0.0
  jtf = torch.einsum("wrn,wr->wn", torch.where(rows[..., None], J, torch.zeros_like(J)), f)
  scale = d.qfrc_constraint.abs().amax(dim=1).clamp_min(1e-3)
  assert float(((jtf - d.qfrc_constraint).abs().amax(dim=1) / scale).max()) < 2e-5
  # penalty law of the rows, from the published J, qacc, aref, D
  jar = torch.einsum("wrn,wn->wr", J, d.qacc) - d.efc_aref
  law = torch.where(rows & (jar < 0), -d.efc_D * jar, torch.zeros_like(jar))
  fscale = f.amax(dim=1).clamp_min(1.0)
  assert float(((law - f).abs().amax(dim=1) / fscale).quantile(0.99)) < 2e-3  # fp32 J qacc - aref: a difference of large terms
  # stationarity
  grad = torch.einsum("wij,wj->wi", d.qM, d.qacc) - d.qfrc_smooth - d.qfrc_constraint
  g = grad.norm(dim=1) / (float(model.meaninertia) * nv)
  capped = d.solver_niter.view(-1) >= model.opt.iterations
  assert float(capped.float().mean()) < 0.05
  assert float(g[~capped].quantile(0.99)) < 1e-4 and float(g[~capped].median()) < 1e-5, (float(g.median()), float(g.quantile(0.99)))
  # worlds without constraints: qacc is the unconstrained acceleration
  free = d.nefc.view(-1) == 0
  if bool(free.any()):
    assert torch.equal(d.qacc[free], d.qacc_smooth[free])
