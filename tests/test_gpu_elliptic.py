"""Elliptic friction cones on the device (csrc/stage_cone.h + the ELL instantiation of the constraint stage):
MujocoCfg(cone="elliptic") (reference sim/sim.py:49,52) as stage kernels and their fused variants, against the CPU restatement (oracle/mjoracle.c), which
tests/test_oracle_elliptic.py holds to Coulomb's law, the optimality conditions and cone membership.  The cone model is restated from
MuJoCo's documentation and UNPINNED on both sides; Newton and (round 6) CG -- the dual solver keeps the pyramid."""

import copy
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

from make_golden import golden_inputs, models  # noqa: E402

from oracle.oracle import OracleSim  # noqa: E402

pytestmark = pytest.mark.gpu


def _np(t):
  import torch

  torch.cuda.synchronize()
  return t.cpu().numpy()


def _per_world(a, b):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  return np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-6)


@pytest.mark.parametrize("lsp", [False, True], ids=["exact_ls", "grid_ls"])
@pytest.mark.parametrize("name", ["mixed", "go1_velocity_flat", "g1_velocity_flat", "g1_velocity_rough"])
def test_elliptic_forward_and_rollout_track_the_restatement(name, lsp):
  import torch

  from mjlab_amd import _abi, mjcf
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = copy.deepcopy(models()[name])
  model.opt.cone = mjcf.CONE_ELLIPTIC
  model.opt.impratio = 2.0 if name == "mixed" else 1.0
  flags = 0
  if name == "go1_velocity_flat":  # friction-loss rows next to the cones
    model.dof_frictionloss = np.asarray(model.dof_frictionloss, dtype=np.float64).copy()
    model.dof_frictionloss[6:] = 0.2
    flags = _abi.OPT_FRICTIONLOSS
  nworld, nv = 16, model.nv
  # (the rough scene: robots on stair edges up to ~100 m from the origin -- fp32 contact geometry there is good to ~1e-5 of a Jacobian
  # entry and the solve inherits it, as in the pyramid's tests of that scene)
  rough = name.endswith("rough")
  qpos, qvel, ctrl = golden_inputs(model, nworld, 43)
  sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=False, ls_parallel=lsp, fuse="stage"), model, "cuda:0")
  ora = OracleSim(model, nworld, njmax=300, precision="f64", flags=flags, ls_parallel=lsp)
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    getattr(ora, f)[:] = v.astype(np.float32)
  sim.data.qacc_warmstart.zero_()  # (the constructor's forward() left its own solution there)
  sim.forward()
  ora.forward(nthread=8)
  # ---- the rows: three per condim-3 contact, friction rows without position, R_k from impratio
  assert np.array_equal(_np(sim.data.ncon).ravel(), ora.ncon.ravel())
  assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  nefc = ora.nefc.ravel()
  assert nefc.max() >= 12
  with_cones = 0
  tg, ig = _np(sim.data.efc_type), _np(sim.data.efc_id)
  for w in range(nworld):
    n = int(nefc[w])
    if n == 0:
      continue
    assert np.array_equal(tg[w, :n], ora.efc_type[w, :n]) and np.array_equal(ig[w, :n], ora.efc_id[w, :n])
    with_cones += (tg[w, :n] == 7).sum() >= 3
    Jg = _np(sim.data.efc_J)[w].reshape(-1, nv)[:n]
    assert np.abs(Jg - ora.efc_J[w].reshape(-1, nv)[:n]).max() < (1e-4 if rough else 2e-6) * max(1.0, np.abs(Jg).max())
    for f, tol in (("efc_D", 1e-3), ("efc_aref", 1e-3), ("efc_pos", 1e-4), ("efc_margin", 1e-6)):
      a, b = _np(getattr(sim.data, f))[w, :n], getattr(ora, f)[w, :n]
      assert np.abs(a - b).max() <= (30 if rough else 1) * tol * max(1e-6, np.abs(b).max()), (f, w)
  assert with_cones >= nworld // 2
  # ---- the solve
  qa, fo = _np(sim.data.qacc), _np(sim.data.efc_force)
  err = _per_world(qa, ora.qacc)
  fric = _np(sim.data.contact_friction).reshape(nworld, -1, 5)
  mu_scale = 1.0 / np.sqrt(model.opt.impratio)
  kkt, incone = [], 0
  for w in range(nworld):
    n = int(nefc[w])
    if n == 0:
      continue
    Jw = _np(sim.data.efc_J)[w].reshape(-1, nv)[:n].astype(np.float64)
    Mw = _np(sim.data.qM)[w].reshape(nv, nv).astype(np.float64)
    res = Mw @ qa[w] - _np(sim.data.qfrc_smooth)[w] - Jw.T @ fo[w, :n]
    kkt.append(np.abs(res).max() / max(1.0, np.abs(_np(sim.data.qfrc_smooth)[w]).max()))
    assert np.abs(_np(sim.data.qfrc_constraint)[w] - Jw.T @ fo[w, :n]).max() < 1e-4 * max(1.0, np.abs(fo[w, :n]).max())
    ell = np.flatnonzero(tg[w, :n] == 7)
    for r in ell[::3]:
      f0, f1, f2 = fo[w, r : r + 3]
      fr = fric[w, ig[w, r]]
      # inside the friction cone of the regularised problem: (f1 / friction1)^2 + (f2 / friction2)^2 <= (f0 / mu)^2 ... in force space
      # |f_t| <= friction * f_n with mu = friction / sqrt(impratio) scaling the normal axis of the dual cone
      assert f0 >= -1e-5
      assert np.hypot(f1 / fr[0], f2 / fr[1]) <= f0 * (1 + 1e-3) + 1e-4 * max(1.0, abs(f0)), (w, r, f0, f1, f2)
      incone += f0 > 1e-3
  assert incone >= nworld // 2  # (contacts that push)
  print(f"\n{name} lsp={lsp}: elliptic qacc device vs restatement median {np.median(err):.2e} max {err.max():.2e}; stationarity residual max {max(kkt):.2e}; "
        f"iterations device {_np(sim.data.solver_niter).mean():.1f} restatement {ora.solver_niter.mean():.1f}; mu scale {mu_scale:.3f}")
  # measured (profiles/r05_v20_elliptic.txt): exact search median <= 2.3e-6, max <= 2.1e-5; grid search median <= 2.7e-6, max <= 9.8e-5
  assert np.median(err) < 1e-5 and err.max() < (4e-4 if lsp else (1e-4 if rough else 8e-5)), err  # (rough, measured: 2.6e-6 / 2.6e-5)
  assert max(kkt) < (2e-2 if lsp else (1.5e-4 if rough else 5e-5))  # (the grid search stops where no candidate step improves: stationary to the grid only)
  # ---- a short rollout (stage launches: constraint-cone, solve-cone, integrate)
  for _ in range(10):
    sim.step()
  ora.step(10, nthread=8)
  assert np.isfinite(_np(sim.data.qpos)).all()
  perr = _per_world(_np(sim.data.qpos), ora.qpos)
  print(f"{name} lsp={lsp}: qpos after 10 steps median {np.median(perr):.2e} max {perr.max():.2e}")
  assert (np.median(perr) < 1.5e-5 and perr.max() < 8e-5) if lsp else (np.median(perr) < 1e-6 and perr.max() < 3e-6)  # measured: 3.5e-6 / 1.7e-5 | 1.6e-7 / 6.9e-7


def test_elliptic_slab_slides_by_coulombs_law_on_the_device():
  """The same physics check the restatement passes (tests/test_oracle_elliptic.py), on the device: a slab on a 38.7 degree incline
  (tan = 0.8, mu = 0.5) whose downhill direction is NOT a contact-frame axis accelerates at g (sin - mu cos) along the slope and
  not across it, with every live contact force on its cone."""
  import torch

  from mjlab_amd.sim import Simulation, SimulationCfg

  sys.path.insert(0, str(ROOT / "tests"))
  from test_oracle_elliptic import G, _box, _tilt

  mu, th, phi = 0.5, np.arctan(0.8), 1.1
  model = _box(mu, _tilt(th, phi))
  sim = Simulation(4, SimulationCfg(ls_parallel=False), model, "cuda:0")  # (use_graph default: the stage launches inside a captured step)
  for _ in range(100):
    sim.step()
  v0 = _np(sim.data.qvel)[:, :3].copy()
  ratios = []
  for _ in range(500):
    sim.step()
    n = int(_np(sim.data.nefc)[0].ravel()[0])
    if n:
      f = _np(sim.data.efc_force)[0, :n].reshape(-1, 3)
      live = f[:, 0] > 1e-6
      ratios.append(np.hypot(f[live, 1], f[live, 2]) / f[live, 0])
  a = (_np(sim.data.qvel)[:, :3] - v0) / (500 * model.opt.timestep)
  want = G * (np.sin(th) - mu * np.cos(th))
  along = a[:, 0] * np.cos(phi) + a[:, 1] * np.sin(phi)
  across = -a[:, 0] * np.sin(phi) + a[:, 1] * np.cos(phi)
  ratios = np.concatenate(ratios)
  print(f"\nslab: along / want {along / want}, across / want {across / want}, |f_t| / f_n in [{ratios.min():.4f}, {ratios.max():.4f}] over {ratios.size} contact-steps")
  assert ratios.size > 200 and np.allclose(ratios, mu, rtol=5e-3)
  assert np.allclose(along, want, rtol=0.06) and (np.abs(across) < 0.02 * want).all()


@pytest.mark.parametrize("njmax,lsp", [(300, False), (300, True), (1700, False)], ids=["rows_64_plus", "rows_64_plus_grid", "layout_without_M"])
def test_elliptic_many_rows_and_the_layout_without_M(njmax, lsp):
  """The paths the seeded states never reach: worlds with more than 64 rows (the constraint update then runs in 64-row trips with the
  tiles carried across them) and the layout without M in LDS (the BIG instantiation: here forced for every world by an njmax so large
  that no row fits next to M).  States: the 32 worlds with the most rows among the parity gate's rollout states (fallen, self-colliding
  robots; per-world foot friction), against the fp64 restatement."""
  import torch

  from mjlab_amd import mjcf, robots
  from mjlab_amd.sim import Simulation, SimulationCfg

  z = np.load(ROOT / "tests" / "golden" / "rollout_states_g1_velocity_flat.npz")
  model = copy.deepcopy(robots.load_model("g1_velocity_flat"))
  model.opt.cone = mjcf.CONE_ELLIPTIC
  probe = OracleSim(model, z["qpos"].shape[0], njmax=300, precision="f64")
  for f in ("qpos", "qvel", "ctrl"):
    getattr(probe, f)[:] = z[f]
  probe.forward(nthread=8)
  pick = np.argsort(-probe.nefc.ravel(), kind="stable")[:32]
  n = pick.size
  for fuse in ("stage", "step"):
    sim = Simulation(n, SimulationCfg(njmax=njmax, use_graph=False, ls_parallel=lsp, fuse=fuse), model, "cuda:0")
    ora = OracleSim(model, n, njmax=njmax, precision="f64", ls_parallel=lsp)
    sim.expand_model_fields(["geom_friction"])
    sim.model.geom_friction[:] = torch.from_numpy(z["dr_geom_friction"][pick].astype(np.float32)).cuda()
    ora.expand_model_field("geom_friction")[:] = z["dr_geom_friction"][pick]
    for f in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
      getattr(sim.data, f)[:] = torch.from_numpy(z[f][pick].astype(np.float32)).cuda()
      getattr(ora, f)[:] = z[f][pick].astype(np.float32)
    sim.forward()
    ora.forward(nthread=8)
    nefc = ora.nefc.ravel()
    assert np.array_equal(_np(sim.data.nefc).ravel(), nefc) and (nefc > 64).sum() >= 3 and nefc.max() >= 80
    if njmax > 300:  # no row fits next to M: every world runs the layout without it
      lds = sim.lds_bytes()["solve"]
      assert lds > 4 * 3 * njmax, lds
    err = _per_world(_np(sim.data.qacc), ora.qacc)
    big = nefc > 64
    print(f"\nelliptic, rollout states, njmax {njmax}, fuse {fuse}, lsp {lsp}: qacc median {np.median(err):.2e} max {err.max():.2e}; worlds with > 64 rows: max {err[big].max():.2e}; "
          f"iterations device {_np(sim.data.solver_niter).mean():.1f} restatement {ora.solver_niter.mean():.1f}")
    # (measured: profiles/r05_v33_elliptic_many_rows.txt)
    assert np.median(err) < 1e-5 and err.max() < (1.5e-4 if lsp else 6e-5), err  # measured 2.4e-6 / 1.4e-5 (grid 3.7e-6 / 2.7e-5)
    for _ in range(4):
      sim.step()
    ora.step(4, nthread=8)
    perr = _per_world(_np(sim.data.qpos), ora.qpos)
    assert np.isfinite(_np(sim.data.qpos)).all() and np.median(perr) < (5e-5 if lsp else 5e-6), perr


@pytest.mark.parametrize("lsp", [False, True], ids=["exact_ls", "grid_ls"])
def test_elliptic_on_the_parity_gates_rollout_states(lsp):
  """The distribution over all 256 rollout states of the pyramid's parity gate (G1 velocity-flat under its task events: standing,
  pushed, fallen and self-colliding robots, per-world foot friction, each with its own warm start), elliptic cones, device against the
  fp64 restatement: forward() and one step."""
  import torch

  from mjlab_amd import mjcf, robots
  from mjlab_amd.sim import Simulation, SimulationCfg

  z = np.load(ROOT / "tests" / "golden" / "rollout_states_g1_velocity_flat.npz")
  model = copy.deepcopy(robots.load_model("g1_velocity_flat"))
  model.opt.cone = mjcf.CONE_ELLIPTIC
  n = z["qpos"].shape[0]
  sim = Simulation(n, SimulationCfg(njmax=300, use_graph=False, ls_parallel=lsp), model, "cuda:0")
  ora = OracleSim(model, n, njmax=300, precision="f64", ls_parallel=lsp)
  sim.expand_model_fields(["geom_friction"])
  sim.model.geom_friction[:] = torch.from_numpy(z["dr_geom_friction"].astype(np.float32)).cuda()
  ora.expand_model_field("geom_friction")[:] = z["dr_geom_friction"]

  def load():
    for f in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
      getattr(sim.data, f)[:] = torch.from_numpy(z[f].astype(np.float32)).cuda()
      getattr(ora, f)[:] = z[f].astype(np.float32)

  load()
  sim.forward()
  ora.forward(nthread=8)
  assert np.array_equal(_np(sim.data.nefc).ravel(), ora.nefc.ravel())
  err = _per_world(_np(sim.data.qacc), ora.qacc)
  load()
  sim.step()
  ora.step(1, nthread=8)
  verr = _per_world(_np(sim.data.qvel), ora.qvel)
  perr = _per_world(_np(sim.data.qpos), ora.qpos)
  q = lambda e: (float(np.median(e)), float(np.percentile(e, 99)), float(e.max()))  # noqa: E731
  print(f"\nelliptic, 256 rollout states, lsp {lsp}: qacc median / p99 / max {q(err)[0]:.2e} / {q(err)[1]:.2e} / {q(err)[2]:.2e}; one step: qvel {q(verr)[0]:.2e} / {q(verr)[1]:.2e} / {q(verr)[2]:.2e}, "
        f"qpos {q(perr)[0]:.2e} / {q(perr)[1]:.2e} / {q(perr)[2]:.2e}; iterations device {_np(sim.data.solver_niter).mean():.2f} restatement {ora.solver_niter.mean():.2f}")
  # literals: measured x ~3 (profiles/r05_v35_elliptic_gate.txt)
  assert q(err)[0] < 8e-6 and q(err)[1] < (1.2e-4 if lsp else 4e-5) and q(err)[2] < (3e-4 if lsp else 5e-5), q(err)
  assert q(perr)[0] < 3e-7 and q(perr)[2] < (5e-5 if lsp else 1.2e-6), q(perr)


def test_elliptic_rows_that_do_not_fit_are_dropped_like_the_restatement():
  """njmax smaller than the rows the state wants: a cone's three rows fit together or not at all, both sides drop the same contacts and
  say so (data.overflow, MJLAB_OVF_NJMAX); and a state without contacts (nefc = limits only / 0) goes through the cone kernels too."""
  import torch

  from mjlab_amd import mjcf
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = copy.deepcopy(models()["g1_velocity_flat"])
  model.opt.cone = mjcf.CONE_ELLIPTIC
  n = 8
  qpos, qvel, ctrl = golden_inputs(model, n, 11)
  sim = Simulation(n, SimulationCfg(njmax=20, use_graph=False, ls_parallel=False), model, "cuda:0")
  ora = OracleSim(model, n, njmax=20, precision="f64")
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    getattr(ora, f)[:] = v.astype(np.float32)
  sim.forward()
  ora.forward()
  nefc = ora.nefc.ravel()
  assert np.array_equal(_np(sim.data.nefc).ravel(), nefc) and nefc.max() <= 20
  assert np.array_equal(_np(sim.data.overflow).ravel(), ora.overflow.ravel()) and (ora.overflow.ravel() & 2).any()
  tg = _np(sim.data.efc_type)
  for w in range(n):
    assert (tg[w, : nefc[w]] == 7).sum() % 3 == 0 and np.array_equal(tg[w, : nefc[w]], ora.efc_type[w, : nefc[w]])
  assert _per_world(_np(sim.data.qacc), ora.qacc).max() < 5e-5
  # in the air: no contact rows
  sim.data.qpos[:, 2] += 2.0
  ora.qpos[:, 2] += 2.0
  sim.step()
  ora.step()
  assert int(_np(sim.data.ncon).max()) == 0
  assert _per_world(_np(sim.data.qvel), ora.qvel).max() < 1e-5


def test_elliptic_launch_structures_are_bit_identical():
  """The cone variants of the fused kernels (kernels.h: k_substep_cone, k_control_step_cone) run the same stage bodies as the one-kernel-
  per-stage pipeline: every output bit-identical over a rollout with task events, resets, forward() folds and a masked forward -- the
  pyramid's test of the same name (tests/test_gpu_fullsize.py), with elliptic cones."""
  import torch

  from mjlab_amd import mjcf, robots
  from mjlab_amd.rollout import VELOCITY_TASK_EVENTS, PhysicsRollout
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = copy.deepcopy(robots.load_model("g1_velocity_flat"))
  model.opt.cone = mjcf.CONE_ELLIPTIC
  fields = ("qpos", "qvel", "qacc", "qacc_warmstart", "qfrc_constraint", "efc_force", "efc_J", "efc_aref", "efc_D", "efc_type", "nefc", "ncon", "xpos", "cvel", "qM",
            "sensordata", "qfrc_smooth", "contact_pos", "solver_niter")
  out = {}
  variants = (("stage", 1), ("step", 1), ("stage", 4), ("step", 4), ("step", 0))  # (launch structure, substeps per call; 0 = one mjlab_control_step launch)
  for fuse, nsub in variants:
    s = Simulation(256, SimulationCfg(njmax=300, fuse=fuse), model, "cuda:0")
    assert s.fuse == fuse
    roll = PhysicsRollout(s, action_scale=0.25, seed=9, min_height=0.3, substeps_per_call=max(nsub, 1), control_kernel=nsub == 0, **VELOCITY_TASK_EVENTS["g1"])
    resets = 0
    for k in range(12):
      resets += int(roll.step(roll.random_action()).sum())
    s.forward(torch.arange(256, device="cuda") % 3 == 0)
    s.step()
    torch.cuda.synchronize()
    out[(fuse, nsub)] = {f: getattr(s.data, f).clone() for f in fields}
    assert bool(torch.isfinite(s.data.qpos).all()) and int((s.data.efc_type == 7).sum()) >= 3 * 256 // 2
  for v in variants[1:]:
    for f in fields:
      assert torch.equal(out[variants[0]][f], out[v][f]), (v, f)


def test_elliptic_kernels_of_different_sizes_back_to_back():
  """Kernels of different padded sizes, launch structures and line searches one after the other in one process (a kernel must not
  depend on what the previous one left in registers, LDS or scratch: a build of the fused cone kernels at four waves per SIMD passed
  every test alone and faulted on the 32-dof instantiation right after the 36-dof one -- round 5.  Root cause, round 6: a spill store
  miscompiled to execute under EXEC == 0; the deterministic tests are tests/test_gpu_scratch.py -- scratch poisoned
  in front of every launch -- and tests/test_code_object.py -- the instruction pattern; this one keeps the original sequence)."""
  import gc

  import torch

  from mjlab_amd import mjcf
  from mjlab_amd.sim import Simulation, SimulationCfg

  base = models()
  order = ["g1_velocity_flat", "mixed", "go1_velocity_flat", "mixed", "g1_velocity_flat", "box", "mixed"]
  for rep, (fuse, lsp) in enumerate((("step", False), ("stage", False), ("step", True), ("stage", True))):
    for name in order:
      model = copy.deepcopy(base[name])
      model.opt.cone = mjcf.CONE_ELLIPTIC
      qpos, qvel, ctrl = golden_inputs(model, 8, 50 + rep)
      sim = Simulation(8, SimulationCfg(njmax=300, fuse=fuse, ls_parallel=lsp), model, "cuda:0")
      for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
        getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
      sim.forward()
      for _ in range(3):
        sim.step()
      sim.step(4)
      sim.forward()
      torch.cuda.synchronize()
      assert bool(torch.isfinite(sim.data.qpos).all()) and bool(torch.isfinite(sim.data.qacc).all()), (fuse, lsp, name)
      del sim
      gc.collect()


def test_elliptic_is_refused_where_it_is_not_carried():
  """The "presolve" launch structure and the other solvers carry the pyramid only: asked for elliptic cones they say so."""
  from mjlab_amd import _abi, mjcf
  from mjlab_amd.sim import Simulation, SimulationCfg, check_supported

  model = copy.deepcopy(models()["g1_velocity_flat"])
  model.opt.cone = mjcf.CONE_ELLIPTIC
  sim = Simulation(4, SimulationCfg(njmax=300, use_graph=False, fuse="presolve"), model, "cuda:0")
  assert sim.fuse == "stage"  # (the configuration's "presolve" gives way, as every fused structure does for the dual solver)
  sim._m.opt.flags |= _abi.OPT_FUSE_PRESOLVE
  with pytest.raises(RuntimeError, match="ELLIPTIC"):
    sim.forward()
  sim._m.opt.flags &= ~_abi.OPT_FUSE_PRESOLVE
  sim.forward()
  pgs = copy.deepcopy(model)
  pgs.opt.solver = mjcf.SOL_PGS
  with pytest.raises(NotImplementedError, match="elliptic"):
    check_supported(pgs)
  cg = copy.deepcopy(model)
  cg.opt.solver = mjcf.SOL_CG
  check_supported(cg)  # (round 6: CG carries the elliptic cone too -- test_elliptic_cg_*)



@pytest.mark.parametrize("fuse", ["stage", "step"])
@pytest.mark.parametrize("name,iterations", [("mixed", 200), ("go1_velocity_flat", 200), ("g1_velocity_flat", 10), ("g1_velocity_flat", 200)])
def test_elliptic_cg_tracks_the_restatement_and_converges_to_newton(name, iterations, fuse):
  """MujocoCfg(solver="cg", cone="elliptic") (reference sim/sim.py:49-56 accepts the pair): mj_solPrimal's Polak-Ribiere directions,
  preconditioned by M, over the cone solver's constraint update and line search.  Device vs the restatement's CG on the same states:
  at the iteration cap the same iterate as its fp32 build; with enough iterations the NEWTON solution of the elliptic problem (fp64
  restatement), stationary, forces inside their cones; both launch structures."""
  import torch

  from mjlab_amd import mjcf
  from mjlab_amd.sim import Simulation, SimulationCfg

  model = copy.deepcopy(models()[name])
  model.opt.cone = mjcf.CONE_ELLIPTIC
  model.opt.impratio = 2.0 if name == "mixed" else 1.0
  model.opt.solver = mjcf.SOL_CG
  model.opt.iterations = iterations
  nworld, nv = 16, model.nv
  qpos, qvel, ctrl = golden_inputs(model, nworld, 43)
  sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=False, ls_parallel=False, fuse=fuse), model, "cuda:0")
  assert sim.fuse == fuse
  newton = copy.deepcopy(model)
  newton.opt.solver, newton.opt.iterations = mjcf.SOL_NEWTON, 50
  oras = {"cg64": OracleSim(model, nworld, njmax=300, precision="f64", ls_parallel=False), "cg32": OracleSim(model, nworld, njmax=300, precision="f32", ls_parallel=False),
          "newton": OracleSim(newton, nworld, njmax=300, precision="f64", ls_parallel=False)}
  for f, v in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl)):
    getattr(sim.data, f)[:] = torch.from_numpy(v.astype(np.float32)).cuda()
    for o in oras.values():
      getattr(o, f)[:] = v.astype(np.float32)
  sim.data.qacc_warmstart.zero_()
  sim.forward()
  for o in oras.values():
    o.forward(nthread=8)
  nefc = oras["cg64"].nefc.ravel()
  assert np.array_equal(_np(sim.data.nefc).ravel(), nefc) and nefc.max() >= 12
  it_g = _np(sim.data.solver_niter).ravel()
  assert it_g.max() <= iterations and (it_g > 0).any()
  qa = _np(sim.data.qacc)
  busy = nefc > 0
  if iterations >= 50:
    e_cg, e_nt = _per_world(qa, oras["cg64"].qacc), _per_world(qa, oras["newton"].qacc)
    print(f"\n{name} {fuse}: elliptic CG device vs restatement CG median {np.median(e_cg):.2e} max {e_cg.max():.2e}; vs restatement NEWTON median {np.median(e_nt):.2e} max {e_nt.max():.2e}; "
          f"iterations device {it_g[busy].mean():.1f} restatement fp32 {oras['cg32'].solver_niter.ravel()[busy].mean():.1f} fp64 {oras['cg64'].solver_niter.ravel()[busy].mean():.1f}")
    # linear convergence + the fp32 noise floor of the termination test (the pyramid's CG test: 1e-2 in the same norm)
    # -- and the worst world no further from the fp64 iterate than twice what the restatement's OWN fp32 build is (mixed, impratio 2: 1.4e-2)
    e_ref = _per_world(oras["cg32"].qacc, oras["cg64"].qacc)
    bound = max(1e-2, 2.0 * float(e_ref.max()))
    print(f"{name} {fuse}: fp32 restatement vs fp64 restatement max {e_ref.max():.2e}")
    assert e_cg.max() < bound and e_nt.max() < bound and np.median(e_nt) < 2e-3
    assert (it_g[busy] < iterations).mean() > 0.7  # converged by the tolerance, not by the cap
    fo, tg, ig = _np(sim.data.efc_force), _np(sim.data.efc_type), _np(sim.data.efc_id)
    fric = _np(sim.data.contact_friction).reshape(nworld, -1, 5)
    for w in np.flatnonzero(busy):
      n = int(nefc[w])
      Jw = _np(sim.data.efc_J)[w].reshape(-1, nv)[:n].astype(np.float64)
      assert np.abs(_np(sim.data.qfrc_constraint)[w] - Jw.T @ fo[w, :n]).max() < 1e-4 * max(1.0, np.abs(fo[w, :n]).max())
      for r in np.flatnonzero(tg[w, :n] == 7)[::3]:
        f0, f1, f2 = fo[w, r : r + 3]
        fr = fric[w, ig[w, r]]
        assert f0 >= -1e-5 and np.hypot(f1 / fr[0], f2 / fr[1]) <= f0 * (1 + 1e-3) + 1e-4 * max(1.0, abs(f0)), (w, r, f0, f1, f2)
  else:
    e32 = _per_world(qa, oras["cg32"].qacc)
    print(f"\n{name} {fuse}: elliptic CG at the cap of {iterations}: device vs fp32 restatement median {np.median(e32):.2e} max {e32.max():.2e}; at the cap {float((it_g == iterations).mean()):.2f}")
    assert (oras["cg64"].solver_niter.ravel() == iterations).mean() > 0.3
    assert e32.max() < 5e-3  # (the pyramid's capped CG: 2e-3 in the relative norm; the same growth along the CG path)
    return
  for _ in range(5):
    sim.step()
  oras["cg64"].step(5, nthread=8)
  assert np.isfinite(_np(sim.data.qpos)).all()
  perr = _per_world(_np(sim.data.qpos), oras["cg64"].qpos)
  print(f"{name} {fuse}: qpos after 5 steps median {np.median(perr):.2e} max {perr.max():.2e}")
  assert perr.max() < 3e-3
