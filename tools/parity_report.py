"""GPU-vs-oracle parity over the state distribution of a rollout (GPU box; oracle on the host cores).

For each scene: N worlds are rolled out on the GPU with random actions (falls, self-collisions
and resets included), the resulting states (qpos, qvel, ctrl, qacc_warmstart) are handed to the
CPU oracle, both sides run forward() and one step(), and per-field relative errors
(max |gpu - oracle| / max |oracle| per world) are summarised.

  python tools/parity_report.py [N] [scene,scene,...] [control_steps] [f64|f32]

`scene_report()` is also what tests/test_gpu_parity_gate.py asserts on; the CLI output is
committed as profiles/<tag>/parity_report.txt.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots  # noqa: E402
from mjlab_amd.rollout import TRACKING_TASK_EVENTS, PhysicsRollout, g1_action_scale, go1_action_scale, synthetic_motion  # noqa: E402
from mjlab_amd.sim import Simulation, SimulationCfg  # noqa: E402
from oracle.oracle import OracleSim  # noqa: E402

KIN = ["xpos", "xquat", "xipos", "subtree_com", "geom_xpos", "site_xpos", "qM"]
VEL = ["cvel", "qfrc_bias", "actuator_force", "qfrc_smooth", "qacc_smooth"]
ROWS = ["efc_J", "efc_D", "efc_aref", "efc_pos"]
SOLVE = ["qacc", "qfrc_constraint"]
STEP = ["qpos", "qvel"]
ALL_SCENES = ["g1_velocity_flat", "g1_tracking_flat", "go1_velocity_flat", "g1_velocity_rough", "go1_velocity_rough"]


# Element-wise metric: |gpu - oracle| <= ATOL[field] + RTOL * |oracle| for EVERY element, so a
# small component next to a large one (a wrist dof's qacc beside the base's) cannot hide behind the array's largest value.
# RTOL is north_star's 1e-5; ATOL is 1e-5 x the magnitude below which an element of the field carries no information at
# fp32 (the scale of the terms it is a sum / difference of), in the field's own unit:
RTOL = 1e-5
ATOL = {
  "xpos": 1e-6, "xipos": 1e-6, "subtree_com": 1e-6, "geom_xpos": 1e-6, "site_xpos": 1e-6,  # metres; terms O(0.1 .. 1 m)
  "xquat": 1e-6, "qM": 1e-6,  # unit quaternions; kg m^2, entries O(1e-3 .. 10)
  "cvel": 1e-5, "qfrc_bias": 1e-4, "actuator_force": 1e-4, "qfrc_smooth": 1e-4,  # rad/s | m/s; N m: sums of terms O(10 .. 100)
  "qacc_smooth": 1e-3, "qacc": 1e-3, "qfrc_constraint": 1e-3,  # rad/s^2, N m: solutions of M a = f with |f| O(100), 1 / M_ii up to 1e3
  "efc_J": 1e-6, "efc_pos": 1e-7, "qpos": 1e-6, "qvel": 1e-5,
}


def per_world_elem(a, b, atol):
  """Per world: the worst element's |a - b| / (atol + RTOL |b|); <= 1 means every element passes."""
  a = a.reshape(a.shape[0], -1).astype(np.float64)
  b = b.reshape(b.shape[0], -1).astype(np.float64)
  return (np.abs(a - b) / (atol + RTOL * np.abs(b))).max(axis=1)


def per_world_rel(a, b):
  a = a.reshape(a.shape[0], -1).astype(np.float64)
  b = b.reshape(b.shape[0], -1).astype(np.float64)
  return np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-6)


def randomize_model(sim: Simulation, ora: OracleSim, model, fields, seed: int) -> None:
  """Per-world values like the reference's startup events draw them (tracking_env_cfg.py:162-198,
  velocity_env_cfg.py:162-172): foot friction U(0.3, 1.2), torso com offset, joint zero offsets --
  the same numbers on both sides."""
  rng = np.random.default_rng(seed)
  n = sim.num_envs
  sim.expand_model_fields(list(fields))
  for f in fields:
    base = np.asarray(getattr(model, f), dtype=np.float64)
    new = np.broadcast_to(base, (n, *base.shape)).copy()
    if f == "geom_friction":
      new[:, :, 0] = rng.uniform(0.3, 1.2, (n, 1)) * np.ones(base.shape[0])
    elif f == "body_ipos":
      root = int(model.jnt_bodyid[0])  # the floating base link
      new[:, root] += rng.uniform(-0.05, 0.05, (n, 3)) * np.array([0.5, 1.0, 1.0])
    elif f == "qpos0":
      new[:, 7:] += rng.uniform(-0.01, 0.01, (n, base.size - 7))
    elif f == "dof_frictionloss":  # what randomize_field("dof_frictionloss") writes: friction-loss rows on about half of the joints
      new = rng.uniform(0.02, 0.5, new.shape) * (rng.random(new.shape) < 0.5)
      new[:, :6] = 0.0
    else:
      raise ValueError(f)
    getattr(sim.model, f)[:] = torch.from_numpy(new.astype(np.float32)).to(sim.data.qpos.device)
    # the oracle gets the values the device holds (fp32-rounded), in its own precision
    ora.expand_model_field(f)[:] = getattr(sim.model, f).cpu().numpy().astype(ora.real)
  sim.create_graph()


def terrain_height(model, xy: np.ndarray, radius: float) -> np.ndarray:
  """Highest terrain-box top within `radius` (in x and y) of each point; boxes are axis aligned."""
  pos, size, mat = np.asarray(model.tbox_pos), np.asarray(model.tbox_size), np.asarray(model.tbox_mat).reshape(-1, 9)
  assert np.allclose(mat, np.eye(3).reshape(1, 9)), "terrain_height assumes axis-aligned terrain boxes"
  inside = (np.abs(xy[:, None, 0] - pos[None, :, 0]) <= size[None, :, 0] + radius) & (np.abs(xy[:, None, 1] - pos[None, :, 1]) <= size[None, :, 1] + radius)
  top = np.where(inside, (pos[:, 2] + size[:, 2])[None, :], -np.inf)
  return top.max(axis=1)


def scene_report(scene: str, n: int = 1024, control_steps: int = 25, precision: str = "f64", seed: int = 123,
                 expand: tuple = (), flags: dict | None = None, spread: float | None = None, settle: int = 15) -> dict:
  """Roll `n` worlds of `scene` for `control_steps` control steps on the GPU, then compare one
  forward() and one step() from the reached states with the oracle.  Returns a dict of statistics
  (per field: median / p99 / max of the per-world relative error over worlds with identical
  contact and row counts) plus the counts themselves.

  `spread` (terrain scenes): after the rollout every robot is put back into its keyframe pose at a
  random point within +-spread metres of its spawn origin, just above the highest box under its
  footprint, and `settle` control steps of zero actions follow -- so the compared states stand, lean
  and fall on stair treads and edges instead of the flat spawn platforms."""
  cores = os.cpu_count() or 8
  model = robots.load_model(scene)
  njmax = 300 if "velocity" in scene else 250
  flags = flags or {}
  sim = Simulation(n, SimulationCfg(njmax=njmax, use_graph=False, **flags), model, "cuda:0")
  oflags = (2 if flags.get("literal_termination") else 0) | (4 if flags.get("warmstart_at_advance") else 0)
  ora = OracleSim(model, n, njmax=njmax, precision=precision, flags=oflags, ls_parallel=sim.ls_parallel)  # the same line search on both sides
  if expand:
    randomize_model(sim, ora, model, expand, seed + 1)
  scale = g1_action_scale(model) if scene.startswith("g1") else go1_action_scale(model)
  if scene == "g1_tracking_flat":
    # BASELINE config 4 under its OWN reset distribution: random phases of a synthetic motion written
    # the way MotionCommand._resample_command writes them, the task's anchor terminations, 6-component pushes, 10 s episodes
    ev = TRACKING_TASK_EVENTS["g1"]
    roll = PhysicsRollout(sim, action_scale=scale, seed=seed, min_height=-1.0e9, fused_reset=False, motion=synthetic_motion(model),
                          motion_reset=ev["motion_reset"], push=ev["push"], episode_length_s=ev["episode_length_s"])
  else:
    roll = PhysicsRollout(sim, action_scale=scale, seed=seed, min_height=0.3 if scene.startswith("g1") else 0.15)
  nreset = 0
  for _ in range(control_steps):
    nreset += int(roll.step(roll.random_action()).sum())
  if spread is not None and model.nterrain:
    q = roll._sample_reset_qpos(n)
    q[:, 0:2] += (torch.rand((n, 2), device=q.device, generator=roll.gen) * 2 - 1) * spread
    h = terrain_height(model, q[:, 0:2].cpu().numpy(), 0.35)
    q[:, 2] = torch.from_numpy(h.astype(np.float32)).to(q.device) + float(roll.key_qpos[2]) + 0.01
    sim.data.qpos[:] = q
    sim.data.qvel[:] = 0.0
    sim.data.qacc_warmstart[:] = 0.0
    sim.forward()
    for _ in range(settle):
      nreset += int(roll.step(torch.zeros((n, model.nu), device=q.device)).sum())
  sim.data.ctrl[:] = roll.default_joint + roll.random_action() * roll.action_scale
  for f in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(ora, f)[:] = getattr(sim.data, f).cpu().numpy().astype(ora.real)
  ws = sim.data.qacc_warmstart.clone()
  sim.forward()
  ora.forward(nthread=cores)
  torch.cuda.synchronize()
  nefc_g, nefc_o = sim.data.nefc.cpu().numpy().ravel(), ora.nefc.ravel()
  ncon_g, ncon_o = sim.data.ncon.cpu().numpy().ravel(), ora.ncon.ravel()
  same = (nefc_g == nefc_o) & (ncon_g == ncon_o)
  out: dict = {
    "scene": scene, "n": n, "control_steps": control_steps, "precision": precision, "resets": nreset,
    "nefc_mean": float(nefc_o.mean()), "nefc_max": int(nefc_o.max()), "ncon_mean": float(ncon_o.mean()),
    "same_counts": int(same.sum()),
    "same_sensordata": int((sim.data.sensordata.cpu().numpy() == ora.sensordata.astype(np.float32)).all(axis=1).sum()),
    "overflow_gpu": int((sim.data.overflow != 0).sum()), "overflow_oracle": int((ora.overflow != 0).sum()),
    "fields": {}, "elem": {},
  }
  nv = model.nv
  # terrain contacts (geom2 is a terrain box) and how many worlds have one that is not a flat-top contact
  if model.nterrain:
    cg = sim.data.contact_geom.cpu().numpy().reshape(n, -1, 2)
    fr = sim.data.contact_frame.cpu().numpy().reshape(n, -1, 9)
    valid = np.arange(cg.shape[1])[None, :] < ncon_g[:, None]
    tb = np.isin(cg[:, :, 1], np.asarray(model.tbox_geom)) & valid
    edge = tb & (np.abs(fr[:, :, 2]) < 0.999)  # contact normal not vertical: an edge / side face / corner
    out["worlds_with_terrain_contact"] = int(tb.any(axis=1).sum())
    out["worlds_with_edge_contact"] = int(edge.any(axis=1).sum())
  # worlds with a DEEP SELF-PENETRATION: two robot geoms more than 5 mm inside each other (the tracking task's reset poses put thin
  # foot capsules into one another: two capsule axes a fraction of a millimetre apart give a contact normal that fp32 resolves to
  # 1e-3 at best -- ADVICE round 3: classify them explicitly instead of widening the literals for everyone)
  cgeom = sim.data.contact_geom.cpu().numpy().reshape(n, -1, 2)
  cdist = sim.data.contact_dist.cpu().numpy().reshape(n, -1)
  cvalid = np.arange(cgeom.shape[1])[None, :] < ncon_g[:, None]
  ns = int(model.nstaticgeom)
  deep = (cvalid & (cgeom[:, :, 0] >= ns) & (cgeom[:, :, 1] >= ns) & (cdist < -0.005)).any(axis=1)
  out["deep_self_penetration"] = int((deep & same).sum())
  out["regular"] = {}  # per field: max over the worlds WITHOUT such a contact whose Newton iteration did not end at its cap
  per_world_err: dict = {}
  for name in KIN + VEL + ROWS + SOLVE:
    g = getattr(sim.data, name).cpu().numpy()
    o = getattr(ora, name)
    if name in ROWS:  # row arrays: only the first nefc rows are defined
      w = nv if name == "efc_J" else 1
      rows = np.arange(njmax)[None, :] < nefc_o[:, None]
      mask = np.repeat(rows, w, axis=1) if w > 1 else rows
      g = np.where(mask, g.reshape(n, -1), 0)
      o = np.where(mask, o.reshape(n, -1), 0)
    e_all = per_world_rel(g, o)
    e = e_all[same]
    out["fields"][name] = (float(np.median(e)), float(np.percentile(e, 99)), float(e.max()))
    per_world_err[name] = e_all
    if name in ATOL:
      el = per_world_elem(g, o, ATOL[name])[same]
      out["elem"][name] = (float(np.median(el)), float(np.percentile(el, 99)), float(el.max()), float((el <= 1.0).mean()))
    if name == "qacc":
      qacc_err = per_world_rel(g, o)
    if name == "efc_pos":  # a distance near zero: the meaningful error is absolute (metres)
      a = np.abs(g.reshape(n, -1).astype(np.float64) - o.reshape(n, -1)).max(axis=1)[same]
      out["fields"]["efc_pos_abs_m"] = (float(np.median(a)), float(np.percentile(a, 99)), float(a.max()))
  niter_g, niter_o = sim.data.solver_niter.cpu().numpy().ravel(), ora.solver_niter.ravel()
  out["niter_gpu"] = (float(niter_g.mean()), int(niter_g.max()))
  out["niter_oracle"] = (float(niter_o.mean()), int(niter_o.max()))
  # Why is a world's qacc off by more than north_star's 1e-5?    Either side's Newton iteration
  # ended at the iteration cap (the iterate then depends on rounding), the two sides ended on different active sets
  # (a row whose J a - aref sits at 0 to rounding), or they took a different number of iterations (a termination test
  # decided by rounding).  What is left is plain fp32 noise of a converged solve: `unexplained`.
  cap = int(model.opt.iterations)
  rows = np.arange(njmax)[None, :] < nefc_o[:, None]
  act_g = (sim.data.efc_force.cpu().numpy().reshape(n, -1) != 0) & rows
  act_o = (ora.efc_force.reshape(n, -1) != 0) & rows
  capped = (niter_g >= cap) | (niter_o >= cap)
  actdiff = (act_g != act_o).any(axis=1)
  iterdiff = niter_g != niter_o
  off = same & (qacc_err > 1e-5)
  explained = capped | actdiff | iterdiff
  out["qacc_off"] = {
    "above_1e-5": int(off.sum()), "capped": int((off & capped).sum()), "active_set_differs": int((off & ~capped & actdiff).sum()),
    "niter_differs": int((off & ~capped & ~actdiff & iterdiff).sum()), "unexplained": int((off & ~explained).sum()),
    "unexplained_max": float(qacc_err[off & ~explained].max()) if (off & ~explained).any() else 0.0,
    "explained_max": float(qacc_err[off & explained].max()) if (off & explained).any() else 0.0,
    "worlds_capped": int((same & capped).sum()), "worlds_active_set_differs": int((same & actdiff).sum()),
  }
  # one step from the same state and warm start
  sim.data.qacc_warmstart[:] = ws
  ora.qacc_warmstart[:] = ws.cpu().numpy().astype(ora.real)
  sim.step()
  ora.step(1, nthread=cores)
  torch.cuda.synchronize()
  for f in STEP:
    e_all = per_world_rel(getattr(sim.data, f).cpu().numpy(), getattr(ora, f))
    e = e_all[same]
    out["fields"]["step_" + f] = (float(np.median(e)), float(np.percentile(e, 99)), float(e.max()))
    per_world_err["step_" + f] = e_all
    el = per_world_elem(getattr(sim.data, f).cpu().numpy(), getattr(ora, f), ATOL[f])[same]
    out["elem"]["step_" + f] = (float(np.median(el)), float(np.percentile(el, 99)), float(el.max()), float((el <= 1.0).mean()))
  regular = same & ~deep & ~capped
  out["regular_worlds"] = int(regular.sum())
  for name, e_all in per_world_err.items():
    out["regular"][name] = float(e_all[regular].max()) if regular.any() else 0.0
  del sim, roll, ora
  torch.cuda.empty_cache()
  return out


def format_report(r: dict) -> str:
  lines = [
    f"== {r['scene']}: {r['n']} worlds after {r['control_steps']} control steps of random actions ({r['resets']} resets on the way), "
    f"oracle {r['precision']}; nefc mean {r['nefc_mean']:.1f} max {r['nefc_max']}, ncon mean {r['ncon_mean']:.1f}",
    f"   identical contact / row counts: {r['same_counts']} of {r['n']} worlds (fp32 vs fp64 can flip a contact that sits exactly on its margin)",
    f"   sensordata identical: {r['same_sensordata']} of {r['n']}; capacity overflow flags: gpu {r['overflow_gpu']}, oracle {r['overflow_oracle']} worlds",
  ]
  if "worlds_with_edge_contact" in r:
    lines.append(f"   worlds with a terrain contact: {r['worlds_with_terrain_contact']}; with a non-vertical (edge / side / corner) terrain contact: {r['worlds_with_edge_contact']}")
  lines.append(f"   {'field':18s} {'median':>10s} {'p99':>10s} {'max':>10s}   (relative error per world, worlds with identical counts)")
  for k, (md, p99, mx) in r["fields"].items():
    lines.append(f"   {k:18s} {md:10.2e} {p99:10.2e} {mx:10.2e}")
  lines.append(f"   {'element-wise':18s} {'median':>10s} {'p99':>10s} {'max':>10s} {'worlds ok':>10s}   (worst element's |gpu - oracle| / (atol + 1e-5 |oracle|) per world; <= 1 passes)")
  for k, (md, p99, mx, ok) in r.get("elem", {}).items():
    lines.append(f"   {k:18s} {md:10.2e} {p99:10.2e} {mx:10.2e} {ok:10.4f}   atol {ATOL[k.replace('step_', '')]:.0e}")
  if "regular" in r and r["regular"]:
    reg = r["regular"]
    lines.append(f"   worlds with a robot-robot contact deeper than 5 mm: {r['deep_self_penetration']}; worst of the {r.get('regular_worlds', 0)} worlds without one and with a Newton iteration below its cap: efc_J {reg['efc_J']:.2e}, qacc {reg['qacc']:.2e}, "
                 f"qfrc_constraint {reg['qfrc_constraint']:.2e}, step_qpos {reg['step_qpos']:.2e}, step_qvel {reg['step_qvel']:.2e}")
  if "qacc_off" in r:
    q = r["qacc_off"]
    lines.append(f"   qacc off by more than 1e-5 in {q['above_1e-5']} worlds: {q['capped']} at the Newton iteration cap, {q['active_set_differs']} with a different final "
                 f"active set, {q['niter_differs']} with a different iteration count (worst of these {q['explained_max']:.2e}); unexplained {q['unexplained']} (worst {q['unexplained_max']:.2e}); "
                 f"all worlds: {q['worlds_capped']} capped, {q['worlds_active_set_differs']} with different active sets")
  lines.append(f"   Newton iterations: gpu mean {r['niter_gpu'][0]:.2f} (max {r['niter_gpu'][1]}), oracle mean {r['niter_oracle'][0]:.2f} (max {r['niter_oracle'][1]})")
  return "\n".join(lines)


if __name__ == "__main__":
  N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
  scenes = sys.argv[2].split(",") if len(sys.argv) > 2 else ALL_SCENES
  steps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
  prec = sys.argv[4] if len(sys.argv) > 4 else "f64"
  for scene in scenes:
    exp = ("geom_friction", "body_ipos", "qpos0") if scene == "g1_tracking_flat" else ("geom_friction",)
    print(format_report(scene_report(scene, N, steps, prec, expand=exp, spread=3.5 if scene.endswith("rough") else None)), flush=True)
