"""Golden vectors for row N1: THE REFERENCE'S OWN reader / writer classes running over this
repository's boundary objects.

The reference reads the simulation through ``EntityData`` (src/mjlab/entity/data.py:34-516: every
property indexes ``sim.data.<field>`` / ``sim.model.<field>`` with an ``EntityIndexing``) and writes
per-world model values through ``randomize_field`` (src/mjlab/envs/mdp/events.py:212-265:
``env.sim.model.<field>[env_grid, ids] = ...``).  Both are imported here unmodified (third-party
packages that are not installed are stubbed, see tools/make_reference_pins.py) and bound to

  * ``mjlab_amd.sim_data.Bridge`` objects -- the class of ``Simulation.data`` / ``Simulation.model`` --
    over host tensors with exactly the device-side shapes and broadcast views
    (``device_state.shape_view``; float model fields ``(nworld, n...)`` with stride 0 until expanded,
    int topology fields without a world dimension),
  * the index tables of ``mjlab_amd.entity_data.entity_indexing`` (built from the compiled model).

States are live rollout states of the CPU oracle (G1 velocity-flat, random actions).  The reference
cannot travel to the GPU box, so what it computes here is committed as
tests/golden/boundary_reference.npz; tests/test_gpu_reference_boundary.py loads the same states into
a real ``Simulation`` and requires ``mjlab_entity_readback`` / ``sim.data`` to reproduce every
property, and the per-world friction table the reference's ``randomize_field`` drew to change the
physics world by world.  tests/test_reference_over_boundary.py (CPU, needs /root/reference) re-runs
this file's computation and checks the committed vectors against it.

Run in the build container:  python tools/make_boundary_golden.py
"""

from __future__ import annotations

import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

from make_reference_pins import REF, _StubFinder  # noqa: E402

from mjlab_amd import device_state, native, robots  # noqa: E402
from mjlab_amd.entity_data import entity_indexing  # noqa: E402
from mjlab_amd.sim_data import Bridge  # noqa: E402

SCENE, NWORLD, CONTROL_STEPS, SEED = "g1_velocity_flat", 32, 12, 7
DST = ROOT / "tests" / "golden" / "boundary_reference.npz"
# every EntityData property the reference defines (entity/data.py:192-516) except the ones that raise
# upstream whatever the engine: joint_torques (NotImplementedError, :327) and root_com_pose_w /
# root_com_pos_w / root_com_quat_w (:212-219 multiplies a (nworld, 4) quaternion with `body_iquat[None]`,
# (1, nworld, 4): quat_mul rejects the shape mismatch) -- RAISING below pins that they behave the same here
PROPERTIES = (
  "root_link_pose_w", "root_link_vel_w", "root_com_vel_w", "body_link_pose_w", "body_link_vel_w",
  "body_com_pose_w", "body_com_vel_w", "body_external_wrench", "geom_pose_w", "geom_vel_w", "site_pose_w", "site_vel_w",
  "joint_pos", "joint_vel", "joint_acc", "actuator_force", "generalized_force",
  "root_link_pos_w", "root_link_quat_w", "root_link_lin_vel_w", "root_link_ang_vel_w",
  "root_com_lin_vel_w", "root_com_ang_vel_w", "body_link_pos_w", "body_link_quat_w", "body_link_lin_vel_w",
  "body_link_ang_vel_w", "body_com_pos_w", "body_com_quat_w", "body_com_lin_vel_w", "body_com_ang_vel_w",
  "body_external_force", "body_external_torque", "geom_pos_w", "geom_quat_w", "geom_lin_vel_w", "geom_ang_vel_w",
  "site_pos_w", "site_quat_w", "site_lin_vel_w", "site_ang_vel_w", "projected_gravity_b", "heading_w",
  "root_link_lin_vel_b", "root_link_ang_vel_b", "root_com_lin_vel_b", "root_com_ang_vel_b",
)  # fmt: skip
RAISING = ("joint_torques", "root_com_pose_w", "root_com_pos_w", "root_com_quat_w")


def import_reference():
  """-> (EntityData, EntityIndexing, events module) of the reference, third parties stubbed."""
  sys.dont_write_bytecode = True  # /root/reference is read-only by contract: no __pycache__ there
  if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _StubFinder())
  if str(REF / "src") not in sys.path:
    sys.path.insert(0, str(REF / "src"))
  from mjlab.entity.data import EntityData
  from mjlab.entity.entity import EntityIndexing
  from mjlab.envs.mdp import events

  return EntityData, EntityIndexing, events


def host_bridges(model, nworld: int, data_arrays: dict[str, np.ndarray] | None = None):
  """``Simulation.model`` / ``Simulation.data``-shaped Bridges over HOST tensors: the layouts come
  from the library (native.layouts), the shapes from device_state.shape_view, float model fields are
  stride-0 broadcasts of one shared copy -- the same construction as device_state.upload_model /
  alloc_data, minus the GPU."""
  mf, df, _, _ = native.layouts()
  from mjlab_amd import _abi

  mview: dict[str, torch.Tensor] = {}
  for f in mf:
    if f.kind == "i":
      mview[f.name] = torch.from_numpy(_abi.model_int_array(model, f.name))
    else:
      t = torch.from_numpy(np.ascontiguousarray(getattr(model, f.name), dtype=np.float32)).unsqueeze(0)
      mview[f.name] = t.expand(nworld, *t.shape[1:])
  ncon, njmax = _abi.default_capacities(model, None, 300)
  dview: dict[str, torch.Tensor] = {}
  for f in df:
    n = _abi.count_of(f.count, model, ncon, njmax)
    flat = torch.zeros((nworld, n * f.ncol), dtype=torch.int32 if f.kind == "i" else torch.float32)
    if data_arrays is not None and f.name in data_arrays:
      flat[:] = torch.from_numpy(np.asarray(data_arrays[f.name]).reshape(nworld, -1).astype(np.int32 if f.kind == "i" else np.float32))
    dview[f.name] = device_state.shape_view(f, flat, n)
  dview["act"] = torch.zeros((nworld, 0))
  scal = {k: int(getattr(model, k)) for k in ("nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite")}
  return Bridge("sim.model", mview, {**scal, "nworld": nworld}), Bridge("sim.data", dview, {"nworld": nworld})


def expand_host_field(model_bridge: Bridge, name: str) -> torch.Tensor:
  """Host-side stand-in of Simulation.expand_model_fields: the broadcast view becomes real per-world storage."""
  t = model_bridge._tensors[name].clone().contiguous()
  model_bridge._tensors[name] = t
  return t


def oracle_states(model):
  """Live rollout states: NWORLD G1s under random actions, as the fp64 oracle computes them."""
  from mjlab_amd.rollout import g1_action_scale
  from oracle.oracle import OracleSim

  ora = OracleSim(model, NWORLD, njmax=300, precision="f64")
  rng = np.random.default_rng(SEED)
  ora.reset(key=0)
  yaw = rng.uniform(-3.14, 3.14, NWORLD)
  ora.qpos[:, 0:2] += rng.uniform(-0.5, 0.5, (NWORLD, 2))
  ora.qpos[:, 3], ora.qpos[:, 4:6], ora.qpos[:, 6] = np.cos(yaw / 2), 0.0, np.sin(yaw / 2)
  jn = model.actuator_trnid[:, 0]
  default = model.key_qpos[0][model.jnt_qposadr[jn]]
  scale = g1_action_scale(model)
  for _ in range(CONTROL_STEPS):
    ora.ctrl[:] = default + scale * rng.uniform(-1, 1, (NWORLD, model.nu))
    ora.step(4, nthread=8)
  ora.xfrc_applied[:, 5, :] = rng.normal(0, 3.0, (NWORLD, 6))  # something for body_external_wrench to show
  ora.qfrc_applied[:, :6] = rng.normal(0, 1.0, (NWORLD, 6))
  inputs = {f: getattr(ora, f).copy() for f in ("qpos", "qvel", "ctrl", "qacc_warmstart", "xfrc_applied", "qfrc_applied")}
  ora.forward(nthread=8)
  return ora, inputs


def reference_entity_data(EntityData, EntityIndexing, model, mb: Bridge, db: Bridge, nworld: int):
  ix = entity_indexing(model, "cpu")
  indexing = EntityIndexing(joints=(), geoms=(), sites=(), actuators=None, **ix)
  z = lambda *s: torch.zeros(s)  # noqa: E731
  nj = int(ix["joint_ids"].numel())
  return EntityData(
    indexing=indexing, data=db, model=mb, device="cpu",
    default_root_state=z(nworld, 13), default_joint_pos=z(nworld, nj), default_joint_vel=z(nworld, nj),
    default_joint_stiffness=z(nworld, nj), default_joint_damping=z(nworld, nj), default_joint_pos_limits=z(nworld, nj, 2),
    joint_pos_limits=z(nworld, nj, 2), soft_joint_pos_limits=z(nworld, nj, 2),
    gravity_vec_w=torch.tensor([0.0, 0.0, -1.0]).repeat(nworld, 1), forward_vec_b=torch.tensor([1.0, 0.0, 0.0]).repeat(nworld, 1),
    is_fixed_base=False, is_articulated=True, is_actuated=True,
  )  # fmt: skip


def mock_env(model, mb: Bridge, nworld: int):
  """What randomize_field touches of ``env``: num_envs, device, sim.model, scene[name].indexing."""
  ix = SimpleNamespace(**entity_indexing(model, "cpu"))
  asset = SimpleNamespace(indexing=ix)
  return SimpleNamespace(num_envs=nworld, device="cpu", sim=SimpleNamespace(model=mb), scene={"robot": asset})


def foot_geom_local_ids(model) -> list[int]:
  """Local (entity) ids of the foot collision geoms: the ones the velocity task randomises
  (tasks/velocity/velocity_env_cfg.py:162-172, geom_names = foot collision geoms)."""
  ix = entity_indexing(model, "cpu")
  names = model.names["geom"]
  gids = ix["geom_ids"].tolist()
  return [k for k, g in enumerate(gids) if "foot" in names[g] and "collision" in names[g]]


def randomize_with_reference(events, model, mb: Bridge, nworld: int) -> dict[str, torch.Tensor]:
  """The reference's randomize_field on four fields, seeded; returns the resulting per-world tables."""
  env = mock_env(model, mb, nworld)
  torch.manual_seed(SEED)
  out = {}
  feet = foot_geom_local_ids(model)
  cfg = lambda **k: SimpleNamespace(name="robot", joint_ids=slice(None), body_ids=slice(None), geom_ids=slice(None), site_ids=slice(None), **k)  # noqa: E731
  fr = cfg()
  fr.geom_ids = feet
  expand_host_field(mb, "geom_friction")
  events.randomize_field(env, None, "geom_friction", ranges=(0.3, 1.2), operation="abs", asset_cfg=fr)
  out["geom_friction"] = mb._tensors["geom_friction"].clone()
  expand_host_field(mb, "body_mass")
  events.randomize_field(env, None, "body_mass", ranges=(0.9, 1.1), operation="scale", asset_cfg=cfg())
  out["body_mass"] = mb._tensors["body_mass"].clone()
  tb = cfg()
  tb.body_ids = [0]
  expand_host_field(mb, "body_ipos")
  events.randomize_field(env, None, "body_ipos", ranges={0: (-0.025, 0.025), 1: (-0.05, 0.05), 2: (-0.05, 0.05)}, operation="add", asset_cfg=tb)
  out["body_ipos"] = mb._tensors["body_ipos"].clone()
  expand_host_field(mb, "dof_damping")
  events.randomize_field(env, torch.arange(0, nworld, 2), "dof_damping", ranges=(0.0, 0.5), operation="add", asset_cfg=cfg())
  out["dof_damping"] = mb._tensors["dof_damping"].clone()
  return out


def compute() -> dict[str, np.ndarray]:
  EntityData, EntityIndexing, events = import_reference()
  model = robots.load_model(SCENE)
  ora, inputs = oracle_states(model)
  mb, db = host_bridges(model, NWORLD, ora.dfield)
  ed = reference_entity_data(EntityData, EntityIndexing, model, mb, db, NWORLD)
  out: dict[str, np.ndarray] = {f"in_{k}": v.astype(np.float32) for k, v in inputs.items()}
  for p in PROPERTIES:
    out[f"ed_{p}"] = getattr(ed, p).detach().clone().numpy()
  for name, t in ed.sensor_data.items():
    out[f"ed_sensor_{name}"] = t.clone().numpy()
  for k, t in randomize_with_reference(events, model, mb, NWORLD).items():
    out[f"dr_{k}"] = t.numpy()
  return out


if __name__ == "__main__":
  vec = compute()
  np.savez_compressed(DST, **vec)
  print("wrote", DST, f"({len(vec)} arrays, {DST.stat().st_size / 1024:.0f} KiB)")
