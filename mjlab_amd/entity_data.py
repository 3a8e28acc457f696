"""Fused read-back of the entity-level quantities the reference derives in ``EntityData``
(reference src/mjlab/entity/data.py) -- SURVEY.md section 8f row 1, an *extension*: the
reference computes each of these with 3-6 small torch kernels every time a property is
touched; here one kernel launch (``mjlab_entity_readback``) refreshes all of them.

Property names, shapes and conventions are the reference's (velocities are [linear, angular],
poses are [pos, quat wxyz]); ``update()`` must be called after ``sim.step()`` /
``sim.forward()`` (the reference recomputes lazily on access instead).
"""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import native
from .mjcf import JNT_FREE
from .sim import Simulation


class _View(ctypes.Structure):
  _fields_ = [
    ("nbody", ctypes.c_int), ("njoint", ctypes.c_int), ("root_body_id", ctypes.c_int), ("pad_", ctypes.c_int),
    ("body_ids", ctypes.c_void_p), ("joint_q_adr", ctypes.c_void_p), ("joint_v_adr", ctypes.c_void_p),
    ("gravity_vec_w", ctypes.c_float * 3), ("forward_vec_b", ctypes.c_float * 3),
    ("body_link_pose_w", ctypes.c_void_p), ("body_link_vel_w", ctypes.c_void_p),
    ("body_com_pose_w", ctypes.c_void_p), ("body_com_vel_w", ctypes.c_void_p),
    ("root_derived", ctypes.c_void_p),
    ("joint_pos", ctypes.c_void_p), ("joint_vel", ctypes.c_void_p), ("joint_acc", ctypes.c_void_p),
  ]  # fmt: skip


def entity_indexing(model, device: str | torch.device = "cpu", root_body: int | None = None) -> dict:
  """Index tables of one entity, named and typed like the reference's ``EntityIndexing``
  (src/mjlab/entity/entity.py:20-47, built at :588-652): what its ``EntityData`` and
  ``randomize_field`` index ``sim.data`` / ``sim.model`` with.  The entity is the subtree of
  ``root_body`` (default: the last child of the world body).  Returned as a dict of keyword
  arguments, so ``EntityIndexing(bodies=..., joints=(), geoms=(), sites=(), actuators=None, **tables)``
  works on the reference side; ``bodies`` here is a tuple of objects with an ``id`` (the only
  attribute the reference reads, ``root_body_id``)."""
  from types import SimpleNamespace

  m = model
  if root_body is None:
    root_body = int(np.nonzero(np.asarray(m.body_parentid) == 0)[0][-1])
  body_ids = np.arange(root_body, root_body + int(m.body_subtreenum[root_body]))
  inb = lambda ids: np.isin(np.asarray(ids), body_ids)  # noqa: E731
  it = lambda a: torch.tensor(np.asarray(a, dtype=np.int64), dtype=torch.int, device=device)  # noqa: E731
  jb = np.asarray(m.jnt_bodyid)
  jsel = [j for j in range(m.njnt) if inb(jb[j]) and m.jnt_type[j] != JNT_FREE]
  fsel = [j for j in range(m.njnt) if inb(jb[j]) and m.jnt_type[j] == JNT_FREE]
  fq = [a for j in fsel for a in range(int(m.jnt_qposadr[j]), int(m.jnt_qposadr[j]) + 7)]
  fv = [a for j in fsel for a in range(int(m.jnt_dofadr[j]), int(m.jnt_dofadr[j]) + 6)]
  sensors = {}
  for k, name in enumerate(m.names.get("sensor", [])):
    sensors[name.split("/")[-1]] = torch.arange(int(m.sensor_adr[k]), int(m.sensor_adr[k]) + int(m.sensor_dim[k]), dtype=torch.int, device=device)
  return {
    "bodies": tuple(SimpleNamespace(id=int(b)) for b in body_ids),
    "body_ids": it(body_ids),
    "geom_ids": it(np.nonzero(inb(m.geom_bodyid))[0]),
    "site_ids": it(np.nonzero(inb(m.site_bodyid))[0]),
    "ctrl_ids": it([a for a in range(m.nu) if inb(jb[m.actuator_trnid[a, 0]])]),
    "joint_ids": it(jsel),
    "joint_q_adr": it(np.asarray(m.jnt_qposadr)[jsel]),
    "joint_v_adr": it(np.asarray(m.jnt_dofadr)[jsel]),
    "free_joint_q_adr": it(fq),
    "free_joint_v_adr": it(fv),
    "sensor_adr": sensors,
  }


class EntityReadback:
  """Entity = the subtree of ``root_body`` (default: the last child of the world, i.e. the
  robot attached after the terrain, reference scene/scene.py:133-147)."""

  def __init__(self, sim: Simulation, root_body: int | None = None) -> None:
    m = sim.host_model
    if root_body is None:
      root_body = int(np.nonzero(np.asarray(m.body_parentid) == 0)[0][-1])
    nsub = int(m.body_subtreenum[root_body])
    body_ids = np.arange(root_body, root_body + nsub, dtype=np.int32)
    jsel = [j for j in range(m.njnt) if m.jnt_bodyid[j] in body_ids and m.jnt_type[j] != JNT_FREE]
    self.sim = sim
    self.body_ids = body_ids
    self.joint_ids = np.asarray(jsel, dtype=np.int32)
    dev = sim.data.qpos.device
    n = sim.num_envs
    self._idx = {
      "body_ids": torch.from_numpy(body_ids).to(dev),
      "joint_q_adr": torch.from_numpy(np.asarray(m.jnt_qposadr)[jsel].astype(np.int32)).to(dev),
      "joint_v_adr": torch.from_numpy(np.asarray(m.jnt_dofadr)[jsel].astype(np.int32)).to(dev),
    }
    nb, nj = len(body_ids), len(jsel)
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)  # noqa: E731
    self.body_link_pose_w, self.body_link_vel_w = z(n, nb, 7), z(n, nb, 6)
    self.body_com_pose_w, self.body_com_vel_w = z(n, nb, 7), z(n, nb, 6)
    self._root = z(n, 16)
    self.joint_pos, self.joint_vel, self.joint_acc = z(n, nj), z(n, nj), z(n, nj)
    v = _View()
    v.nbody, v.njoint, v.root_body_id = nb, nj, int(root_body)
    for k, t in self._idx.items():
      setattr(v, k, t.data_ptr())
    v.gravity_vec_w[:] = (0.0, 0.0, -1.0)
    v.forward_vec_b[:] = (1.0, 0.0, 0.0)
    for k in ("body_link_pose_w", "body_link_vel_w", "body_com_pose_w", "body_com_vel_w", "joint_pos", "joint_vel", "joint_acc"):
      setattr(v, k, getattr(self, k).data_ptr())
    v.root_derived = self._root.data_ptr()
    self._view = v

  def update(self) -> None:
    s = self.sim
    with torch.cuda.device(s._dev):
      native.check(
        s._lib.mjlab_entity_readback(ctypes.byref(s._m), ctypes.byref(s._d), ctypes.byref(self._view), s._stream()),
        "mjlab_entity_readback",
      )

  # root link = first entity body
  @property
  def root_link_pose_w(self) -> torch.Tensor:
    return self.body_link_pose_w[:, 0]

  @property
  def root_link_vel_w(self) -> torch.Tensor:
    return self.body_link_vel_w[:, 0]

  @property
  def root_com_pose_w(self) -> torch.Tensor:
    return self.body_com_pose_w[:, 0]

  @property
  def root_com_vel_w(self) -> torch.Tensor:
    return self.body_com_vel_w[:, 0]

  @property
  def projected_gravity_b(self) -> torch.Tensor:
    return self._root[:, 0:3]

  @property
  def heading_w(self) -> torch.Tensor:
    return self._root[:, 3]

  @property
  def root_link_lin_vel_b(self) -> torch.Tensor:
    return self._root[:, 4:7]

  @property
  def root_link_ang_vel_b(self) -> torch.Tensor:
    return self._root[:, 7:10]

  @property
  def root_com_lin_vel_b(self) -> torch.Tensor:
    return self._root[:, 10:13]

  @property
  def root_com_ang_vel_b(self) -> torch.Tensor:
    return self._root[:, 13:16]
