"""Host model from a ``mujoco.MjModel`` (the type the reference hands to ``Simulation``:
reference src/mjlab/sim/sim.py:97-99, built by ``Scene.compile()`` -> ``spec.compile()``).

``mjlab_amd.mjcf.Model`` names its arrays after mjModel's, so the conversion is a field-by-field
copy of the subset the physics step uses, a check that the model stays inside what the HIP kernels
implement, and ``mjcf.finalize_topology`` for the derived tables (tree levels, ancestor-dof masks,
candidate pair list, terrain grid).  Only attribute access is used -- ``mujoco`` itself is not
imported, so the function also accepts any object that exposes the same attributes (that is how
tests/test_from_mujoco.py exercises it in a container without the ``mujoco`` wheel).

Names are read from the raw ``names`` byte blob and the ``name_*adr`` arrays (present on every
MjModel), so no ``mj_id2name`` call is needed.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from . import mjcf
from .mjcf import Model, Option

_BASE_INT = (
  "body_parentid", "body_rootid", "body_weldid", "body_jntnum", "body_jntadr", "body_dofnum", "body_dofadr", "body_geomnum", "body_geomadr",
  "jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited", "dof_bodyid", "dof_jntid", "dof_parentid",
  "geom_type", "geom_bodyid", "geom_contype", "geom_conaffinity", "geom_condim", "geom_priority", "site_bodyid",
  "actuator_trnid", "actuator_ctrllimited", "actuator_forcelimited",
  "sensor_type", "sensor_objtype", "sensor_objid", "sensor_reftype", "sensor_refid", "sensor_intprm", "sensor_dim", "sensor_adr",
)  # fmt: skip
_BASE_REAL = (
  "qpos0", "qpos_spring", "body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_subtreemass", "body_inertia", "body_invweight0",
  "jnt_pos", "jnt_axis", "jnt_range", "jnt_margin", "jnt_stiffness", "jnt_solref", "jnt_solimp",
  "dof_armature", "dof_damping", "dof_frictionloss", "dof_invweight0", "dof_solref", "dof_solimp", "dof_M0",
  "geom_size", "geom_pos", "geom_quat", "geom_friction", "geom_solref", "geom_solimp", "geom_solmix", "geom_margin", "geom_gap",
  "geom_rbound", "geom_rgba", "site_pos", "site_quat",
  "actuator_gainprm", "actuator_biasprm", "actuator_ctrlrange", "actuator_forcerange", "actuator_gear",
  "key_qpos", "key_qvel", "key_ctrl",
)  # fmt: skip
_SCALARS = ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "nsensor", "nsensordata", "nkey", "nexclude")
_NAME_ADR = {"body": "name_bodyadr", "joint": "name_jntadr", "geom": "name_geomadr", "site": "name_siteadr",
             "actuator": "name_actuatoradr", "sensor": "name_sensoradr", "key": "name_keyadr"}  # fmt: skip
_UNSUPPORTED_COUNTS = ("neq", "ntendon", "nmocap", "nflex", "nhfield", "nmesh", "npair")  # explicit <pair>s included


def _names(mjm: Any, adr_field: str) -> list[str]:
  blob = bytes(mjm.names)
  return [blob[a : blob.index(b"\0", a)].decode() for a in np.asarray(getattr(mjm, adr_field)).tolist()]


def model_from_mujoco(mjm: Any) -> Model:
  """``mujoco.MjModel`` (or any object with the same attributes) -> ``mjcf.Model``."""
  for n in _UNSUPPORTED_COUNTS:
    if int(getattr(mjm, n, 0)) != 0:
      raise NotImplementedError(f"{n} = {int(getattr(mjm, n))}: not implemented by the HIP physics step")
  m = Model()
  for n in _SCALARS:
    setattr(m, n, int(getattr(mjm, n, 0)))
  for n in _BASE_INT:
    if hasattr(mjm, n):
      setattr(m, n, np.ascontiguousarray(np.asarray(getattr(mjm, n)), dtype=np.int32))
  for n in _BASE_REAL:
    if hasattr(mjm, n):
      setattr(m, n, np.ascontiguousarray(np.asarray(getattr(mjm, n)), dtype=np.float64))
  o, mo = mjm.opt, Option()
  mo.timestep, mo.impratio, mo.tolerance, mo.ls_tolerance = float(o.timestep), float(o.impratio), float(o.tolerance), float(o.ls_tolerance)
  mo.gravity = tuple(float(g) for g in np.asarray(o.gravity))
  mo.iterations, mo.ls_iterations = int(o.iterations), int(o.ls_iterations)
  mo.integrator, mo.cone, mo.solver = int(o.integrator), int(o.cone), int(o.solver)
  m.opt = mo
  m.meaninertia = float(mjm.stat.meaninertia)
  m.names = {kind: _names(mjm, adr) for kind, adr in _NAME_ADR.items()}

  # ---- inside the implemented subset? (the same limits the MJCF compiler enforces)
  if (m.jnt_type == mjcf.JNT_BALL).any():
    raise NotImplementedError("ball joints are not implemented")
  if m.nsensor and (m.sensor_type != mjcf.SENS_CONTACT).any():
    raise NotImplementedError("only contact sensors (mjSENS_CONTACT) are implemented")
  col = (m.geom_contype != 0) | (m.geom_conaffinity != 0)
  bad = col & np.isin(m.geom_type, (mjcf.GEOM_HFIELD, mjcf.GEOM_MESH))  # cylinders / ellipsoids: rejected only inside a candidate pair
  if bad.any():
    raise NotImplementedError(f"colliding geom types {sorted(set(m.geom_type[bad].tolist()))} are not implemented")
  if not np.isin(m.geom_condim[col], (1, 3)).all():
    raise NotImplementedError("only condim 1 and 3 are implemented")
  if m.nu:
    trntype = np.asarray(getattr(mjm, "actuator_trntype", np.zeros(m.nu)))
    gaintype = np.asarray(getattr(mjm, "actuator_gaintype", np.zeros(m.nu)))
    biastype = np.asarray(getattr(mjm, "actuator_biastype", np.ones(m.nu)))
    dyntype = np.asarray(getattr(mjm, "actuator_dyntype", np.zeros(m.nu)))
    # mjTRN_JOINT = 0, mjGAIN_FIXED = 0, mjBIAS_NONE = 0 / mjBIAS_AFFINE = 1, mjDYN_NONE = 0
    if (trntype != 0).any() or (gaintype != 0).any() or (biastype > 1).any() or (dyntype != 0).any():
      raise NotImplementedError("only joint-transmission actuators with fixed gain and affine bias are implemented")

  # excluded body pairs: signature = (body1 << 16) + body2 (mjModel.exclude_signature)
  excl = set()
  m.exclude_signature = np.ascontiguousarray(np.asarray(getattr(mjm, "exclude_signature", np.zeros(0, np.int32))), dtype=np.int32)
  for sig in m.exclude_signature.tolist():
    b1, b2 = int(sig) >> 16, int(sig) & 0xFFFF
    excl.add((min(b1, b2), max(b1, b2)))
  mjcf.finalize_topology(m, excl)
  return m
