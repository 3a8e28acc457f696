"""A ``mujoco``-shaped face of this package's model compiler, so that the reference's Scene / Entity / task
configuration code -- which builds its model through ``mujoco.MjSpec`` -- runs UNMODIFIED on top of
``mjlab_amd`` (SURVEY.md Appendix D lists the surface; call sites: scene/scene.py:25-147,
entity/entity.py:120-214,588-652, terrains/terrain_importer.py:71-163, utils/spec_config.py, utils/spec.py,
sim/sim.py:22-82, envs/manager_based_rl_env.py:37).

``install()`` registers this module as ``sys.modules["mujoco"]`` (only when the real wheel is absent or
``force=True``).  What is here is the model-building API the physics path consumes: ``MjSpec`` (from_file /
from_string / attach / add_actuator / add_sensor / add_key / compile and the element lists), the ``Mjs*`` element
views (the dataclasses of ``mjlab_amd.mjcf`` themselves: attribute names are mujoco's), the enums with the numeric
values of the pinned build (reference typings/mujoco/_enums.pyi), ``MjModel`` = ``mjlab_amd.mjcf.Model``, a host
``MjData`` holding the qpos0 state.  Rendering-only calls (textures, materials, lights, cameras) are recorded and
ignored; anything else a caller reaches for resolves to an inert placeholder type so that annotations such as
``mujoco.MjvScene`` evaluate, and calling a placeholder function raises.
"""

from __future__ import annotations

import enum
import sys
import types
from pathlib import Path
from typing import Any

import numpy as np

from . import mjcf
from .mjcf import Model as MjModel  # noqa: F401  (the compiled host model carries mjModel's field names)
from .mjcf import SpecBody as MjsBody
from .mjcf import SpecGeom as MjsGeom
from .mjcf import SpecJoint as MjsJoint
from .mjcf import SpecSite as MjsSite

__version__ = "3.3.7-mjlab_amd-shim"  # reference pin: mujoco==3.3.7.dev811775910 (pyproject.toml:96)


def _enum(name: str, **members: int):
  return enum.IntEnum(name, members)


# numeric values: reference typings/mujoco/_enums.pyi (relied upon e.g. by utils/mujoco.py:21,28)
mjtJoint = _enum("mjtJoint", mjJNT_FREE=0, mjJNT_BALL=1, mjJNT_SLIDE=2, mjJNT_HINGE=3)
mjtGeom = _enum("mjtGeom", mjGEOM_PLANE=0, mjGEOM_HFIELD=1, mjGEOM_SPHERE=2, mjGEOM_CAPSULE=3, mjGEOM_ELLIPSOID=4, mjGEOM_CYLINDER=5,
                mjGEOM_BOX=6, mjGEOM_MESH=7, mjGEOM_SDF=8, mjGEOM_ARROW=100, mjGEOM_ARROW1=101, mjGEOM_ARROW2=102, mjGEOM_LINE=103,
                mjGEOM_LINEBOX=104, mjGEOM_FLEX=105, mjGEOM_SKIN=106, mjGEOM_LABEL=107, mjGEOM_TRIANGLE=108, mjGEOM_NONE=1001)
mjtTrn = _enum("mjtTrn", mjTRN_JOINT=0, mjTRN_JOINTINPARENT=1, mjTRN_SLIDERCRANK=2, mjTRN_TENDON=3, mjTRN_SITE=4, mjTRN_BODY=5, mjTRN_UNDEFINED=1000)
mjtGain = _enum("mjtGain", mjGAIN_FIXED=0, mjGAIN_AFFINE=1, mjGAIN_MUSCLE=2, mjGAIN_USER=3)
mjtBias = _enum("mjtBias", mjBIAS_NONE=0, mjBIAS_AFFINE=1, mjBIAS_MUSCLE=2, mjBIAS_USER=3)
mjtDyn = _enum("mjtDyn", mjDYN_NONE=0, mjDYN_INTEGRATOR=1, mjDYN_FILTER=2, mjDYN_FILTEREXACT=3, mjDYN_MUSCLE=4, mjDYN_USER=5)
mjtLimited = _enum("mjtLimited", mjLIMITED_FALSE=0, mjLIMITED_TRUE=1, mjLIMITED_AUTO=2)
mjtSensor = _enum("mjtSensor", mjSENS_TOUCH=0, mjSENS_ACCELEROMETER=1, mjSENS_VELOCIMETER=2, mjSENS_GYRO=3, mjSENS_FORCE=4, mjSENS_TORQUE=5,
                  mjSENS_FRAMEPOS=26, mjSENS_FRAMEQUAT=27, mjSENS_FRAMEXAXIS=28, mjSENS_FRAMEYAXIS=29, mjSENS_FRAMEZAXIS=30,
                  mjSENS_FRAMELINVEL=31, mjSENS_FRAMEANGVEL=32, mjSENS_FRAMELINACC=33, mjSENS_FRAMEANGACC=34, mjSENS_SUBTREECOM=35,
                  mjSENS_SUBTREELINVEL=36, mjSENS_SUBTREEANGMOM=37, mjSENS_CONTACT=42)
mjtObj = _enum("mjtObj", mjOBJ_UNKNOWN=0, mjOBJ_BODY=1, mjOBJ_XBODY=2, mjOBJ_JOINT=3, mjOBJ_DOF=4, mjOBJ_GEOM=5, mjOBJ_SITE=6, mjOBJ_CAMERA=7,
               mjOBJ_LIGHT=8, mjOBJ_ACTUATOR=19, mjOBJ_SENSOR=20, mjOBJ_KEY=24)
mjtSolver = _enum("mjtSolver", mjSOL_PGS=0, mjSOL_CG=1, mjSOL_NEWTON=2)
mjtCone = _enum("mjtCone", mjCONE_PYRAMIDAL=0, mjCONE_ELLIPTIC=1)
mjtIntegrator = _enum("mjtIntegrator", mjINT_EULER=0, mjINT_RK4=1, mjINT_IMPLICIT=2, mjINT_IMPLICITFAST=3)
mjtJacobian = _enum("mjtJacobian", mjJAC_DENSE=0, mjJAC_SPARSE=1, mjJAC_AUTO=2)
mjtState = _enum("mjtState", mjSTATE_TIME=1, mjSTATE_QPOS=2, mjSTATE_QVEL=4, mjSTATE_ACT=8, mjSTATE_WARMSTART=16, mjSTATE_CTRL=32,
                 mjSTATE_PHYSICS=14, mjSTATE_FULLPHYSICS=4111)
mjtTexture = _enum("mjtTexture", mjTEXTURE_2D=0, mjTEXTURE_CUBE=1, mjTEXTURE_SKYBOX=2)
mjtBuiltin = _enum("mjtBuiltin", mjBUILTIN_NONE=0, mjBUILTIN_GRADIENT=1, mjBUILTIN_CHECKER=2, mjBUILTIN_FLAT=3)
mjtMark = _enum("mjtMark", mjMARK_NONE=0, mjMARK_EDGE=1, mjMARK_CROSS=2, mjMARK_RANDOM=3)
mjtLightType = _enum("mjtLightType", mjLIGHT_SPOT=0, mjLIGHT_DIRECTIONAL=1, mjLIGHT_POINT=2, mjLIGHT_IMAGE=3)
mjtCamLight = _enum("mjtCamLight", mjCAMLIGHT_FIXED=0, mjCAMLIGHT_TRACK=1, mjCAMLIGHT_TRACKCOM=2, mjCAMLIGHT_TARGETBODY=3, mjCAMLIGHT_TARGETBODYCOM=4)
mjtTextureRole = _enum("mjtTextureRole", mjTEXROLE_USER=0, mjTEXROLE_RGB=1)
mjtCamera = _enum("mjtCamera", mjCAMERA_FREE=0, mjCAMERA_TRACKING=1, mjCAMERA_FIXED=2, mjCAMERA_USER=3)
mjtCatBit = _enum("mjtCatBit", mjCAT_STATIC=1, mjCAT_DYNAMIC=2, mjCAT_DECOR=4, mjCAT_ALL=7)


class _Record:
  """A rendering-only spec element (texture, material, light, camera, frame): attributes are kept, nothing is compiled."""

  def __init__(self, **kw: Any) -> None:
    self.__dict__.update(kw)


class MjsActuator:
  """``spec.add_actuator(...)`` result (utils/spec_config.py:441-453): ``gainprm`` / ``biasprm`` are written element-wise."""

  def __init__(self, name: str = "", target: str = "", trntype: int = 0, gaintype: int = 0, biastype: int = 0, dyntype: int = 0,
               inheritrange: float = 0.0, forcerange=(0.0, 0.0), ctrlrange=(0.0, 0.0), gear=(1.0, 0, 0, 0, 0, 0), **_: Any) -> None:
    self.name, self.target = name, target
    self.trntype, self.gaintype, self.biastype, self.dyntype = trntype, gaintype, biastype, dyntype
    self.inheritrange = float(inheritrange)
    self.forcerange = np.array(forcerange, dtype=np.float64)
    self.ctrlrange = np.array(ctrlrange, dtype=np.float64)
    self.gear = np.array(gear, dtype=np.float64)
    self.gainprm = np.zeros(10)
    self.gainprm[0] = 1.0  # mujoco's default gain
    self.biasprm = np.zeros(10)
    self.id = -1
    self.joint = target  # mjcf.SpecActuator's name for it


class MjsSensor:
  def __init__(self, name: str = "", type: int = 0, objtype: int = 0, objname: str = "", reftype: int | None = None, refname: str | None = None,
               intprm=(0, 0, 0), **_: Any) -> None:
    self.name, self.type, self.objtype, self.objname = name, int(type), int(objtype), objname
    self.reftype = None if reftype is None else int(reftype)
    self.refname = refname
    ip = list(intprm) + [0] * (3 - len(intprm))
    self.intprm = np.array(ip[:3], dtype=np.int64)
    self.id = -1


class MjsKey:
  def __init__(self, name: str = "", qpos=(), qvel=None, ctrl=None, **_: Any) -> None:
    self.name = name
    self.qpos = np.asarray(qpos, dtype=np.float64)
    self.qvel = None if qvel is None else np.asarray(qvel, dtype=np.float64)
    self.ctrl = None if ctrl is None else np.asarray(ctrl, dtype=np.float64)
    self.id = -1


class _Option:
  """``spec.option``: attribute bag with mujoco's names (sim/sim.py:66-82 sets them one by one)."""

  def __init__(self, base: mjcf.Option) -> None:
    self.__dict__.update(base.__dict__)
    self.jacobian = mjtJacobian.mjJAC_AUTO


class MjSpec:
  """The editable model description (``mujoco.MjSpec``) backed by ``mjlab_amd.mjcf.Spec``."""

  def __init__(self) -> None:
    self._spec = mjcf.Spec()
    self.option = _Option(self._spec.option)
    self.stat = _Record(extent=None, meansize=None, center=None)
    self.visual = _Record()
    self.assets: dict[str, bytes] = {}
    self.meshdir = ""
    self.modelname = "model"
    self._actuators: list[MjsActuator] = []
    self._sensors: list[MjsSensor] = []
    self._keys: list[MjsKey] = []
    self._records: list[_Record] = []  # textures / materials / lights / cameras / frames
    self._children: list[tuple["MjSpec", str]] = []  # attached child specs (their elements stay listed there too)
    self._prefix = ""  # set when this spec is attached somewhere: own names carry it from then on
    _install_body_extras()

  # -- construction ----------------------------------------------------------------------------------------------
  @classmethod
  def from_file(cls, path: str) -> "MjSpec":
    out = cls.from_string(Path(path).read_text())
    return out

  @classmethod
  def from_string(cls, xml: str, *_: Any, **__: Any) -> "MjSpec":
    import xml.etree.ElementTree as ET

    out = cls()
    root = ET.fromstring(xml)
    out._spec = mjcf.Spec.from_string(xml)
    out.option = _Option(out._spec.option)
    out.modelname = out._spec.modelname
    comp = root.find("compiler")
    if comp is not None and comp.get("meshdir"):
      out.meshdir = comp.get("meshdir")
    st = root.find("statistic")
    if st is not None:
      for k in ("extent", "meansize"):
        if st.get(k):
          setattr(out.stat, k, float(st.get(k)))
    return out

  # -- element lists (ordered like mujoco's: depth-first body order) -------------------------------------------------
  @property
  def worldbody(self) -> MjsBody:
    return self._spec.world

  @property
  def bodies(self) -> list[MjsBody]:
    return self._spec.bodies

  @property
  def joints(self) -> list[MjsJoint]:
    return self._spec.joints

  @property
  def geoms(self) -> list[MjsGeom]:
    return self._spec.geoms

  @property
  def sites(self) -> list[MjsSite]:
    return self._spec.sites

  @property
  def actuators(self) -> list[MjsActuator]:
    return list(self._actuators)

  @property
  def sensors(self) -> list[MjsSensor]:
    return list(self._sensors)

  @property
  def keys(self) -> list[MjsKey]:
    return list(self._keys)

  @property
  def tendons(self) -> list:
    return []

  @property
  def cameras(self) -> list:
    return [r for r in self._records if getattr(r, "kind", "") == "camera"]

  @property
  def lights(self) -> list:
    return [r for r in self._records if getattr(r, "kind", "") == "light"]

  def body(self, name: str) -> MjsBody | None:
    return next((b for b in self.bodies if b.name == name), None)

  def joint(self, name: str) -> MjsJoint | None:
    return next((j for j in self.joints if j.name == name), None)

  def geom(self, name: str) -> MjsGeom | None:
    return next((g for g in self.geoms if g.name == name), None)

  def site(self, name: str) -> MjsSite | None:
    return next((s for s in self.sites if s.name == name), None)

  # -- editing ---------------------------------------------------------------------------------------------------------
  def add_actuator(self, **kw: Any) -> MjsActuator:
    a = MjsActuator(**kw)
    self._actuators.append(a)
    return a

  def add_sensor(self, **kw: Any) -> MjsSensor:
    s = MjsSensor(**kw)
    self._sensors.append(s)
    return s

  def add_key(self, **kw: Any) -> MjsKey:
    k = MjsKey(**kw)
    self._keys.append(k)
    return k

  def add_texture(self, **kw: Any) -> _Record:
    r = _Record(kind="texture", **kw)
    self._records.append(r)
    return r

  def add_material(self, **kw: Any) -> _Record:
    r = _Record(kind="material", textures=[""] * 10, **kw)
    self._records.append(r)
    return r

  def add_exclude(self, bodyname1: str = "", bodyname2: str = "", **_: Any) -> None:
    self._spec.excludes.append((bodyname1, bodyname2))

  def attach(self, child: "MjSpec", prefix: str = "", frame: Any = None, **_: Any) -> None:
    """``spec.attach(child_spec, prefix=, frame=)`` (scene/scene.py:137-138,146-147): the child's world contents move under
    this world BY REFERENCE -- the child spec keeps listing the same element objects, whose names now carry the prefix
    and whose ids become valid when THIS spec compiles (entity/entity.py:589-600 relies on exactly that)."""
    fpos = np.zeros(3) if frame is None else np.asarray(getattr(frame, "pos", np.zeros(3)), dtype=np.float64)
    if np.any(fpos != 0):
      raise NotImplementedError("attach(): frames with an offset are not supported")
    pw, cw = self._spec.world, child._spec.world
    # names a child-side reference can resolve to inside the child; a reference to something outside it (the velocity task's
    # foot sensors name the scene's "terrain" body: tasks/velocity/config/g1/rough_env_cfg.py:22-24) keeps its name
    inside = {e.name for seq in (child._spec.bodies, child._spec.geoms, child._spec.sites, child._spec.joints) for e in seq if e.name}
    for b in child._spec.bodies:
      if b is cw:
        continue
      b.name = prefix + b.name if b.name else b.name
      for seq in (b.joints, b.geoms, b.sites):
        for e in seq:
          if e.name:
            e.name = prefix + e.name
    for seq, dst in ((cw.geoms, pw.geoms), (cw.sites, pw.sites)):
      for e in seq:
        if e.name:
          e.name = prefix + e.name
        e.body = pw
        dst.append(e)
    for b in cw.children:
      b.parent = pw
      pw.children.append(b)
    self._spec.excludes += [(prefix + a, prefix + b) for a, b in child._spec.excludes]
    for a in child._actuators:
      a.name, a.target = prefix + a.name, prefix + a.target
      a.joint = a.target
      self._actuators.append(a)
    for s in child._sensors:
      s.name = prefix + s.name
      if s.objname in inside:
        s.objname = prefix + s.objname
      if s.refname and s.refname in inside:
        s.refname = prefix + s.refname
      self._sensors.append(s)
    for k in child._keys:
      k.name = prefix + k.name
    self._children.append((child, prefix))
    child._prefix = prefix
    self._records += child._records

  # -- compile ----------------------------------------------------------------------------------------------------------
  def _export_option(self) -> None:
    o, dst = self.option, self._spec.option
    for k in dst.__dict__:
      v = getattr(o, k)
      dst.__dict__[k] = tuple(float(x) for x in v) if k == "gravity" else (int(v) if isinstance(v, enum.IntEnum) else v)

  def compile(self) -> MjModel:
    sp = self._spec
    self._export_option()
    joints = {j.name: j for j in sp.joints}
    sp.actuators = []
    for a in self._actuators:
      if int(a.trntype) != mjtTrn.mjTRN_JOINT or int(a.gaintype) != mjtGain.mjGAIN_FIXED or int(a.biastype) not in (mjtBias.mjBIAS_AFFINE, mjtBias.mjBIAS_NONE):
        raise NotImplementedError(f"actuator '{a.name}': only joint transmission with fixed gain and affine bias is supported")
      j = joints.get(a.target)
      if j is None:
        raise ValueError(f"actuator '{a.name}': joint '{a.target}' not found")
      ctrlrange = None
      if a.inheritrange > 0:  # ctrlrange := the joint range scaled about its centre (MuJoCo's inheritrange)
        mid, half = 0.5 * (j.range[0] + j.range[1]), 0.5 * (j.range[1] - j.range[0]) * a.inheritrange
        ctrlrange = (mid - half, mid + half)
      elif a.ctrlrange[0] < a.ctrlrange[1]:
        ctrlrange = (float(a.ctrlrange[0]), float(a.ctrlrange[1]))
      frc = (float(a.forcerange[0]), float(a.forcerange[1])) if a.forcerange[0] < a.forcerange[1] else None
      sa = mjcf.SpecActuator(a.name, a.target, float(a.gainprm[0]), tuple(float(x) for x in a.biasprm[:3]), frc, ctrlrange, gear=float(a.gear[0]))
      sp.actuators.append(sa)
    sp.sensors = []
    for s in self._sensors:
      if s.type != mjtSensor.mjSENS_CONTACT:
        raise NotImplementedError(f"sensor '{s.name}': only contact sensors are on the physics path (type {s.type})")
      sp.sensors.append(mjcf.SpecSensor(s.name, s.objtype, s.objname, s.reftype, s.refname if s.reftype is not None else None, tuple(int(x) for x in s.intprm)))
    sp.keys = []
    model_keys = self._full_keys()
    sp.keys = model_keys
    m = sp.compile()
    for seq_src, seq_dst in ((self._actuators, sp.actuators), (self._sensors, sp.sensors)):
      for a, b in zip(seq_src, seq_dst, strict=True):
        a.id = b.id
    m.stat_extent = self.stat.extent
    return m

  def _full_keys(self) -> list[mjcf.SpecKey]:
    """Keyframes of this spec and of attached children, each padded to the whole model like mujoco's attach does:
    joints outside the child keep qpos0 / zero ctrl."""
    sp = self._spec
    joints = sp.joints
    qadr, a = {}, 0
    for j in joints:
      qadr[id(j)] = a
      a += 7 if j.type == mjcf.JNT_FREE else 1
    nq = a
    qpos0 = np.zeros(nq)
    for j in joints:
      if j.type == mjcf.JNT_FREE:
        qpos0[qadr[id(j)] : qadr[id(j)] + 7] = np.concatenate([j.body.pos, mjcf.quat_normalize(j.body.quat)])
      else:
        qpos0[qadr[id(j)]] = j.ref
    act_index = {id(a_): i for i, a_ in enumerate(self._actuators)}
    out: list[mjcf.SpecKey] = []

    def emit(owner: "MjSpec", key: MjsKey, name: str) -> None:
      own_joints = owner._spec.joints
      own_nq = sum(7 if j.type == mjcf.JNT_FREE else 1 for j in own_joints)
      q = qpos0.copy()
      if len(key.qpos) == own_nq:
        o = 0
        for j in own_joints:
          w = 7 if j.type == mjcf.JNT_FREE else 1
          q[qadr[id(j)] : qadr[id(j)] + w] = key.qpos[o : o + w]
          o += w
      elif len(key.qpos) not in (0, nq):
        raise ValueError(f"key '{name}': qpos has {len(key.qpos)} entries, the spec has nq = {own_nq}")
      elif len(key.qpos) == nq:
        q = key.qpos.copy()
      ctrl = np.zeros(len(self._actuators))
      if key.ctrl is not None and len(key.ctrl):
        idx = [act_index[id(a_)] for a_ in owner._actuators]
        if len(idx) != len(key.ctrl):
          raise ValueError(f"key '{name}': ctrl has {len(key.ctrl)} entries for {len(idx)} actuators")
        ctrl[idx] = key.ctrl
      out.append(mjcf.SpecKey(name, q, None, ctrl))

    for k in self._keys:
      emit(self, k, k.name)
    for child, _prefix in self._children:
      for k in child._keys:
        emit(child, k, k.name)
    return out

  def to_xml(self) -> str:
    raise NotImplementedError("MjSpec.to_xml is not provided by the mjlab_amd shim")

  @staticmethod
  def to_zip(spec: "MjSpec", file: Any) -> None:
    raise NotImplementedError("MjSpec.to_zip is not provided by the mjlab_amd shim")


_BODY_EXTRAS_DONE = False


def _install_body_extras() -> None:
  """Rendering-only body methods (lights, cameras, frames): accepted and recorded on the body, never compiled."""
  global _BODY_EXTRAS_DONE
  if _BODY_EXTRAS_DONE:
    return
  _BODY_EXTRAS_DONE = True

  def add_light(self, **kw: Any) -> _Record:
    return _Record(kind="light", **kw)

  def add_camera(self, **kw: Any) -> _Record:
    return _Record(kind="camera", **kw)

  def add_frame(self, pos=(0.0, 0.0, 0.0), quat=(1.0, 0.0, 0.0, 0.0), **kw: Any) -> _Record:
    return _Record(kind="frame", pos=np.array(pos, dtype=np.float64), quat=np.array(quat, dtype=np.float64), **kw)

  MjsBody.add_light, MjsBody.add_camera, MjsBody.add_frame = add_light, add_camera, add_frame


class MjData:
  """Host ``mjData`` at the qpos0 state (sim/sim.py:106-107; viewers and exporters read it)."""

  def __init__(self, model: MjModel) -> None:
    self.qpos = np.array(model.qpos0, dtype=np.float64)
    self.qvel = np.zeros(model.nv)
    self.act = np.zeros(int(getattr(model, "na", 0)))
    self.ctrl = np.zeros(model.nu)
    self.xfrc_applied = np.zeros((model.nbody, 6))
    self.time = 0.0


def mj_forward(model: MjModel, data: MjData) -> None:
  """Host-side forward kinematics (the only derived host quantities anything on the path reads are body poses)."""
  kin = mjcf.kinematics_np(model, data.qpos)
  data.xpos, data.xquat = kin["xpos"], kin["xquat"]


def mj_resetData(model: MjModel, data: MjData) -> None:
  data.qpos[:] = model.qpos0
  data.qvel[:] = 0
  data.ctrl[:] = 0
  data.time = 0.0


def mj_resetDataKeyframe(model: MjModel, data: MjData, key: int) -> None:
  data.qpos[:] = model.key_qpos[key]
  data.qvel[:] = model.key_qvel[key]
  data.ctrl[:] = model.key_ctrl[key]
  data.time = 0.0


def mj_stateSize(model: MjModel, spec: int) -> int:
  n = 0
  for bit, size in ((mjtState.mjSTATE_TIME, 1), (mjtState.mjSTATE_QPOS, model.nq), (mjtState.mjSTATE_QVEL, model.nv), (mjtState.mjSTATE_ACT, int(getattr(model, "na", 0)))):
    if spec & bit:
      n += size
  return n


def mj_name2id(model: MjModel, objtype: int, name: str) -> int:
  kind = {mjtObj.mjOBJ_BODY: "body", mjtObj.mjOBJ_XBODY: "body", mjtObj.mjOBJ_JOINT: "joint", mjtObj.mjOBJ_GEOM: "geom", mjtObj.mjOBJ_SITE: "site",
          mjtObj.mjOBJ_ACTUATOR: "actuator", mjtObj.mjOBJ_SENSOR: "sensor", mjtObj.mjOBJ_KEY: "key"}[int(objtype)]
  try:
    return model.names[kind].index(name)
  except ValueError:
    return -1


def mj_id2name(model: MjModel, objtype: int, i: int) -> str:
  kind = {mjtObj.mjOBJ_BODY: "body", mjtObj.mjOBJ_JOINT: "joint", mjtObj.mjOBJ_GEOM: "geom", mjtObj.mjOBJ_SITE: "site", mjtObj.mjOBJ_ACTUATOR: "actuator",
          mjtObj.mjOBJ_SENSOR: "sensor"}[int(objtype)]
  return model.names[kind][i]


def _placeholder(name: str):
  if name[:1].isupper() or name.startswith("mjt"):
    return type(name, (), {"__init__": lambda self, *a, **k: None, "__doc__": f"placeholder for mujoco.{name} (not on the physics path)"})

  def missing(*_a: Any, **_k: Any):
    raise NotImplementedError(f"mujoco.{name} is not provided by the mjlab_amd shim (it is not on the physics path)")

  missing.__name__ = name
  return missing


def __getattr__(name: str):  # PEP 562: anything else resolves to an inert placeholder, remembered
  if name.startswith("__"):
    raise AttributeError(name)
  val = _placeholder(name)
  globals()[name] = val
  return val


class _ViewerModule(types.ModuleType):
  """``mujoco.viewer``: interactive viewers are outside the physics path; names resolve, calls raise."""

  def __getattr__(self, name: str):
    if name.startswith("__"):
      raise AttributeError(name)
    val = _placeholder(name)
    setattr(self, name, val)
    return val


viewer = _ViewerModule("mujoco.viewer")


def install(force: bool = False) -> types.ModuleType:
  """Make ``import mujoco`` resolve to this module (no-op when the real wheel is importable, unless ``force``)."""
  me = sys.modules[__name__]
  if not force:
    try:
      import importlib.util

      if "mujoco" not in sys.modules and importlib.util.find_spec("mujoco") is not None:
        import mujoco  # the real one

        return mujoco
    except (ImportError, ValueError):
      pass
  cur = sys.modules.get("mujoco")
  if cur is not None and cur is not me and not force and hasattr(cur, "MjSpec") and not isinstance(getattr(cur, "MjSpec"), type(None)) \
     and getattr(cur, "__file__", None):
    return cur
  sys.modules["mujoco"] = me
  me.__path__ = []  # `import mujoco.viewer` (reference viewer/native.py:11) needs a package
  sys.modules["mujoco.viewer"] = viewer
  return me
