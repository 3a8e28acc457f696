"""Minimal MJCF -> mjModel compiler for the physics hot path.

The reference obtains its ``mjModel`` from ``mujoco.MjSpec.compile()``
(reference: src/mjlab/scene/scene.py:38-39, src/mjlab/sim/sim.py:97-107).  The
``mujoco`` package is not available on the MI355X image, so this module restates
the part of MuJoCo's model compiler that the benchmark scenes need:

* ``<compiler angle=radian autolimits=true>``, nested ``<default>`` classes with
  ``childclass``, bodies with ``<inertial>`` or with mass / inertia inferred from their
  primitive geoms (``mass`` / ``density``), ``<freejoint>``, hinge / slide joints, sphere / capsule (``fromto``) / box / plane geoms, visual mesh
  geoms (kept for id parity, never collide), sites, ``<contact><exclude>``;
* the spec edits mjlab applies (reference: src/mjlab/utils/spec_config.py:245-276
  collision attributes, :400-453 position actuators, :554-629 contact sensors);
* scene assembly: world + ``terrain`` body with a ground plane + robot attached
  with a ``"robot/"`` prefix (reference: src/mjlab/scene/scene.py:133-147,
  src/mjlab/terrains/terrain_importer.py:154-163);
* compile-time constants: subtree masses, ``dof_invweight0``,
  ``body_invweight0``, ``stat.meaninertia``, ``geom_rbound``, the static list of
  candidate collision pairs.

All host arithmetic is float64 (like ``mjModel``); the device copy is float32.
Field names follow ``mjModel`` (reference catalogue:
typings/mujoco/_structs.pyi:916ff) so that code written against the reference
reads the same.
"""

from __future__ import annotations

import math
import re
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any

import numpy as np

# mjtJoint / mjtGeom / misc enums (numeric values relied upon by the reference:
# src/mjlab/utils/mujoco.py:21,28).
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE = 0, 1, 2, 3
GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 4, 5, 6, 7
_GEOM_TYPES = {
  "plane": GEOM_PLANE,
  "hfield": GEOM_HFIELD,
  "sphere": GEOM_SPHERE,
  "capsule": GEOM_CAPSULE,
  "ellipsoid": GEOM_ELLIPSOID,
  "cylinder": GEOM_CYLINDER,
  "box": GEOM_BOX,
  "mesh": GEOM_MESH,
}
_JNT_TYPES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}

OBJ_BODY, OBJ_XBODY, OBJ_GEOM, OBJ_SITE = 1, 2, 5, 6  # mjtObj
SENS_CONTACT = 42  # mjtSensor.mjSENS_CONTACT of the pinned mujoco build (reference typings/mujoco/_enums.pyi:5166)

INT_EULER, INT_IMPLICITFAST = 0, 3  # mjtIntegrator
SOL_PGS, SOL_CG, SOL_NEWTON = 0, 1, 2  # mjtSolver
CONE_PYRAMIDAL, CONE_ELLIPTIC = 0, 1  # mjtCone

MJ_MINVAL = 1e-15


# ----------------------------------------------------------------------------
# small math helpers (float64, quaternions are w-x-y-z)
# ----------------------------------------------------------------------------


def quat_mul(a, b):
  aw, ax, ay, az = a
  bw, bx, by, bz = b
  return np.array(
    [
      aw * bw - ax * bx - ay * by - az * bz,
      aw * bx + ax * bw + ay * bz - az * by,
      aw * by - ax * bz + ay * bw + az * bx,
      aw * bz + ax * by - ay * bx + az * bw,
    ]
  )


def quat_to_mat(q):
  w, x, y, z = q
  return np.array(
    [
      [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
      [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
      [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ]
  )


def quat_normalize(q):
  q = np.asarray(q, dtype=np.float64)
  n = np.linalg.norm(q)
  if n < MJ_MINVAL:
    return np.array([1.0, 0.0, 0.0, 0.0])
  return q / n


def z_to_quat(vec):
  """Quaternion rotating (0,0,1) onto ``vec`` (MuJoCo compiler's fromto rule)."""
  v = np.asarray(vec, dtype=np.float64)
  v = v / np.linalg.norm(v)
  axis = np.cross([0.0, 0.0, 1.0], v)
  s = np.linalg.norm(axis)
  if s < 1e-10:
    axis = np.array([1.0, 0.0, 0.0])
  else:
    axis = axis / s
  ang = math.atan2(s, v[2])
  return np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)])


def _floats(s: str | None, n: int | None = None, default=None):
  if s is None:
    return None if default is None else np.array(default, dtype=np.float64)
  v = np.array([float(t) for t in s.split()], dtype=np.float64)
  if n is not None and len(v) < n and default is not None:
    out = np.array(default, dtype=np.float64)
    out[: len(v)] = v
    return out
  return v


# ----------------------------------------------------------------------------
# Spec tree
# ----------------------------------------------------------------------------


@dataclass
class SpecJoint:
  name: str
  type: int
  pos: np.ndarray
  axis: np.ndarray
  range: np.ndarray
  limited: bool
  armature: float = 0.0
  damping: float = 0.0
  frictionloss: float = 0.0
  stiffness: float = 0.0
  margin: float = 0.0
  ref: float = 0.0
  springref: float = 0.0
  solref: np.ndarray = field(default_factory=lambda: np.array([0.02, 1.0]))
  solimp: np.ndarray = field(default_factory=lambda: np.array([0.9, 0.95, 0.001, 0.5, 2.0]))
  solref_friction: np.ndarray = field(default_factory=lambda: np.array([0.02, 1.0]))
  solimp_friction: np.ndarray = field(default_factory=lambda: np.array([0.9, 0.95, 0.001, 0.5, 2.0]))
  body: "SpecBody | None" = None


@dataclass
class SpecGeom:
  name: str
  type: int
  size: np.ndarray
  pos: np.ndarray
  quat: np.ndarray
  contype: int = 1
  conaffinity: int = 1
  condim: int = 3
  priority: int = 0
  group: int = 0
  friction: np.ndarray = field(default_factory=lambda: np.array([1.0, 0.005, 0.0001]))
  solref: np.ndarray = field(default_factory=lambda: np.array([0.02, 1.0]))
  solimp: np.ndarray = field(default_factory=lambda: np.array([0.9, 0.95, 0.001, 0.5, 2.0]))
  solmix: float = 1.0
  margin: float = 0.0
  gap: float = 0.0
  rgba: np.ndarray = field(default_factory=lambda: np.array([0.5, 0.5, 0.5, 1.0]))
  body: "SpecBody | None" = None
  mass: float | None = None  # overrides density when given (MJCF geom/mass)
  density: float = 1000.0
  material: str | None = None  # (rendering only; the terrain generator reads it back: terrains/terrain_generator.py:215)


@dataclass
class SpecSite:
  name: str
  pos: np.ndarray
  quat: np.ndarray
  body: "SpecBody | None" = None


@dataclass
class SpecBody:
  name: str
  pos: np.ndarray
  quat: np.ndarray
  parent: "SpecBody | None" = None
  ipos: np.ndarray | None = None
  iquat: np.ndarray | None = None
  mass: float = 0.0
  inertia: np.ndarray | None = None
  joints: list[SpecJoint] = field(default_factory=list)
  geoms: list[SpecGeom] = field(default_factory=list)
  sites: list[SpecSite] = field(default_factory=list)
  children: list["SpecBody"] = field(default_factory=list)

  # ``mujoco.MjsBody``-style editing (keyword names of the reference's call sites: terrains/terrain_importer.py:113-120,
  # 157-163, terrains/utils.py:28-105, utils/spec_config.py:311-361); rendering-only arguments are accepted and dropped
  def add_body(self, name: str = "", pos=(0.0, 0.0, 0.0), quat=(1.0, 0.0, 0.0, 0.0), **_: Any) -> "SpecBody":
    b = SpecBody(name, np.array(pos, dtype=np.float64), np.array(quat, dtype=np.float64), parent=self)
    self.children.append(b)
    return b

  def add_geom(self, name: str = "", type: int = GEOM_SPHERE, size=(0.0, 0.0, 0.0), pos=(0.0, 0.0, 0.0), quat=(1.0, 0.0, 0.0, 0.0),
               rgba=None, material: str | None = None, **kw: Any) -> SpecGeom:
    sz = np.zeros(3)
    size = np.atleast_1d(np.asarray(size, dtype=np.float64))
    sz[: len(size)] = size
    keep = {k: v for k, v in kw.items() if k in SpecGeom.__dataclass_fields__}
    g = SpecGeom(name, int(type), sz, np.array(pos, dtype=np.float64), np.array(quat, dtype=np.float64), body=self, **keep)
    if rgba is not None:
      g.rgba = np.array(rgba, dtype=np.float64)
    g.material = material
    self.geoms.append(g)
    return g

  def add_site(self, name: str = "", pos=(0.0, 0.0, 0.0), quat=(1.0, 0.0, 0.0, 0.0), **_: Any) -> SpecSite:
    s = SpecSite(name, np.array(pos, dtype=np.float64), np.array(quat, dtype=np.float64), body=self)
    self.sites.append(s)
    return s


@dataclass
class SpecActuator:
  """Joint-transmission actuator with FIXED gain and AFFINE bias
  (reference: src/mjlab/utils/spec_config.py:441-453)."""

  name: str
  joint: str
  gainprm0: float
  biasprm: tuple[float, float, float]
  forcerange: tuple[float, float] | None
  ctrlrange: tuple[float, float] | None  # None -> unlimited
  gear: float = 1.0


@dataclass
class SpecSensor:
  """``mjSENS_CONTACT`` sensor (reference: src/mjlab/utils/spec_config.py:554-629)."""

  name: str
  objtype: int
  objname: str
  reftype: int | None
  refname: str | None
  intprm: tuple[int, int, int]  # [dataspec bitmask, reduce, num]


@dataclass
class SpecKey:
  name: str
  qpos: np.ndarray
  qvel: np.ndarray | None = None
  ctrl: np.ndarray | None = None


@dataclass
class Option:
  """mjOption subset; defaults are MuJoCo's, overridden like
  reference src/mjlab/sim/sim.py:43-82 (``MujocoCfg.edit_spec``)."""

  timestep: float = 0.002
  gravity: tuple[float, float, float] = (0.0, 0.0, -9.81)
  impratio: float = 1.0
  tolerance: float = 1e-8
  ls_tolerance: float = 0.01
  iterations: int = 100
  ls_iterations: int = 50
  integrator: int = INT_EULER
  cone: int = CONE_PYRAMIDAL
  solver: int = SOL_NEWTON


class Spec:
  """Editable model description (the subset of ``mujoco.MjSpec`` the path needs)."""

  def __init__(self) -> None:
    self.world = SpecBody("world", np.zeros(3), np.array([1.0, 0, 0, 0]))
    self.excludes: list[tuple[str, str]] = []
    self.actuators: list[SpecActuator] = []
    self.sensors: list[SpecSensor] = []
    self.keys: list[SpecKey] = []
    self.option = Option()
    self.modelname = "model"

  # -- construction --------------------------------------------------------

  @classmethod
  def from_file(cls, path: str | Path) -> "Spec":
    return cls.from_string(Path(path).read_text())

  @classmethod
  def from_string(cls, xml: str) -> "Spec":
    spec = cls()
    _MjcfParser(spec).parse(ET.fromstring(xml))
    return spec

  # -- queries -------------------------------------------------------------

  def _walk(self, body: SpecBody | None = None):
    body = body or self.world
    yield body
    for c in body.children:
      yield from self._walk(c)

  @property
  def bodies(self) -> list[SpecBody]:
    return list(self._walk())

  @property
  def joints(self) -> list[SpecJoint]:
    return [j for b in self._walk() for j in b.joints]

  @property
  def geoms(self) -> list[SpecGeom]:
    return [g for b in self._walk() for g in b.geoms]

  @property
  def sites(self) -> list[SpecSite]:
    return [s for b in self._walk() for s in b.sites]

  def body(self, name: str) -> SpecBody:
    for b in self._walk():
      if b.name == name:
        return b
    raise KeyError(f"body '{name}' not found")

  def joint(self, name: str) -> SpecJoint:
    for j in self.joints:
      if j.name == name:
        return j
    raise KeyError(f"joint '{name}' not found")

  def geom(self, name: str) -> SpecGeom:
    for g in self.geoms:
      if g.name == name:
        return g
    raise KeyError(f"geom '{name}' not found")

  # -- editing -------------------------------------------------------------

  def add_body(self, name: str, parent: SpecBody | None = None, pos=(0, 0, 0), quat=(1, 0, 0, 0)) -> SpecBody:
    parent = parent or self.world
    b = SpecBody(name, np.array(pos, dtype=np.float64), np.array(quat, dtype=np.float64), parent=parent)
    parent.children.append(b)
    return b

  def add_geom(self, body: SpecBody, name: str, type: int, size, pos=(0, 0, 0), quat=(1, 0, 0, 0), **kw) -> SpecGeom:
    sz = np.zeros(3)
    sz[: len(size)] = size
    g = SpecGeom(name, type, sz, np.array(pos, dtype=np.float64), np.array(quat, dtype=np.float64), body=body, **kw)
    body.geoms.append(g)
    return g

  def add_site(self, body: SpecBody, name: str, pos=(0, 0, 0), quat=(1, 0, 0, 0)) -> SpecSite:
    s = SpecSite(name, np.array(pos, dtype=np.float64), np.array(quat, dtype=np.float64), body=body)
    body.sites.append(s)
    return s

  def attach(self, child: "Spec", prefix: str = "") -> None:
    """Attach ``child``'s world children under this world (reference:
    src/mjlab/scene/scene.py:137-138,146-147 -- ``spec.attach(..., prefix=)``)."""
    for b in child._walk():
      if b is child.world:
        continue
      b.name = prefix + b.name
      for j in b.joints:
        j.name = prefix + j.name
      for g in b.geoms:
        if g.name:
          g.name = prefix + g.name
      for s in b.sites:
        if s.name:
          s.name = prefix + s.name
    for g in child.world.geoms:
      g.body = self.world
      if g.name:
        g.name = prefix + g.name
      self.world.geoms.append(g)
    for s in child.world.sites:
      s.body = self.world
      if s.name:
        s.name = prefix + s.name
      self.world.sites.append(s)
    for b in child.world.children:
      b.parent = self.world
      self.world.children.append(b)
    self.excludes += [(prefix + a, prefix + b) for a, b in child.excludes]
    for a in child.actuators:
      a.name = prefix + a.name
      a.joint = prefix + a.joint
      self.actuators.append(a)
    for s in child.sensors:
      s.name = prefix + s.name
      s.objname = prefix + s.objname
      if s.refname is not None:
        s.refname = prefix + s.refname
      self.sensors.append(s)
    # Keyframes of an attached child are re-expressed by the caller (the reference
    # keeps one key per entity; see robots.build_scene).
    child.world.children = []

  def compile(self) -> "Model":
    return _compile(self)


# ----------------------------------------------------------------------------
# MJCF parser (defaults classes, bodies, joints, geoms, sites, excludes)
# ----------------------------------------------------------------------------


class _MjcfParser:
  def __init__(self, spec: Spec) -> None:
    self.spec = spec
    self.defaults: dict[str, dict[str, dict[str, str]]] = {"main": {}}
    self.default_parent: dict[str, str | None] = {"main": None}
    self.autolimits = True
    self.angle_scale = math.pi / 180.0  # MJCF default is degrees
    self._anon = 0

  def parse(self, root: ET.Element) -> None:
    self.spec.modelname = root.get("model", "model")
    comp = root.find("compiler")
    if comp is not None:
      if comp.get("angle", "degree") == "radian":
        self.angle_scale = 1.0
      self.autolimits = comp.get("autolimits", "true") == "true"
      for unsupported in ("eulerseq", "coordinate", "inertiafromgeom", "fitaabb"):
        if comp.get(unsupported) is not None:
          raise NotImplementedError(f"<compiler {unsupported}=...> is not supported")
    for d in root.findall("default"):
      self._parse_default(d, None)
    opt = root.find("option")
    if opt is not None:
      o = self.spec.option
      if opt.get("timestep"):
        o.timestep = float(opt.get("timestep"))
      if opt.get("gravity"):
        o.gravity = tuple(_floats(opt.get("gravity")))
      if opt.get("integrator"):
        o.integrator = {"Euler": INT_EULER, "implicitfast": INT_IMPLICITFAST}[opt.get("integrator")]
      if opt.get("iterations"):
        o.iterations = int(opt.get("iterations"))
      if opt.get("ls_iterations"):
        o.ls_iterations = int(opt.get("ls_iterations"))
      if opt.get("tolerance"):
        o.tolerance = float(opt.get("tolerance"))
      if opt.get("cone"):
        o.cone = {"pyramidal": CONE_PYRAMIDAL, "elliptic": CONE_ELLIPTIC}[opt.get("cone")]
      if opt.get("impratio"):
        o.impratio = float(opt.get("impratio"))
    wb = root.find("worldbody")
    if wb is not None:
      self._parse_body_children(wb, self.spec.world, None)
    con = root.find("contact")
    if con is not None:
      for ex in con.findall("exclude"):
        self.spec.excludes.append((ex.get("body1"), ex.get("body2")))
      if con.find("pair") is not None:
        raise NotImplementedError("<contact><pair> is not supported")
    for tag in ("equality", "tendon"):
      if root.find(tag) is not None and len(root.find(tag)):
        raise NotImplementedError(f"<{tag}> is not supported")
    act = root.find("actuator")
    if act is not None and len(act):
      raise NotImplementedError("XML <actuator> is not supported; use robots.ActuatorCfg")

  # defaults ---------------------------------------------------------------

  def _parse_default(self, elem: ET.Element, parent: str | None) -> None:
    name = elem.get("class", "main" if parent is None else None)
    if name is None:
      raise ValueError("nested <default> needs a class")
    if name not in self.defaults:
      self.defaults[name] = {}
      self.default_parent[name] = parent
    for child in elem:
      if child.tag == "default":
        self._parse_default(child, name)
      else:
        self.defaults[name].setdefault(child.tag, {}).update(child.attrib)

  def _resolved(self, tag: str, cls: str | None, attrib: dict[str, str]) -> dict[str, str]:
    chain = []
    c = cls if cls is not None else "main"
    while c is not None:
      chain.append(c)
      c = self.default_parent.get(c)
    out: dict[str, str] = {}
    for c in reversed(chain):
      out.update(self.defaults.get(c, {}).get(tag, {}))
    out.update(attrib)
    return out

  # bodies -----------------------------------------------------------------

  def _orientation(self, a: dict[str, str]) -> np.ndarray:
    if "quat" in a:
      return quat_normalize(_floats(a["quat"]))
    for k in ("euler", "axisangle", "xyaxes", "zaxis"):
      if k in a:
        raise NotImplementedError(f"orientation attribute '{k}' is not supported")
    return np.array([1.0, 0.0, 0.0, 0.0])

  def _parse_body_children(self, elem: ET.Element, body: SpecBody, childclass: str | None) -> None:
    for child in elem:
      tag = child.tag
      if tag == "body":
        a = child.attrib
        nb = SpecBody(
          a.get("name") or self._auto("body"),
          _floats(a.get("pos"), default=[0, 0, 0]),
          self._orientation(a),
          parent=body,
        )
        body.children.append(nb)
        self._parse_body_children(child, nb, a.get("childclass", childclass))
      elif tag == "inertial":
        a = child.attrib
        if "fullinertia" in a:
          raise NotImplementedError("fullinertia is not supported")
        body.ipos = _floats(a.get("pos"), default=[0, 0, 0])
        body.iquat = self._orientation(a)
        body.mass = float(a["mass"])
        body.inertia = _floats(a["diaginertia"])
      elif tag in ("joint", "freejoint"):
        body.joints.append(self._parse_joint(child, childclass, body))
      elif tag == "geom":
        body.geoms.append(self._parse_geom(child, childclass, body))
      elif tag == "site":
        a = self._resolved("site", child.get("class", childclass), child.attrib)
        body.sites.append(
          SpecSite(
            a.get("name") or self._auto("site"),
            _floats(a.get("pos"), default=[0, 0, 0]),
            self._orientation(a),
            body=body,
          )
        )
      elif tag in ("light", "camera"):
        continue  # rendering only; not on the physics path
      else:
        raise NotImplementedError(f"<{tag}> inside <body> is not supported")

  def _auto(self, kind: str) -> str:
    self._anon += 1
    return ""

  def _parse_joint(self, elem: ET.Element, childclass: str | None, body: SpecBody) -> SpecJoint:
    if elem.tag == "freejoint":
      return SpecJoint(
        elem.get("name", ""), JNT_FREE, np.zeros(3), np.array([0.0, 0, 1]), np.zeros(2), False, body=body
      )
    a = self._resolved("joint", elem.get("class", childclass), elem.attrib)
    jtype = _JNT_TYPES[a.get("type", "hinge")]
    if jtype == JNT_BALL:
      raise NotImplementedError("ball joints are not supported")
    rng = _floats(a.get("range"), default=[0, 0])
    if jtype == JNT_HINGE:
      rng = rng * self.angle_scale
    if "limited" in a and a["limited"] != "auto":
      limited = a["limited"] == "true"
    else:
      limited = self.autolimits and ("range" in a) and rng[0] < rng[1]
    axis = _floats(a.get("axis"), default=[0, 0, 1])
    axis = axis / np.linalg.norm(axis)
    j = SpecJoint(
      a.get("name", ""),
      jtype,
      _floats(a.get("pos"), default=[0, 0, 0]) if jtype != JNT_FREE else np.zeros(3),
      axis,
      rng,
      limited,
      armature=float(a.get("armature", 0)),
      damping=float(a.get("damping", 0)),
      frictionloss=float(a.get("frictionloss", 0)),
      stiffness=float(a.get("stiffness", 0)),
      margin=float(a.get("margin", 0)),
      ref=float(a.get("ref", 0)) * (self.angle_scale if jtype == JNT_HINGE else 1.0),
      springref=float(a.get("springref", 0)) * (self.angle_scale if jtype == JNT_HINGE else 1.0),
      body=body,
    )
    if "solreflimit" in a:
      j.solref = _floats(a["solreflimit"])
    if "solimplimit" in a:
      j.solimp = _floats(a["solimplimit"], 5, [0.9, 0.95, 0.001, 0.5, 2.0])
    if "solreffriction" in a:
      j.solref_friction = _floats(a["solreffriction"])
    if "solimpfriction" in a:
      j.solimp_friction = _floats(a["solimpfriction"], 5, [0.9, 0.95, 0.001, 0.5, 2.0])
    if "actuatorfrcrange" in a:
      raise NotImplementedError("actuatorfrcrange is not supported")
    return j

  def _parse_geom(self, elem: ET.Element, childclass: str | None, body: SpecBody) -> SpecGeom:
    a = self._resolved("geom", elem.get("class", childclass), elem.attrib)
    gtype = _GEOM_TYPES[a.get("type", "sphere")]
    size = np.zeros(3)
    sz = _floats(a.get("size"))
    if sz is not None:
      size[: len(sz)] = sz
    pos = _floats(a.get("pos"), default=[0, 0, 0])
    quat = self._orientation(a)
    if "fromto" in a:
      if gtype not in (GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_ELLIPSOID):
        raise ValueError("fromto needs capsule/cylinder/box/ellipsoid")
      ft = _floats(a["fromto"])
      vec = ft[0:3] - ft[3:6]  # MuJoCo's convention: the frame z-axis points to -> from
      pos = 0.5 * (ft[0:3] + ft[3:6])
      quat = z_to_quat(vec)
      half = 0.5 * np.linalg.norm(vec)
      if gtype in (GEOM_CAPSULE, GEOM_CYLINDER):
        size[1] = half
      else:
        size[2] = half
        size[1] = size[0]
    g = SpecGeom(
      a.get("name", ""),
      gtype,
      size,
      pos,
      quat,
      contype=int(a.get("contype", 1)),
      conaffinity=int(a.get("conaffinity", 1)),
      condim=int(a.get("condim", 3)),
      priority=int(a.get("priority", 0)),
      group=int(a.get("group", 0)),
      solmix=float(a.get("solmix", 1.0)),
      margin=float(a.get("margin", 0.0)),
      gap=float(a.get("gap", 0.0)),
      body=body,
    )
    if "friction" in a:
      f = _floats(a["friction"])
      g.friction[: len(f)] = f
    if "solref" in a:
      g.solref = _floats(a["solref"])
    if "solimp" in a:
      g.solimp = _floats(a["solimp"], 5, [0.9, 0.95, 0.001, 0.5, 2.0])
    if "rgba" in a:
      g.rgba = _floats(a["rgba"])
    if "mass" in a:
      g.mass = float(a["mass"])
    if "density" in a:
      g.density = float(a["density"])
    if gtype == GEOM_MESH and (g.contype or g.conaffinity):
      raise NotImplementedError("colliding mesh geoms are not supported")
    return g


# ----------------------------------------------------------------------------
# Compiled model
# ----------------------------------------------------------------------------


class _Named:
  """Tiny accessor mirroring ``model.joint(name).qposadr`` style lookups."""

  def __init__(self, **kw: Any) -> None:
    self.__dict__.update(kw)


class Model:
  """Host ``mjModel`` (float64 / int32 numpy arrays named like MuJoCo's fields)."""

  _INT_FIELDS: tuple[str, ...] = ()

  def __init__(self) -> None:
    self.names: dict[str, list[str]] = {}
    self.opt = Option()
    self.meaninertia = 1.0

  @property
  def stat(self):
    """``mjModel.stat`` (the subset that exists here): ``meaninertia`` scales the solver's tolerances, ``extent`` the viewers."""
    import types

    return types.SimpleNamespace(meaninertia=self.meaninertia, extent=float(getattr(self, "stat_extent", None) or 1.0))

  # name lookups (reference use: src/mjlab/entity/entity.py:611-634)
  def _id(self, kind: str, name: str) -> int:
    try:
      return self.names[kind].index(name)
    except ValueError as e:
      raise KeyError(f"{kind} '{name}' not found") from e

  def body(self, name: str) -> _Named:
    i = self._id("body", name)
    return _Named(id=i, name=name)

  def joint(self, name: str) -> _Named:
    i = self._id("joint", name)
    return _Named(
      id=i,
      name=name,
      type=self.jnt_type[i : i + 1],
      dofadr=self.jnt_dofadr[i : i + 1],
      qposadr=self.jnt_qposadr[i : i + 1],
    )

  def geom(self, key: str | int) -> _Named:
    i = key if isinstance(key, (int, np.integer)) else self._id("geom", key)
    return _Named(
      id=i,
      name=self.names["geom"][i],
      condim=self.geom_condim[i : i + 1],
      priority=self.geom_priority[i : i + 1],
      friction=self.geom_friction[i],
    )

  def actuator(self, key: str | int) -> _Named:
    i = key if isinstance(key, (int, np.integer)) else self._id("actuator", key)
    return _Named(
      id=i,
      name=self.names["actuator"][i],
      gainprm=self.actuator_gainprm[i],
      biasprm=self.actuator_biasprm[i],
      forcerange=self.actuator_forcerange[i],
    )

  def sensor(self, name: str) -> _Named:
    i = self._id("sensor", name)
    return _Named(id=i, name=name, dim=self.sensor_dim[i : i + 1], adr=self.sensor_adr[i : i + 1])

  def key(self, name: str) -> _Named:
    i = self._id("key", name)
    return _Named(id=i, name=name, qpos=self.key_qpos[i], qvel=self.key_qvel[i], ctrl=self.key_ctrl[i])

  # serialisation: a compiled model travels as one .npz (the analogue of an .mjb)
  def save(self, path: str | Path) -> None:
    blob: dict[str, Any] = {}
    for k, v in self.__dict__.items():
      if isinstance(v, np.ndarray):
        blob["a:" + k] = v
      elif isinstance(v, (int, float)):
        blob["s:" + k] = np.array(v)
    for k, v in self.opt.__dict__.items():
      blob["o:" + k] = np.array(v)
    for k, v in self.names.items():
      blob["n:" + k] = np.array(v, dtype=np.str_) if v else np.array([], dtype=np.str_)
    np.savez_compressed(path, **blob)

  @classmethod
  def load(cls, path: str | Path) -> "Model":
    m = cls()
    with np.load(path, allow_pickle=False) as z:
      for k in z.files:
        kind, name = k.split(":", 1)
        v = z[k]
        if kind == "a":
          setattr(m, name, v)
        elif kind == "s":
          setattr(m, name, v.item())
        elif kind == "o":
          val = v.tolist()
          setattr(m.opt, name, tuple(val) if isinstance(val, list) else val)
        elif kind == "n":
          m.names[name] = [str(s) for s in v.tolist()]
    if not hasattr(m, "qpos_spring"):  # saved before the field existed: no springref in those models
      m.qpos_spring = np.asarray(m.qpos0, dtype=np.float64).copy()
    if not hasattr(m, "dof_solref"):  # saved before friction-loss rows existed: MuJoCo's defaults
      m.dof_solref = np.tile([0.02, 1.0], (m.nv, 1))
      m.dof_solimp = np.tile([0.9, 0.95, 0.001, 0.5, 2.0], (m.nv, 1))
    if not hasattr(m, "nstaticsite"):  # saved before static sites were told apart
      sstatic = m.body_weldid[m.site_bodyid] == 0
      m.nstaticsite = (int(np.argmin(sstatic)) if not sstatic.all() else int(m.nsite)) if m.nsite else 0
    if not hasattr(m, "tgrid_ztop"):  # saved before the static-geometry / terrain fields existed
      static = m.body_weldid[m.geom_bodyid] == 0
      m.nstaticgeom = int(np.argmin(static)) if not static.all() else m.ngeom
      m.geom_lds0 = int(min(m.nstaticgeom, m.pair_geom.min())) if m.npair else m.nstaticgeom
      _compile_terrain(m, np.zeros(0, np.int64), [])
    return m


# geom-type pairs (type1 <= type2) the collision stage has a function for; moving spheres, capsules
# and (by their corners) boxes additionally collide with static boxes through the terrain path
_PAIR_FUNCS = {
  (GEOM_PLANE, GEOM_SPHERE), (GEOM_PLANE, GEOM_CAPSULE), (GEOM_PLANE, GEOM_BOX),
  (GEOM_SPHERE, GEOM_SPHERE), (GEOM_SPHERE, GEOM_CAPSULE), (GEOM_CAPSULE, GEOM_CAPSULE),
}  # fmt: skip
TERRAIN_CELL = 0.5  # m, edge of a broadphase grid cell
TCAND_MAX = 12  # terrain boxes kept per moving geom and step (the ones with the smallest ids)


def _geom_mass_inertia(g: SpecGeom) -> tuple[float, np.ndarray]:
  """Mass and principal moments (geom frame, about the geom centre) of a solid primitive."""
  t, s = g.type, g.size
  pi = math.pi
  if t == GEOM_SPHERE:
    vol, unit = 4.0 / 3.0 * pi * s[0] ** 3, np.full(3, 0.4 * s[0] ** 2)
  elif t == GEOM_BOX:
    vol = 8.0 * s[0] * s[1] * s[2]
    unit = np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2]) / 3.0
  elif t == GEOM_ELLIPSOID:
    vol = 4.0 / 3.0 * pi * s[0] * s[1] * s[2]
    unit = np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2]) / 5.0
  elif t == GEOM_CYLINDER:
    r, h = s[0], s[1]
    vol = 2.0 * pi * r * r * h
    unit = np.array([r * r / 4.0 + h * h / 3.0, r * r / 4.0 + h * h / 3.0, r * r / 2.0])
  elif t == GEOM_CAPSULE:
    r, h = s[0], s[1]
    vc, vs = 2.0 * pi * r * r * h, 4.0 / 3.0 * pi * r**3  # cylinder + the two half spheres
    vol = vc + vs
    ixx = vc * (r * r / 4.0 + h * h / 3.0) + vs * (0.4 * r * r + h * h + 0.75 * h * r)
    izz = vc * r * r / 2.0 + vs * 0.4 * r * r
    unit = np.array([ixx, ixx, izz]) / vol
  else:
    return 0.0, np.zeros(3)  # planes, meshes, heightfields carry no inferred mass here
  mass = g.mass if g.mass is not None else g.density * vol
  return float(mass), unit * mass


def inertia_from_geoms(geoms: list[SpecGeom]) -> tuple[float, np.ndarray, np.ndarray, np.ndarray]:
  """Body mass, centre of mass, principal frame (quaternion) and principal moments from its geoms:
  what the MJCF compiler derives when a body has no ``<inertial>`` (inertiafromgeom).  Principal
  moments are listed in decreasing order, the frame is right handed."""
  mass, first = 0.0, np.zeros(3)
  parts = []
  for g in geoms:
    mg, ig = _geom_mass_inertia(g)
    if mg <= 0.0:
      continue
    r = quat_to_mat(quat_normalize(g.quat))
    parts.append((mg, np.asarray(g.pos, dtype=np.float64), r @ np.diag(ig) @ r.T))
    mass += mg
    first += mg * np.asarray(g.pos, dtype=np.float64)
  if mass <= 0.0:
    return 0.0, np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
  com = first / mass
  inertia = np.zeros((3, 3))
  for mg, p, ig in parts:
    d = p - com
    inertia += ig + mg * (d @ d * np.eye(3) - np.outer(d, d))  # parallel axes
  w, v = np.linalg.eigh(inertia)
  w, v = w[::-1], v[:, ::-1]
  if np.linalg.det(v) < 0:
    v[:, 2] = -v[:, 2]
  return mass, com, mat_to_quat(v), w


def mat_to_quat(r: np.ndarray) -> np.ndarray:
  """Rotation matrix -> unit quaternion (w, x, y, z), w >= 0."""
  t = np.trace(r)
  if t > 0:
    s4 = math.sqrt(t + 1.0) * 2
    q = np.array([0.25 * s4, (r[2, 1] - r[1, 2]) / s4, (r[0, 2] - r[2, 0]) / s4, (r[1, 0] - r[0, 1]) / s4])
  else:
    i = int(np.argmax(np.diag(r)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s4 = math.sqrt(1.0 + r[i, i] - r[j, j] - r[k, k]) * 2
    q = np.zeros(4)
    q[0] = (r[k, j] - r[j, k]) / s4
    q[1 + i] = 0.25 * s4
    q[1 + j] = (r[j, i] + r[i, j]) / s4
    q[1 + k] = (r[k, i] + r[i, k]) / s4
  q = q / np.linalg.norm(q)
  return q if q[0] >= 0 else -q


def _static_body_poses(m: "Model") -> tuple[np.ndarray, np.ndarray]:
  """World pose of every static body (weldid 0); rows of moving bodies are left at identity."""
  xpos = np.zeros((m.nbody, 3))
  xquat = np.tile([1.0, 0, 0, 0], (m.nbody, 1))
  for b in range(1, m.nbody):
    if m.body_weldid[b] == 0:
      p = m.body_parentid[b]
      xpos[b] = xpos[p] + quat_to_mat(xquat[p]) @ m.body_pos[b]
      xquat[b] = quat_normalize(quat_mul(xquat[p], m.body_quat[b]))
  return xpos, xquat


def _compile_terrain(m: "Model", tids: np.ndarray, moving: list[int]) -> None:
  """Terrain boxes in the world frame + the uniform xy grid the collision stage walks.

  ``tgrid_item[tgrid_start[c] : tgrid_start[c + 1]]`` lists (ascending) the boxes whose
  footprint touches cell ``c = ix * ny + iy``; ``tbox_cell0`` is the lowest cell of a box, which
  lets a walker visit a (geom, box) pair exactly once: in the cell
  ``(max(ix0_geom, ix0_box), max(iy0_geom, iy0_box))``."""
  nt = len(tids)
  m.nterrain = nt
  m.tbox_geom = tids.astype(np.int32)
  m.tbox_pos = np.zeros((nt, 3))
  m.tbox_mat = np.zeros((nt, 9))
  m.tbox_size = m.geom_size[tids].copy().reshape(nt, 3)
  m.tbox_cell0 = np.zeros((nt, 2), np.int32)
  m.tgeom = np.zeros(0, np.int32)
  m.tgrid_start = np.zeros(1, np.int32)
  m.tgrid_item = np.zeros(0, np.int32)
  m.tgrid_nx = m.tgrid_ny = 0
  m.tgrid_x0 = m.tgrid_y0 = 0.0
  m.tgrid_cell = TERRAIN_CELL
  m.tgrid_ztop = np.zeros(0)
  if nt == 0:
    m.ntgeom, m.ntcell, m.ntcellp1, m.ntitem = 0, 0, 1, 0
    return
  if np.any(m.geom_margin[tids] != 0) or np.any(m.geom_gap[tids] != 0):
    raise NotImplementedError("terrain boxes must have margin = gap = 0")
  ct, ca = m.geom_contype[tids], m.geom_conaffinity[tids]
  if (ct != ct[0]).any() or (ca != ca[0]).any():
    raise NotImplementedError("terrain boxes must share one contype / conaffinity")
  tg = [g for g in moving if (m.geom_contype[g] & ca[0]) or (ct[0] & m.geom_conaffinity[g])]
  for g in tg:
    if m.geom_type[g] not in (GEOM_SPHERE, GEOM_CAPSULE, GEOM_BOX):
      raise NotImplementedError(f"geom '{m.names['geom'][g]}' (type {m.geom_type[g]}) vs box terrain: only spheres, capsules and boxes (corner contacts) collide with static boxes")
  m.tgeom = np.array(tg, np.int32)
  m.ntgeom = len(tg)
  xpos, xquat = _static_body_poses(m)
  half = np.zeros((nt, 3))
  for i, g in enumerate(tids):
    b = m.geom_bodyid[g]
    rb = quat_to_mat(xquat[b])
    r = rb @ quat_to_mat(m.geom_quat[g])
    m.tbox_pos[i] = xpos[b] + rb @ m.geom_pos[g]
    m.tbox_mat[i] = r.reshape(9)
    half[i] = np.abs(r) @ m.geom_size[g]
  lo, hi = m.tbox_pos - half, m.tbox_pos + half
  cell = TERRAIN_CELL
  x0, y0 = float(lo[:, 0].min()), float(lo[:, 1].min())
  nx = max(1, int(np.ceil((hi[:, 0].max() - x0) / cell)))
  ny = max(1, int(np.ceil((hi[:, 1].max() - y0) / cell)))
  ix0 = np.clip(np.floor((lo[:, 0] - x0) / cell).astype(np.int64), 0, nx - 1)
  ix1 = np.clip(np.floor((hi[:, 0] - x0) / cell).astype(np.int64), 0, nx - 1)
  iy0 = np.clip(np.floor((lo[:, 1] - y0) / cell).astype(np.int64), 0, ny - 1)
  iy1 = np.clip(np.floor((hi[:, 1] - y0) / cell).astype(np.int64), 0, ny - 1)
  cells, items = [], []
  for i in range(nt):
    c = np.add.outer(np.arange(ix0[i], ix1[i] + 1) * ny, np.arange(iy0[i], iy1[i] + 1)).ravel()
    cells.append(c)
    items.append(np.full(c.size, i, np.int64))
  cells, items = np.concatenate(cells), np.concatenate(items)
  order = np.argsort(cells, kind="stable")  # boxes stay in ascending order inside a cell
  m.tgrid_item = items[order].astype(np.int32)
  m.tgrid_start = np.searchsorted(cells[order], np.arange(nx * ny + 1)).astype(np.int32)
  m.tbox_cell0 = np.stack([ix0, iy0], axis=1).astype(np.int32)
  m.tgrid_nx, m.tgrid_ny, m.tgrid_x0, m.tgrid_y0 = nx, ny, x0, y0
  m.ntcell, m.ntcellp1, m.ntitem = nx * ny, nx * ny + 1, int(m.tgrid_item.size)
  # highest box top per cell (-inf-like for empty cells): a geom whose bounding sphere is wholly
  # above it cannot reach any box of the cell
  m.tgrid_ztop = np.full(nx * ny, -1.0e30)
  np.maximum.at(m.tgrid_ztop, cells, hi[items, 2])


def finalize_topology(m: "Model", excl: set) -> None:
  """Everything the kernels need beyond mjModel's own arrays, derived from them: tree levels,
  subtree sizes (bodies are numbered depth first, so a subtree is an id range), the ancestor-dof
  bit mask of every body, the leading run of static geoms, the static candidate pair list and the
  terrain grid.  ``excl`` holds the excluded body pairs ``(min, max)``.  Shared by the MJCF compiler
  below and by ``from_mujoco.model_from_mujoco`` (a model that arrives as ``mujoco.MjModel``)."""
  nbody, nv, ngeom = m.nbody, m.nv, m.ngeom
  # body level in the tree (world = 0); used by the level-parallel sweeps
  m.body_depth = np.zeros(nbody, np.int32)
  for i in range(1, nbody):
    m.body_depth[i] = m.body_depth[m.body_parentid[i]] + 1

  m.nlevel = int(m.body_depth.max()) + 1
  order = np.argsort(m.body_depth, kind="stable").astype(np.int32)
  m.level_body = order
  m.level_adr = np.searchsorted(m.body_depth[order], np.arange(m.nlevel + 1)).astype(np.int32)
  # bodies are in depth-first order, so a subtree is the contiguous id range [b, b + subtreenum[b])
  m.body_subtreenum = np.ones(nbody, np.int32)
  for i in range(nbody - 1, 0, -1):
    m.body_subtreenum[m.body_parentid[i]] += m.body_subtreenum[i]

  if nv > 64:
    raise NotImplementedError("nv > 64 is not supported by the wave-per-world kernels")
  # per-body bitmask of the dofs that move it (ancestor chain); nv <= 64
  m.body_dofmask = np.zeros(nbody, np.uint64)
  for i in range(1, nbody):
    mask = int(m.body_dofmask[m.body_parentid[i]])
    for k in range(m.body_dofnum[i]):
      mask |= 1 << int(m.body_dofadr[i] + k)
    m.body_dofmask[i] = np.uint64(mask)

  # Static geoms (world body or welded to it) never move: their poses are computed once.  Static
  # colliding BOXES are the terrain (reference src/mjlab/terrains/primitive_terrains.py: every
  # terrain piece is a box under the static ``terrain`` body); there can be thousands, so they
  # never enter the static pair list -- moving geoms find them through a uniform xy grid.
  static = m.body_weldid[m.geom_bodyid] == 0
  collides = (m.geom_contype != 0) | (m.geom_conaffinity != 0)
  terrain = static & (m.geom_type == GEOM_BOX) & collides
  m.nstaticgeom = int(np.argmin(static)) if not static.all() else ngeom  # leading run of static geoms
  sstatic = m.body_weldid[m.site_bodyid] == 0
  m.nstaticsite = (int(np.argmin(sstatic)) if not sstatic.all() else int(m.nsite)) if m.nsite else 0  # leading run of static sites
  pairs = []
  cand = [g for g in range(ngeom) if collides[g] and not terrain[g]]
  for i1, g1 in enumerate(cand):
    for g2 in cand[i1 + 1 :]:
      ct1, ca1 = m.geom_contype[g1], m.geom_conaffinity[g1]
      ct2, ca2 = m.geom_contype[g2], m.geom_conaffinity[g2]
      if not ((ct1 & ca2) or (ct2 & ca1)):
        continue
      b1, b2 = m.geom_bodyid[g1], m.geom_bodyid[g2]
      w1, w2 = m.body_weldid[b1], m.body_weldid[b2]
      if w1 == w2:
        continue  # same (welded) body, includes static-static
      if w1 != 0 and w2 != 0:
        pw1 = m.body_weldid[m.body_parentid[w1]]
        pw2 = m.body_weldid[m.body_parentid[w2]]
        if pw1 == w2 or pw2 == w1:
          continue  # parent-child filter (only when neither is welded to the world)
      if (min(b1, b2), max(b1, b2)) in excl:
        continue
      # collision functions are defined for type1 <= type2
      a_, b_ = (g1, g2) if m.geom_type[g1] <= m.geom_type[g2] else (g2, g1)
      t1, t2 = m.geom_type[a_], m.geom_type[b_]
      if (t1, t2) not in _PAIR_FUNCS:
        raise NotImplementedError(f"no collision function for geom types ({t1}, {t2}): '{m.names['geom'][a_]}' vs '{m.names['geom'][b_]}'")
      pairs.append((a_, b_))
  m.pair_geom = np.array(pairs, np.int32).reshape(len(pairs), 2)
  m.npair = len(pairs)
  _compile_terrain(m, np.flatnonzero(terrain), [g for g in cand if not static[g]])
  # geoms [geom_lds0, ngeom) are the ones the collision stage keeps on chip
  m.geom_lds0 = int(min(m.nstaticgeom, m.pair_geom.min())) if m.npair else m.nstaticgeom



def _compile(spec: Spec) -> Model:
  m = Model()
  m.opt = Option(**spec.option.__dict__)
  bodies = spec.bodies  # depth-first, world first (MuJoCo body order)
  bid = {id(b): i for i, b in enumerate(bodies)}
  nbody = len(bodies)
  joints = [j for b in bodies for j in b.joints]
  geoms = [g for b in bodies for g in b.geoms]
  sites = [s for b in bodies for s in b.sites]
  njnt, ngeom, nsite = len(joints), len(geoms), len(sites)
  # element ids, valid after compile like mujoco's `Mjs*.id` (reference use: entity/entity.py:589-600)
  for seq in (bodies, joints, geoms, sites, spec.actuators, spec.sensors, spec.keys):
    for i, e in enumerate(seq):
      e.id = i

  m.names = {
    "body": [b.name for b in bodies],
    "joint": [j.name for j in joints],
    "geom": [g.name for g in geoms],
    "site": [s.name for s in sites],
    "actuator": [a.name for a in spec.actuators],
    "sensor": [s.name for s in spec.sensors],
    "key": [k.name for k in spec.keys],
  }
  for kind in ("body", "joint", "actuator", "sensor"):
    named = [n for n in m.names[kind] if n]
    if len(set(named)) != len(named):
      raise ValueError(f"repeated {kind} name")

  # ---- bodies -------------------------------------------------------------
  f64 = np.float64
  m.body_parentid = np.zeros(nbody, np.int32)
  m.body_rootid = np.zeros(nbody, np.int32)
  m.body_weldid = np.zeros(nbody, np.int32)
  m.body_jntnum = np.zeros(nbody, np.int32)
  m.body_jntadr = np.full(nbody, -1, np.int32)
  m.body_dofnum = np.zeros(nbody, np.int32)
  m.body_dofadr = np.full(nbody, -1, np.int32)
  m.body_geomnum = np.zeros(nbody, np.int32)
  m.body_geomadr = np.full(nbody, -1, np.int32)
  m.body_pos = np.zeros((nbody, 3), f64)
  m.body_quat = np.tile([1.0, 0, 0, 0], (nbody, 1))
  m.body_ipos = np.zeros((nbody, 3), f64)
  m.body_iquat = np.tile([1.0, 0, 0, 0], (nbody, 1))
  m.body_mass = np.zeros(nbody, f64)
  m.body_inertia = np.zeros((nbody, 3), f64)

  jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid = [], [], [], []
  nq = nv = 0
  jadr = gadr = 0
  for i, b in enumerate(bodies):
    p = 0 if b.parent is None else bid[id(b.parent)]
    m.body_parentid[i] = p
    m.body_pos[i] = b.pos
    m.body_quat[i] = quat_normalize(b.quat)
    if b.inertia is not None:
      m.body_ipos[i] = b.ipos
      m.body_iquat[i] = quat_normalize(b.iquat)
      m.body_mass[i] = b.mass
      m.body_inertia[i] = b.inertia
    elif b.geoms and (b.joints or any(g.mass is not None for g in b.geoms)):
      # no <inertial>: mass and inertia from the body's geoms (MJCF compiler inertiafromgeom="auto")
      mass, ipos, iquat, inertia = inertia_from_geoms(b.geoms)
      if b.joints and mass <= 0.0:
        raise ValueError(f"moving body '{b.name}' has no mass: give it an <inertial> or geoms with mass / density")
      m.body_ipos[i], m.body_iquat[i], m.body_mass[i], m.body_inertia[i] = ipos, iquat, mass, inertia
    elif b.joints:
      raise ValueError(f"moving body '{b.name}' needs an <inertial> or geoms to take its inertia from")
    if i > 0:
      m.body_rootid[i] = i if p == 0 else m.body_rootid[p]
      m.body_weldid[i] = i if b.joints else m.body_weldid[p]
    if b.joints:
      if any(j.type == JNT_FREE for j in b.joints) and (len(b.joints) > 1 or p != 0):
        raise ValueError("free joint must be alone on a child of the world")
      m.body_jntnum[i] = len(b.joints)
      m.body_jntadr[i] = jadr
      m.body_dofadr[i] = nv
      for j in b.joints:
        jnt_type.append(j.type)
        jnt_qposadr.append(nq)
        jnt_dofadr.append(nv)
        jnt_bodyid.append(i)
        nq += 7 if j.type == JNT_FREE else 1
        nv += 6 if j.type == JNT_FREE else 1
      m.body_dofnum[i] = nv - m.body_dofadr[i]
      jadr += len(b.joints)
    if b.geoms:
      m.body_geomnum[i] = len(b.geoms)
      m.body_geomadr[i] = gadr
      gadr += len(b.geoms)

  m.nq, m.nv, m.nbody, m.njnt, m.ngeom, m.nsite = nq, nv, nbody, njnt, ngeom, nsite
  m.na = 0

  # subtree mass
  m.body_subtreemass = m.body_mass.copy()
  for i in range(nbody - 1, 0, -1):
    m.body_subtreemass[m.body_parentid[i]] += m.body_subtreemass[i]

  # ---- joints / dofs -------------------------------------------------------
  m.jnt_type = np.array(jnt_type, np.int32).reshape(njnt)
  m.jnt_qposadr = np.array(jnt_qposadr, np.int32).reshape(njnt)
  m.jnt_dofadr = np.array(jnt_dofadr, np.int32).reshape(njnt)
  m.jnt_bodyid = np.array(jnt_bodyid, np.int32).reshape(njnt)
  m.jnt_limited = np.array([j.limited for j in joints], np.int32).reshape(njnt)
  m.jnt_pos = np.array([j.pos for j in joints], f64).reshape(njnt, 3)
  m.jnt_axis = np.array([j.axis for j in joints], f64).reshape(njnt, 3)
  m.jnt_range = np.array([j.range for j in joints], f64).reshape(njnt, 2)
  m.jnt_margin = np.array([j.margin for j in joints], f64).reshape(njnt)
  m.jnt_stiffness = np.array([j.stiffness for j in joints], f64).reshape(njnt)
  m.jnt_solref = np.array([j.solref for j in joints], f64).reshape(njnt, 2)
  m.jnt_solimp = np.array([j.solimp for j in joints], f64).reshape(njnt, 5)

  m.qpos0 = np.zeros(nq, f64)
  m.qpos_spring = np.zeros(nq, f64)  # mjModel.qpos_spring: reference pose of the joint springs (springref; free joints: qpos0)
  m.dof_bodyid = np.zeros(nv, np.int32)
  m.dof_jntid = np.zeros(nv, np.int32)
  m.dof_parentid = np.full(nv, -1, np.int32)
  m.dof_armature = np.zeros(nv, f64)
  m.dof_damping = np.zeros(nv, f64)
  m.dof_frictionloss = np.zeros(nv, f64)
  m.dof_solref = np.tile([0.02, 1.0], (nv, 1)).reshape(nv, 2)
  m.dof_solimp = np.tile([0.9, 0.95, 0.001, 0.5, 2.0], (nv, 1)).reshape(nv, 5)
  for ji, j in enumerate(joints):
    qa, da = m.jnt_qposadr[ji], m.jnt_dofadr[ji]
    if j.type == JNT_FREE:
      b = j.body
      m.qpos0[qa : qa + 3] = b.pos
      m.qpos0[qa + 3 : qa + 7] = quat_normalize(b.quat)
      m.qpos_spring[qa : qa + 7] = m.qpos0[qa : qa + 7]
      nd = 6
    else:
      m.qpos0[qa] = j.ref
      m.qpos_spring[qa] = j.springref
      nd = 1
    for k in range(nd):
      m.dof_bodyid[da + k] = m.jnt_bodyid[ji]
      m.dof_jntid[da + k] = ji
      m.dof_armature[da + k] = j.armature
      m.dof_damping[da + k] = j.damping
      m.dof_frictionloss[da + k] = j.frictionloss
      m.dof_solref[da + k] = j.solref_friction
      m.dof_solimp[da + k] = j.solimp_friction
  # dof_parentid: previous dof in the same body, else last dof of nearest moving ancestor
  last_dof_of_body = np.full(nbody, -1, np.int32)
  for i in range(1, nbody):
    p = m.body_parentid[i]
    inherited = last_dof_of_body[p]
    if m.body_dofnum[i] > 0:
      a = m.body_dofadr[i]
      for k in range(m.body_dofnum[i]):
        m.dof_parentid[a + k] = inherited if k == 0 else a + k - 1
      last_dof_of_body[i] = a + m.body_dofnum[i] - 1
    else:
      last_dof_of_body[i] = inherited
  # ---- geoms / sites -------------------------------------------------------
  m.geom_type = np.array([g.type for g in geoms], np.int32).reshape(ngeom)
  m.geom_bodyid = np.array([bid[id(g.body)] for g in geoms], np.int32).reshape(ngeom)
  m.geom_contype = np.array([g.contype for g in geoms], np.int32).reshape(ngeom)
  m.geom_conaffinity = np.array([g.conaffinity for g in geoms], np.int32).reshape(ngeom)
  m.geom_condim = np.array([g.condim for g in geoms], np.int32).reshape(ngeom)
  m.geom_priority = np.array([g.priority for g in geoms], np.int32).reshape(ngeom)
  m.geom_size = np.array([g.size for g in geoms], f64).reshape(ngeom, 3)
  m.geom_pos = np.array([g.pos for g in geoms], f64).reshape(ngeom, 3)
  m.geom_quat = np.array([quat_normalize(g.quat) for g in geoms], f64).reshape(ngeom, 4)
  m.geom_friction = np.array([g.friction for g in geoms], f64).reshape(ngeom, 3)
  m.geom_solref = np.array([g.solref for g in geoms], f64).reshape(ngeom, 2)
  m.geom_solimp = np.array([g.solimp for g in geoms], f64).reshape(ngeom, 5)
  m.geom_solmix = np.array([g.solmix for g in geoms], f64).reshape(ngeom)
  m.geom_margin = np.array([g.margin for g in geoms], f64).reshape(ngeom)
  m.geom_gap = np.array([g.gap for g in geoms], f64).reshape(ngeom)
  m.geom_rgba = np.array([g.rgba for g in geoms], f64).reshape(ngeom, 4)
  m.geom_rbound = np.zeros(ngeom, f64)
  for gi, g in enumerate(geoms):
    t, s = g.type, g.size
    if t == GEOM_SPHERE:
      m.geom_rbound[gi] = s[0]
    elif t == GEOM_CAPSULE:
      m.geom_rbound[gi] = s[0] + s[1]
    elif t == GEOM_CYLINDER:
      m.geom_rbound[gi] = math.hypot(s[0], s[1])
    elif t in (GEOM_BOX, GEOM_ELLIPSOID):
      m.geom_rbound[gi] = np.linalg.norm(s) if t == GEOM_BOX else max(s)
    elif t == GEOM_PLANE:
      m.geom_rbound[gi] = 0.0
    # cylinders / ellipsoids may exist and even be collidable: finalize_topology rejects the model
    # only if one of them ends up in a candidate pair (no collision function) -- a lone static
    # cylinder, as in the reference's own fixtures (tests/test_entity.py:84), is harmless
    if t == GEOM_HFIELD and (g.contype or g.conaffinity):
      raise NotImplementedError("colliding height fields are not supported")
    if g.condim not in (1, 3):
      raise NotImplementedError("only condim 1 and 3 are supported")
  m.site_bodyid = np.array([bid[id(s.body)] for s in sites], np.int32).reshape(nsite)
  m.site_pos = np.array([s.pos for s in sites], f64).reshape(nsite, 3)
  m.site_quat = np.array([quat_normalize(s.quat) for s in sites], f64).reshape(nsite, 4)

  # ---- actuators (joint transmission, fixed gain, affine bias) --------------
  nu = len(spec.actuators)
  m.nu = nu
  m.actuator_trnid = np.zeros((nu, 2), np.int32)
  m.actuator_gainprm = np.zeros((nu, 10), f64)
  m.actuator_biasprm = np.zeros((nu, 10), f64)
  m.actuator_ctrllimited = np.zeros(nu, np.int32)
  m.actuator_forcelimited = np.zeros(nu, np.int32)
  m.actuator_ctrlrange = np.zeros((nu, 2), f64)
  m.actuator_forcerange = np.zeros((nu, 2), f64)
  m.actuator_gear = np.zeros((nu, 6), f64)
  for ai, a in enumerate(spec.actuators):
    ji = m.names["joint"].index(a.joint)
    if m.jnt_type[ji] == JNT_FREE:
      raise ValueError("actuator on a free joint")
    m.actuator_trnid[ai] = (ji, -1)
    m.actuator_gainprm[ai, 0] = a.gainprm0
    m.actuator_biasprm[ai, :3] = a.biasprm
    m.actuator_gear[ai, 0] = a.gear
    if a.ctrlrange is not None:
      m.actuator_ctrllimited[ai] = 1
      m.actuator_ctrlrange[ai] = a.ctrlrange
    if a.forcerange is not None:
      m.actuator_forcelimited[ai] = 1
      m.actuator_forcerange[ai] = a.forcerange

  # ---- sensors (contact sensors only) ----------------------------------------
  ns = len(spec.sensors)
  m.nsensor = ns
  m.sensor_type = np.full(ns, SENS_CONTACT, np.int32)
  m.sensor_objtype = np.zeros(ns, np.int32)
  m.sensor_objid = np.zeros(ns, np.int32)
  m.sensor_reftype = np.full(ns, -1, np.int32)
  m.sensor_refid = np.full(ns, -1, np.int32)
  m.sensor_intprm = np.zeros((ns, 3), np.int32)
  m.sensor_dim = np.zeros(ns, np.int32)
  m.sensor_adr = np.zeros(ns, np.int32)
  adr = 0

  def _objid(otype: int, name: str) -> int:
    kind = {OBJ_BODY: "body", OBJ_XBODY: "body", OBJ_GEOM: "geom", OBJ_SITE: "site"}[otype]
    return m.names[kind].index(name)

  for si, s in enumerate(spec.sensors):
    dataspec, reduce_, num = s.intprm
    if s.objtype == OBJ_SITE:
      raise NotImplementedError("site-volume contact sensors are not supported")
    m.sensor_objtype[si] = s.objtype
    m.sensor_objid[si] = _objid(s.objtype, s.objname)
    if s.reftype is not None:
      m.sensor_reftype[si] = s.reftype
      m.sensor_refid[si] = _objid(s.reftype, s.refname)
    m.sensor_intprm[si] = s.intprm
    # slot size: found 1, force 3, torque 3, dist 1, pos 3, normal 3, tangent 3
    sizes = [1, 3, 3, 1, 3, 3, 3]
    slot = sum(sz for bit, sz in enumerate(sizes) if dataspec & (1 << bit))
    m.sensor_dim[si] = slot * num
    m.sensor_adr[si] = adr
    adr += slot * num
  m.nsensordata = adr

  # ---- excludes + static candidate pair list ---------------------------------
  excl = set()
  for a, b in spec.excludes:
    ia, ib = m.names["body"].index(a), m.names["body"].index(b)
    excl.add((min(ia, ib), max(ia, ib)))
  m.nexclude = len(excl)
  m.exclude_signature = np.array(sorted((a << 16) + b for a, b in excl), np.int32)  # mjModel's encoding
  finalize_topology(m, excl)

  # ---- keyframes ----------------------------------------------------------------
  nkey = len(spec.keys)
  m.nkey = nkey
  m.key_qpos = np.zeros((nkey, nq), f64)
  m.key_qvel = np.zeros((nkey, nv), f64)
  m.key_ctrl = np.zeros((nkey, nu), f64)
  for ki, k in enumerate(spec.keys):
    if len(k.qpos) != nq:
      raise ValueError(f"key '{k.name}' qpos has size {len(k.qpos)}, expected {nq}")
    m.key_qpos[ki] = k.qpos
    if k.qvel is not None:
      m.key_qvel[ki] = k.qvel
    if k.ctrl is not None:
      m.key_ctrl[ki] = k.ctrl

  _set_const(m)
  return m


# ----------------------------------------------------------------------------
# mj_setConst restatement: invweight0 / meaninertia at qpos0 (float64, numpy)
# ----------------------------------------------------------------------------


def _rot(q, v):
  return quat_to_mat(q) @ v


def kinematics_np(m: Model, qpos: np.ndarray) -> dict[str, np.ndarray]:
  """Forward kinematics + com-frame quantities in numpy (compile-time use only)."""
  nb = m.nbody
  xpos = np.zeros((nb, 3))
  xquat = np.tile([1.0, 0, 0, 0], (nb, 1))
  xanchor = np.zeros((m.njnt, 3))
  xaxis = np.zeros((m.njnt, 3))
  for i in range(1, nb):
    p = m.body_parentid[i]
    ja, jn = m.body_jntadr[i], m.body_jntnum[i]
    if jn == 1 and m.jnt_type[ja] == JNT_FREE:
      qa = m.jnt_qposadr[ja]
      xpos[i] = qpos[qa : qa + 3]
      xquat[i] = quat_normalize(qpos[qa + 3 : qa + 7])
      xanchor[ja] = xpos[i]
      xaxis[ja] = m.jnt_axis[ja]
    else:
      pos = xpos[p] + _rot(xquat[p], m.body_pos[i])
      quat = quat_mul(xquat[p], m.body_quat[i])
      for j in range(ja, ja + jn):
        qa = m.jnt_qposadr[j]
        xaxis[j] = _rot(quat, m.jnt_axis[j])
        xanchor[j] = _rot(quat, m.jnt_pos[j]) + pos
        if m.jnt_type[j] == JNT_SLIDE:
          pos = pos + xaxis[j] * (qpos[qa] - m.qpos0[qa])
        else:
          ang = qpos[qa] - m.qpos0[qa]
          ql = np.concatenate([[math.cos(ang / 2)], m.jnt_axis[j] * math.sin(ang / 2)])
          quat = quat_mul(quat, ql)
          pos = xanchor[j] - _rot(quat, m.jnt_pos[j])
      xpos[i] = pos
      xquat[i] = quat_normalize(quat)
  xipos = np.array([xpos[i] + _rot(xquat[i], m.body_ipos[i]) for i in range(nb)])
  ximat = np.array([quat_to_mat(quat_mul(xquat[i], m.body_iquat[i])) for i in range(nb)])
  # subtree com
  com = m.body_mass[:, None] * xipos
  for i in range(nb - 1, 0, -1):
    com[m.body_parentid[i]] += com[i]
  sub = np.where(
    m.body_subtreemass[:, None] < MJ_MINVAL, xipos, com / np.maximum(m.body_subtreemass[:, None], MJ_MINVAL)
  )
  # cdof about subtree_com[root]
  cdof = np.zeros((m.nv, 6))
  for j in range(m.njnt):
    b = m.jnt_bodyid[j]
    da = m.jnt_dofadr[j]
    off = sub[m.body_rootid[b]] - xanchor[j]
    if m.jnt_type[j] == JNT_FREE:
      for k in range(3):
        cdof[da + k, 3 + k] = 1.0
      R = quat_to_mat(xquat[b])
      for k in range(3):
        ax = R[:, k]
        cdof[da + 3 + k, :3] = ax
        cdof[da + 3 + k, 3:] = np.cross(ax, off)
    elif m.jnt_type[j] == JNT_SLIDE:
      cdof[da, 3:] = xaxis[j]
    else:
      cdof[da, :3] = xaxis[j]
      cdof[da, 3:] = np.cross(xaxis[j], off)
  return dict(xpos=xpos, xquat=xquat, xipos=xipos, ximat=ximat, subtree_com=sub, cdof=cdof)


def mass_matrix_np(m: Model, kin: dict[str, np.ndarray]) -> np.ndarray:
  """Dense joint-space inertia via body Jacobians (independent of the CRB code paths)."""
  nv = m.nv
  M = np.zeros((nv, nv))
  for b in range(1, m.nbody):
    if m.body_mass[b] == 0 and not np.any(m.body_inertia[b]):
      continue
    jp, jr = jac_np(m, kin, b, kin["xipos"][b])
    R = kin["ximat"][b]
    Iw = R @ np.diag(m.body_inertia[b]) @ R.T
    M += m.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
  M[np.diag_indices(nv)] += m.dof_armature
  return M


def jac_np(m: Model, kin: dict[str, np.ndarray], body: int, point: np.ndarray):
  nv = m.nv
  jp, jr = np.zeros((3, nv)), np.zeros((3, nv))
  off = point - kin["subtree_com"][m.body_rootid[body]]
  mask = int(m.body_dofmask[body])
  for d in range(nv):
    if mask >> d & 1:
      w, v = kin["cdof"][d, :3], kin["cdof"][d, 3:]
      jr[:, d] = w
      jp[:, d] = v + np.cross(w, off)
  return jp, jr


def _set_const(m: Model) -> None:
  nv = m.nv
  m.dof_invweight0 = np.zeros(nv)
  m.dof_M0 = np.zeros(nv)
  m.body_invweight0 = np.zeros((m.nbody, 2))
  m.meaninertia = 1.0
  if nv == 0:
    return
  kin = kinematics_np(m, m.qpos0)
  M = mass_matrix_np(m, kin)
  Minv = np.linalg.inv(M)
  m.dof_M0 = np.diag(M).copy()
  m.meaninertia = float(np.mean(np.diag(M)))
  for b in range(1, m.nbody):
    if m.body_weldid[b] == 0:
      continue
    jp, jr = jac_np(m, kin, b, kin["xipos"][b])
    Ap = jp @ Minv @ jp.T
    Ar = jr @ Minv @ jr.T
    m.body_invweight0[b, 0] = np.trace(Ap) / 3.0
    m.body_invweight0[b, 1] = np.trace(Ar) / 3.0
  for j in range(m.njnt):
    da = m.jnt_dofadr[j]
    if m.jnt_type[j] == JNT_FREE:
      m.dof_invweight0[da : da + 3] = np.mean(np.diag(Minv)[da : da + 3])
      m.dof_invweight0[da + 3 : da + 6] = np.mean(np.diag(Minv)[da + 3 : da + 6])
    else:
      m.dof_invweight0[da] = Minv[da, da]


def resolve_expr(pattern_map: dict[str, Any], names: list[str], default: Any = 0.0) -> list[Any]:
  """First-match regex resolution (same semantics as reference src/mjlab/utils/string.py:6-24)."""
  compiled = [(re.compile(p), v) for p, v in pattern_map.items()]
  out = []
  for n in names:
    for pat, v in compiled:
      if pat.match(n):
        out.append(v)
        break
    else:
      out.append(default)
  return out


def filter_exp(exprs: list[str], names: list[str]) -> list[str]:
  pats = [re.compile(e) for e in exprs]
  return [n for n in names if any(p.match(n) for p in pats)]
