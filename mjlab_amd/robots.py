"""Benchmark scenes: Unitree G1 / Go1 on a flat plane, plus small analytic models.

This restates, as plain data, the model-definition side of the reference that the
physics step consumes (SURVEY.md section 8a "model def" rows):

* actuator sets  -- reference src/mjlab/asset_zoo/robots/unitree_g1/g1_constants.py:41-186,
  unitree_go1/go1_constants.py:36-81, applied as in src/mjlab/utils/spec_config.py:400-453
  (armature on the joint, FIXED gain kp, AFFINE bias (0,-kp,-kd), forcerange = +-effort,
  ctrlrange inherited from the joint range);
* collision sets -- g1_constants.py:228-233 (``FULL_COLLISION``), go1_constants.py:119-127,
  applied as in spec_config.py:245-276 (non-matching geoms disabled);
* keyframes      -- g1_constants.py:206-219 (``KNEES_BENT_KEYFRAME``), go1_constants.py:88-97;
* contact sensors -- tasks/velocity/config/g1/rough_env_cfg.py:18-28,
  tasks/velocity/config/go1/rough_env_cfg.py:18-28, tasks/tracking/config/g1/flat_env_cfg.py:11-18;
* scene assembly -- scene/scene.py:133-147 (terrain first, then ``robot/``-prefixed entity),
  terrains/terrain_importer.py:154-163 (body ``terrain`` + plane geom ``terrain``);
* solver options -- tasks/velocity/velocity_env_cfg.py:248-256, tasks/tracking/tracking_env_cfg.py:283-291.

The robot MJCF files themselves are *not* copied into this repository: compiled
models are committed as ``mjlab_amd/assets/*.npz`` by ``tools/build_models.py`` (which
reads the XMLs from the reference checkout when it is present).
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from . import mjcf
from .mjcf import (
  GEOM_PLANE,
  INT_IMPLICITFAST,
  JNT_FREE,
  OBJ_BODY,
  OBJ_GEOM,
  OBJ_XBODY,
  Model,
  Spec,
  SpecActuator,
  SpecKey,
  SpecSensor,
  filter_exp,
  resolve_expr,
)

ASSET_DIR = Path(__file__).parent / "assets"
REFERENCE_ROOT = Path(os.environ.get("MJLAB_REFERENCE_ROOT", "/root/reference"))
_ROBOT_XML = {
  "g1": "src/mjlab/asset_zoo/robots/unitree_g1/xmls/g1.xml",
  "go1": "src/mjlab/asset_zoo/robots/unitree_go1/xmls/go1.xml",
}


@dataclass
class ActuatorCfg:
  joint_names_expr: list[str]
  effort_limit: float
  stiffness: float
  damping: float
  armature: float = 0.0
  frictionloss: float = 0.0


@dataclass
class CollisionCfg:
  geom_names_expr: list[str]
  contype: int | dict = 1
  conaffinity: int | dict = 1
  condim: int | dict = 3
  priority: int | dict = 0
  friction: tuple | dict | None = None
  solref: tuple | dict | None = None
  solimp: tuple | dict | None = None
  disable_other_geoms: bool = True


@dataclass
class ContactSensorCfg:
  name: str
  geom1: str | None = None
  body1: str | None = None
  subtree1: str | None = None
  geom2: str | None = None
  body2: str | None = None
  subtree2: str | None = None
  num: int = 1
  data: tuple[str, ...] = ("found",)
  reduce: str = "none"


@dataclass
class InitialState:
  pos: tuple[float, float, float] = (0.0, 0.0, 0.0)
  rot: tuple[float, float, float, float] = (1.0, 0.0, 0.0, 0.0)
  joint_pos: dict[str, float] = field(default_factory=lambda: {".*": 0.0})


_CONTACT_DATA = {"found": 0, "force": 1, "torque": 2, "dist": 3, "pos": 4, "normal": 5, "tangent": 6}
_CONTACT_REDUCE = {"none": 0, "mindist": 1, "maxforce": 2, "netforce": 3}


# ----------------------------------------------------------------------------
# spec edits
# ----------------------------------------------------------------------------


def apply_actuators(spec: Spec, cfgs: tuple[ActuatorCfg, ...]) -> None:
  """Position actuators in joint order (reference spec_config.py:400-453)."""
  jnts = [j for j in spec.joints if j.type != JNT_FREE]
  names = [j.name for j in jnts]
  pairs = []
  for cfg in cfgs:
    for n in filter_exp(cfg.joint_names_expr, names):
      pairs.append((cfg, n))
  if cfgs and not pairs:
    raise ValueError("No joints matched actuator patterns")
  pairs.sort(key=lambda p: names.index(p[1]))
  for cfg, n in pairs:
    j = spec.joint(n)
    if not (j.limited or j.range[0] < j.range[1]):
      raise ValueError(f"Joint {n} must be limited for position control")
    j.armature = cfg.armature
    j.frictionloss = cfg.frictionloss
    spec.actuators.append(
      SpecActuator(
        name=n,
        joint=n,
        gainprm0=cfg.stiffness,
        biasprm=(0.0, -cfg.stiffness, -cfg.damping),
        forcerange=(-cfg.effort_limit, cfg.effort_limit),
        ctrlrange=(float(j.range[0]), float(j.range[1])),  # inheritrange = 1.0
      )
    )


def _resolve_field(value, names, default):
  if isinstance(value, dict):
    return resolve_expr(value, names, default)
  return [value] * len(names)


def apply_collision(spec: Spec, cfg: CollisionCfg) -> None:
  """Collision attributes by regex (reference spec_config.py:245-276)."""
  all_names = [g.name for g in spec.geoms]
  subset = filter_exp(cfg.geom_names_expr, [n for n in all_names if n])
  res = {
    "condim": _resolve_field(cfg.condim, subset, 3),
    "contype": _resolve_field(cfg.contype, subset, 1),
    "conaffinity": _resolve_field(cfg.conaffinity, subset, 1),
    "priority": _resolve_field(cfg.priority, subset, 0),
    "friction": _resolve_field(cfg.friction, subset, None),
    "solref": _resolve_field(cfg.solref, subset, None),
    "solimp": _resolve_field(cfg.solimp, subset, None),
  }
  for i, n in enumerate(subset):
    g = spec.geom(n)
    g.condim = res["condim"][i]
    g.contype = res["contype"][i]
    g.conaffinity = res["conaffinity"][i]
    g.priority = res["priority"][i]
    for fld in ("friction", "solref", "solimp"):
      vals = res[fld][i]
      if vals is not None:
        arr = getattr(g, fld)
        for k, v in enumerate(vals):
          arr[k] = v
  if cfg.disable_other_geoms:
    keep = set(subset)
    for g in spec.geoms:
      if g.name not in keep:
        g.contype = 0
        g.conaffinity = 0


def apply_contact_sensor(spec: Spec, cfg: ContactSensorCfg) -> None:
  """``mjSENS_CONTACT`` sensor (reference spec_config.py:554-629)."""
  prim = [cfg.geom1, cfg.body1, cfg.subtree1]
  if sum(x is not None for x in prim) != 1:
    raise ValueError("Exactly one of geom1, body1, subtree1 must be specified")
  if sum(x is not None for x in (cfg.geom2, cfg.body2, cfg.subtree2)) > 1:
    raise ValueError("At most one of geom2, body2, subtree2 can be specified")
  if cfg.num <= 0:
    raise ValueError("'num' must be positive")
  vals = [_CONTACT_DATA[k] for k in cfg.data] if cfg.data else [0]
  if any(b <= a for a, b in zip(vals, vals[1:])):
    raise ValueError("Data attributes must be in order")
  dataspec = sum(1 << v for v in vals)
  if cfg.geom1 is not None:
    ot, on = OBJ_GEOM, cfg.geom1
  elif cfg.body1 is not None:
    ot, on = OBJ_BODY, cfg.body1
  else:
    ot, on = OBJ_XBODY, cfg.subtree1
  rt = rn = None
  if cfg.geom2 is not None:
    rt, rn = OBJ_GEOM, cfg.geom2
  elif cfg.body2 is not None:
    rt, rn = OBJ_BODY, cfg.body2
  elif cfg.subtree2 is not None:
    rt, rn = OBJ_XBODY, cfg.subtree2
  spec.sensors.append(SpecSensor(cfg.name, ot, on, rt, rn, (dataspec, _CONTACT_REDUCE[cfg.reduce], cfg.num)))


def add_init_keyframe(spec: Spec, init: InitialState, name: str = "init_state") -> None:
  """Keyframe = [pos, rot, joint_pos...], ctrl = joint_pos (reference entity/entity.py:146-162)."""
  comps = []
  if any(j.type == JNT_FREE for j in spec.joints):
    comps += [np.array(init.pos, dtype=np.float64), np.array(init.rot, dtype=np.float64)]
  jn = [j.name for j in spec.joints if j.type != JNT_FREE]
  jp = np.array(resolve_expr(init.joint_pos, jn), dtype=np.float64)
  comps.append(jp)
  key = SpecKey(name, np.concatenate(comps))
  if spec.actuators:
    key.ctrl = jp
  spec.keys.append(key)


# ----------------------------------------------------------------------------
# robot constants (restated; file:line in the module docstring)
# ----------------------------------------------------------------------------

_NATURAL_FREQ = 10 * 2.0 * 3.1415926535
_DAMPING_RATIO = 2.0


def _two_stage(rotor, gear):
  return rotor[0] * (gear[1] * gear[2]) ** 2 + rotor[1] * gear[2] ** 2 + rotor[2]


def _pd(armature):
  return armature * _NATURAL_FREQ**2, 2.0 * _DAMPING_RATIO * armature * _NATURAL_FREQ


def g1_actuators() -> tuple[ActuatorCfg, ...]:
  a5020 = _two_stage((0.139e-4, 0.017e-4, 0.169e-4), (1, 1 + (46 / 18), 1 + (56 / 16)))
  a7520_14 = _two_stage((0.489e-4, 0.098e-4, 0.533e-4), (1, 4.5, 1 + (48 / 22)))
  a7520_22 = _two_stage((0.489e-4, 0.109e-4, 0.738e-4), (1, 4.5, 5))
  a4010 = _two_stage((0.068e-4, 0.0, 0.0), (1, 5, 5))
  k5020, d5020 = _pd(a5020)
  k14, d14 = _pd(a7520_14)
  k22, d22 = _pd(a7520_22)
  k4010, d4010 = _pd(a4010)
  return (
    ActuatorCfg(
      [".*_elbow_joint", ".*_shoulder_pitch_joint", ".*_shoulder_roll_joint", ".*_shoulder_yaw_joint", ".*_wrist_roll_joint"],
      effort_limit=25.0, armature=a5020, stiffness=k5020, damping=d5020,
    ),
    ActuatorCfg([".*_hip_pitch_joint", ".*_hip_yaw_joint", "waist_yaw_joint"], effort_limit=88.0, armature=a7520_14, stiffness=k14, damping=d14),
    ActuatorCfg([".*_hip_roll_joint", ".*_knee_joint"], effort_limit=139.0, armature=a7520_22, stiffness=k22, damping=d22),
    ActuatorCfg([".*_wrist_pitch_joint", ".*_wrist_yaw_joint"], effort_limit=5.0, armature=a4010, stiffness=k4010, damping=d4010),
    ActuatorCfg(["waist_pitch_joint", "waist_roll_joint"], effort_limit=50.0, armature=a5020 * 2, stiffness=k5020 * 2, damping=d5020 * 2),
    ActuatorCfg([".*_ankle_pitch_joint", ".*_ankle_roll_joint"], effort_limit=50.0, armature=a5020 * 2, stiffness=k5020 * 2, damping=d5020 * 2),
  )


_G1_FOOT = r"^(left|right)_foot[1-7]_collision$"
G1_FULL_COLLISION = CollisionCfg(
  geom_names_expr=[".*_collision"],
  condim={_G1_FOOT: 3, ".*_collision": 1},
  priority={_G1_FOOT: 1},
  friction={_G1_FOOT: (0.6,)},
)
G1_KNEES_BENT = InitialState(
  pos=(0, 0, 0.76),
  joint_pos={
    ".*_hip_pitch_joint": -0.312,
    ".*_knee_joint": 0.669,
    ".*_ankle_pitch_joint": -0.363,
    ".*_elbow_joint": 0.6,
    "left_shoulder_roll_joint": 0.2,
    "left_shoulder_pitch_joint": 0.2,
    "right_shoulder_roll_joint": -0.2,
    "right_shoulder_pitch_joint": 0.2,
  },
)


def go1_actuators() -> tuple[ActuatorCfg, ...]:
  rotor = 0.000111842
  hip_arm = rotor * 6**2
  knee_arm = rotor * (6 * 1.5) ** 2
  kh, dh = _pd(hip_arm)
  kk, dk = _pd(knee_arm)
  return (
    ActuatorCfg([".*_hip_joint", ".*_thigh_joint"], effort_limit=23.7, stiffness=kh, damping=dh, armature=hip_arm),
    ActuatorCfg([".*_calf_joint"], effort_limit=35.55, stiffness=kk, damping=dk, armature=knee_arm),
  )


_GO1_FOOT = "^[FR][LR]_foot_collision$"
GO1_FULL_COLLISION = CollisionCfg(
  geom_names_expr=[".*_collision"],
  condim={_GO1_FOOT: 3, ".*_collision": 1},
  priority={_GO1_FOOT: 1},
  friction={_GO1_FOOT: (0.6,)},
  solimp={_GO1_FOOT: (0.9, 0.95, 0.023)},
  contype=1,
  conaffinity=0,
)
GO1_INIT = InitialState(
  pos=(0.0, 0.0, 0.278),
  joint_pos={".*thigh_joint": 0.9, ".*calf_joint": -1.8, ".*R_hip_joint": 0.1, ".*L_hip_joint": -0.1},
)


def action_scale(cfgs: tuple[ActuatorCfg, ...], joint_names: list[str]) -> np.ndarray:
  """0.25 * effort / stiffness per actuated joint (reference g1_constants.py:258-270)."""
  out = np.zeros(len(joint_names))
  for cfg in cfgs:
    for n in filter_exp(cfg.joint_names_expr, joint_names):
      out[joint_names.index(n)] = 0.25 * cfg.effort_limit / cfg.stiffness
  return out


# ----------------------------------------------------------------------------
# scenes
# ----------------------------------------------------------------------------


def _task_options(spec: Spec) -> None:
  o = spec.option
  o.timestep = 0.005
  o.iterations = 10
  o.ls_iterations = 20
  o.integrator = INT_IMPLICITFAST
  o.tolerance = 1e-8
  o.ls_tolerance = 0.01
  o.impratio = 1.0
  o.gravity = (0.0, 0.0, -9.81)


def build_scene(robot: Spec, key: InitialState | None, terrain_cfg=None) -> Spec:
  """World + ``terrain`` body + robot attached with prefix ``robot/``.  The terrain is the ground
  plane (reference terrain_importer.py:183-194) or, with ``terrain_cfg`` (a
  ``terrains.TerrainGeneratorCfg``), the generated boxes (terrain_importer.py:82-90); the
  sub-terrain origins are kept on the spec as ``terrain_origins``."""
  scene = Spec()
  if terrain_cfg is None:
    terrain = scene.add_body("terrain")
    scene.add_geom(terrain, "terrain", GEOM_PLANE, (0, 0, 0.01))
  else:
    from . import terrains

    scene.terrain_origins = terrains.TerrainGenerator(terrain_cfg).compile(scene).origins
  if key is not None:
    add_init_keyframe(robot, key)
  keys = robot.keys
  scene.attach(robot, prefix="robot/")
  scene.keys = keys
  return scene


def robot_xml_path(name: str) -> Path:
  p = REFERENCE_ROOT / _ROBOT_XML[name]
  if not p.exists():
    raise FileNotFoundError(
      f"{p} not found: robot MJCFs live in the reference checkout; use the compiled "
      f"models in {ASSET_DIR} (load_model) on machines without it"
    )
  return p


def g1_spec(sensors: tuple[ContactSensorCfg, ...] = ()) -> Spec:
  spec = Spec.from_file(robot_xml_path("g1"))
  for s in sensors:
    apply_contact_sensor(spec, s)
  apply_collision(spec, G1_FULL_COLLISION)
  apply_actuators(spec, g1_actuators())
  return spec


def go1_spec(sensors: tuple[ContactSensorCfg, ...] = ()) -> Spec:
  spec = Spec.from_file(robot_xml_path("go1"))
  for s in sensors:
    apply_contact_sensor(spec, s)
  apply_collision(spec, GO1_FULL_COLLISION)
  apply_actuators(spec, go1_actuators())
  return spec


def compile_scene(name: str) -> Model:
  """Compile one of the BASELINE.json scenes from the reference MJCF."""
  if name == "g1_velocity_flat":
    sensors = tuple(
      ContactSensorCfg(name=f"{s}_foot_ground_contact", body1=f"{s}_ankle_roll_link", body2="terrain", num=1, data=("found",), reduce="netforce")
      for s in ("left", "right")
    )
    spec = build_scene(g1_spec(sensors), G1_KNEES_BENT)
    # body2="terrain" refers to the un-prefixed terrain body (scene.py:147)
    for s in spec.sensors:
      if s.refname == "robot/terrain":
        s.refname = "terrain"
  elif name == "g1_tracking_flat":
    sensors = (ContactSensorCfg(name="self_collision", subtree1="pelvis", subtree2="pelvis", data=("found",), reduce="netforce", num=10),)
    spec = build_scene(g1_spec(sensors), G1_KNEES_BENT)
  elif name == "go1_velocity_flat":
    sensors = tuple(
      ContactSensorCfg(name=f"{leg}_foot_ground_contact", geom1=f"{leg}_foot_collision", body2="terrain", num=1, data=("found",), reduce="netforce")
      for leg in ("FR", "FL", "RR", "RL")
    )
    spec = build_scene(go1_spec(sensors), GO1_INIT)
    for s in spec.sensors:
      if s.refname == "robot/terrain":
        s.refname = "terrain"
  elif name == "g1_velocity_rough":
    # Mjlab-Velocity-Rough-Unitree-G1: same robot and sensors as the flat task on the generated
    # box terrain (reference tasks/velocity/config/g1/rough_env_cfg.py:13-34,
    # velocity_env_cfg.py:30-37,275-278: ROUGH_TERRAINS_CFG with the curriculum switched on).
    # The reference leaves the generator unseeded; a product build needs a fixed seed.
    from . import terrains

    sensors = tuple(
      ContactSensorCfg(name=f"{s}_foot_ground_contact", body1=f"{s}_ankle_roll_link", body2="terrain", num=1, data=("found",), reduce="netforce")
      for s in ("left", "right")
    )
    spec = build_scene(g1_spec(sensors), G1_KNEES_BENT, terrains.rough_terrains_cfg(seed=ROUGH_TERRAIN_SEED))
    for s in spec.sensors:
      if s.refname == "robot/terrain":
        s.refname = "terrain"
  elif name == "go1_velocity_rough":
    # Mjlab-Velocity-Rough-Unitree-Go1 (reference tasks/velocity/config/go1/rough_env_cfg.py:13-47)
    from . import terrains

    sensors = tuple(
      ContactSensorCfg(name=f"{leg}_foot_ground_contact", geom1=f"{leg}_foot_collision", body2="terrain", num=1, data=("found",), reduce="netforce")
      for leg in ("FR", "FL", "RR", "RL")
    )
    spec = build_scene(go1_spec(sensors), GO1_INIT, terrains.rough_terrains_cfg(seed=ROUGH_TERRAIN_SEED))
    for s in spec.sensors:
      if s.refname == "robot/terrain":
        s.refname = "terrain"
  else:
    raise KeyError(name)
  _task_options(spec)
  model = spec.compile()
  if hasattr(spec, "terrain_origins"):
    model.terrain_origins = np.asarray(spec.terrain_origins, dtype=np.float64)
  return model


ROUGH_TERRAIN_SEED = 0
SCENES = ("g1_velocity_flat", "g1_tracking_flat", "go1_velocity_flat", "g1_velocity_rough", "go1_velocity_rough")


def load_model(name: str) -> Model:
  """Load a committed compiled model (``assets/<name>.npz``)."""
  p = ASSET_DIR / f"{name}.npz"
  if not p.exists():
    raise FileNotFoundError(f"{p} missing; run tools/build_models.py on a machine with the reference checkout")
  return Model.load(p)


# ----------------------------------------------------------------------------
# analytic / fixture models (authored here; none exist in the reference tree)
# ----------------------------------------------------------------------------

PENDULUM_XML = """
<mujoco model="pendulum">
  <compiler angle="radian"/>
  <option timestep="0.002"/>
  <worldbody>
    <body name="arm" pos="0 0 2">
      <inertial pos="0 0 -0.5" mass="1" diaginertia="0.0841667 0.0841667 0.00125"/>
      <joint name="hinge" type="hinge" axis="0 1 0"/>
      <geom name="rod" type="capsule" size="0.05" fromto="0 0 0 0 0 -1" contype="0" conaffinity="0"/>
    </body>
  </worldbody>
</mujoco>
"""

BOX_XML = """
<mujoco model="box_on_plane">
  <compiler angle="radian"/>
  <option timestep="0.002"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 0.01"/>
    <body name="box" pos="0 0 0.1">
      <inertial pos="0 0 0" mass="2" diaginertia="0.0133333 0.0133333 0.0133333"/>
      <freejoint name="root"/>
      <geom name="box_geom" type="box" size="0.1 0.1 0.1"/>
    </body>
  </worldbody>
</mujoco>
"""

# Free capsule + free sphere over a plane plus a slider/hinge arm with a limited joint:
# exercises every collision primitive and constraint type on the path with a small nv.
MIXED_XML = """
<mujoco model="mixed">
  <compiler angle="radian" autolimits="true"/>
  <option timestep="0.004"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 0.01"/>
    <body name="cap" pos="0 0 0.055" quat="0.707107 0 0.707107 0">
      <inertial pos="0 0 0" mass="1.5" diaginertia="0.02 0.02 0.004"/>
      <freejoint name="cap_root"/>
      <geom name="cap_geom" type="capsule" size="0.06 0.2"/>
    </body>
    <body name="ball" pos="0.05 0.0 0.205">
      <inertial pos="0 0 0" mass="0.7" diaginertia="0.003 0.003 0.003"/>
      <freejoint name="ball_root"/>
      <geom name="ball_geom" type="sphere" size="0.1" condim="1"/>
    </body>
    <body name="cap2" pos="-0.1 0 0.16" quat="0.707107 0.707107 0 0">
      <inertial pos="0 0 0" mass="0.8" diaginertia="0.01 0.01 0.002"/>
      <freejoint name="cap2_root"/>
      <geom name="cap2_geom" type="capsule" size="0.05 0.15" friction="0.7 0.005 0.0001"/>
    </body>
    <body name="ball2" pos="0.22 0.02 0.205">
      <inertial pos="0 0 0" mass="0.3" diaginertia="0.001 0.001 0.001"/>
      <freejoint name="ball2_root"/>
      <geom name="ball2_geom" type="sphere" size="0.08" priority="1" solimp="0.9 0.95 0.01"/>
    </body>
    <body name="slider" pos="0.8 0 0.07">
      <inertial pos="0 0 0" mass="1" diaginertia="0.01 0.01 0.01"/>
      <joint name="slide" type="slide" axis="0 0 1" range="-0.3 0.3"/>
      <geom name="slider_geom" type="sphere" size="0.08"/>
      <body name="arm" pos="0 0 0">
        <inertial pos="0.2 0 0" mass="0.5" diaginertia="0.001 0.01 0.01"/>
        <joint name="elbow" type="hinge" axis="0 1 0" range="-0.05 0.05" pos="0 0 0"/>
        <geom name="arm_geom" type="capsule" size="0.04" fromto="0.1 0 0 0.4 0 0"/>
      </body>
    </body>
  </worldbody>
</mujoco>
"""


def pendulum_model(integrator: int = mjcf.INT_EULER) -> Model:
  spec = Spec.from_string(PENDULUM_XML)
  spec.option.integrator = integrator
  return spec.compile()


def box_model() -> Model:
  spec = Spec.from_string(BOX_XML)
  spec.option.integrator = INT_IMPLICITFAST
  return spec.compile()


def mixed_model() -> Model:
  spec = Spec.from_string(MIXED_XML)
  spec.option.integrator = INT_IMPLICITFAST
  apply_actuators(
    spec,
    (
      ActuatorCfg(["elbow"], effort_limit=5.0, stiffness=20.0, damping=1.0, armature=0.01),
      ActuatorCfg(["slide"], effort_limit=30.0, stiffness=100.0, damping=5.0, armature=0.0),
    ),
  )
  apply_contact_sensor(spec, ContactSensorCfg(name="cap_floor", body1="cap", geom2="floor", num=1, data=("found",), reduce="netforce"))
  return spec.compile()
