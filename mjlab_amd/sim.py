"""``Simulation``: the drop-in boundary of the reference's physics hot path.

Mirrors reference ``src/mjlab/sim/sim.py:43-198`` (``MujocoCfg``, ``SimulationCfg``,
``Simulation`` with ``step / forward / create_graph / expand_model_fields / reset / close``
and the ``model`` / ``data`` bridges), with the two foreign calls replaced:

  ``mjwarp.step(wp_model, wp_data)``     -> ``mjlab_step``     (include/mjlab_amd.h)
  ``mjwarp.forward(wp_model, wp_data)``  -> ``mjlab_forward``
  ``repeat_array_kernel``                -> ``mjlab_tile_field``
  ``wp.ScopedCapture`` / ``wp.capture_launch`` (CUDA graph) -> hipGraph capture/replay on
  PyTorch-ROCm's current stream (``torch.cuda.CUDAGraph`` is a hipGraph on ROCm).

Device arrays are allocated here as torch tensors and handed to the C ABI as raw device
pointers; everything is enqueued on torch's current stream, so writes made through
``sim.data.<field>[...] = v`` are ordered before the next ``step()`` without extra fences
(the reference needs ``torch.cuda.ExternalStream`` for that: sim_data.py:33-44,58-64).

There is no CPU path: constructing a ``Simulation`` on a device without a GPU raises.
"""

from __future__ import annotations

import ctypes
import os
import warnings
from dataclasses import dataclass, field
from typing import Any, Literal

import numpy as np
import torch

from . import _abi, device_state, native
from .mjcf import CONE_ELLIPTIC, CONE_PYRAMIDAL, INT_EULER, INT_IMPLICITFAST, SOL_CG, SOL_NEWTON, SOL_PGS, Model, Spec
from .nan_guard import NanGuard, NanGuardCfg
from .sim_data import Bridge

_CONE_MAP = {"pyramidal": CONE_PYRAMIDAL, "elliptic": CONE_ELLIPTIC}
_INTEGRATOR_MAP = {"euler": INT_EULER, "implicitfast": INT_IMPLICITFAST}
_SOLVER_MAP = {"newton": SOL_NEWTON, "cg": SOL_CG, "pgs": SOL_PGS}


@dataclass
class MujocoCfg:
  """Solver / integrator options (reference sim/sim.py:43-82; same names and defaults)."""

  timestep: float = 0.002
  integrator: Literal["euler", "implicitfast"] = "implicitfast"
  impratio: float = 1.0
  cone: Literal["pyramidal", "elliptic"] = "pyramidal"
  jacobian: Literal["auto", "dense", "sparse"] = "auto"
  solver: Literal["newton", "cg", "pgs"] = "newton"
  iterations: int = 100
  tolerance: float = 1e-8
  ls_iterations: int = 50
  ls_tolerance: float = 0.01
  gravity: tuple[float, float, float] = (0, 0, -9.81)

  def edit_spec(self, spec: Spec) -> None:
    o = spec.option
    o.cone = _CONE_MAP[self.cone]
    o.integrator = _INTEGRATOR_MAP[self.integrator]
    o.solver = _SOLVER_MAP[self.solver]
    o.timestep = self.timestep
    o.impratio = self.impratio
    o.gravity = tuple(float(g) for g in self.gravity)
    o.iterations = self.iterations
    o.tolerance = self.tolerance
    o.ls_iterations = self.ls_iterations
    o.ls_tolerance = self.ls_tolerance


# Default of SimulationCfg.ls_parallel: True like the reference's (sim/sim.py:89).  tests/conftest.py sets it to False for the suites
# that compare with the oracle's default (exact) search; bench.py, smoke() and every user run True
DEFAULT_LS_PARALLEL = True


@dataclass(kw_only=True)
class SimulationCfg:
  """Reference sim/sim.py:85-91.  ``nconmax`` is accepted for signature parity; contact
  capacity is per world here and derived from ``njmax`` (see _abi.default_capacities)."""

  nconmax: int | None = None
  njmax: int | None = None
  # True (the reference's default, sim/sim.py:89): mujoco_warp's parallel line search -- the cost at `ls_iterations` log-spaced
  # step sizes in [ls_parallel_min_step, 1], lowest cost wins (MJLAB_OPT_LS_PARALLEL, include/mjlab_fields.h; grid restated from
  # memory of mujoco_warp, unverified).  False: MuJoCo's exact iterative search (mj_solPrimal).  What the configuration says is
  # what runs: no environment variable overrides it (round 4; the test suite changes DEFAULT_LS_PARALLEL, the default of THIS
  # field, through a fixture -- an explicit value, and the reference's own SimulationCfg, always win)
  ls_parallel: bool = field(default_factory=lambda: DEFAULT_LS_PARALLEL)
  ls_parallel_min_step: float = 1.0e-6  # mujoco_warp Option.ls_parallel_min_step (not a field of the reference's cfg)
  # True: the stages compute positions in each world's local frame (origin = the floating base's position rounded to whole
  # metres: include/mjlab_fields.h, xorigin) and add the origin back in the public world-frame arrays -- a robot 100 m from
  # the origin (most of the reference's 4096 environments) is solved as accurately as one at the origin.  False: plain fp32
  # world coordinates, like the reference's engine (MJLAB_OPT_WORLD_FRAME)
  local_frame: bool = True
  mujoco: MujocoCfg = field(default_factory=MujocoCfg)
  nan_guard: NanGuardCfg = field(default_factory=NanGuardCfg)
  use_graph: bool = True
  # step() right after forward() skips the stages that would reproduce that pass bit for bit
  # (include/mjlab_amd.h, mjlab_forward); False recomputes them like the reference does
  fold_forward: bool = True
  # MuJoCo's literal Newton / line-search termination (tolerance, gtol) without the fp32
  # rounding-noise floors (MJLAB_OPT_LITERAL_TERMINATION)
  literal_termination: bool = False
  # ls_parallel only: compare the grid candidates by their literal total costs instead of by cost differences (MJLAB_OPT_LS_LITERAL_COST)
  ls_literal_cost: bool = False
  # qacc_warmstart is saved by the integrator's advance only (forward() leaves it untouched)
  # instead of at the end of every constraint solve (MJLAB_OPT_WARMSTART_AT_ADVANCE)
  warmstart_at_advance: bool = False
  # launch structure: "stage" = one kernel per pipeline stage (5 per substep), "presolve" = the four
  # pre-solve stages in one kernel, "step" = a whole substep in one kernel.  Same results bit for bit.
  # Default "step": +13 % env-steps/s over "stage" at 4096 G1 worlds, +23 % with step(nsubstep=4) (profiles/r02_v3)
  fuse: Literal["stage", "presolve", "step"] = "step"


def check_supported(model: Model) -> None:
  """Reject, loudly, every model feature the HIP kernels do not implement (instead of
  silently simulating something else).  The supported set is what BASELINE.json's
  configurations use."""
  if model.opt.solver not in (SOL_NEWTON, SOL_CG, SOL_PGS):
    raise NotImplementedError("opt.solver must be Newton, CG or PGS")
  if model.opt.cone not in (CONE_PYRAMIDAL, CONE_ELLIPTIC):
    raise NotImplementedError("opt.cone must be pyramidal or elliptic")
  if model.opt.cone == CONE_ELLIPTIC and model.opt.solver not in (SOL_NEWTON, SOL_CG):
    raise NotImplementedError("the elliptic cone is implemented for the Newton and CG solvers (pyramidal: Newton, CG, PGS)")
  if model.opt.cone == CONE_ELLIPTIC and not float(model.opt.impratio) > 0.0:
    raise ValueError("opt.impratio must be positive")
  # (contact friction is clamped at mjMINMU = 1e-5 where the collision stage mixes it, like mj_contactParam: a per-world geom_friction that
  # domain randomisation writes later cannot drive the cone rows' 1 / mu to infinity)
  if model.opt.integrator not in (INT_EULER, INT_IMPLICITFAST):
    raise NotImplementedError("integrator must be 'euler' or 'implicitfast'")
  if model.nv > 64:
    raise NotImplementedError(f"nv = {model.nv} > 64: one dof per lane of a 64-wide wavefront")
  if model.nbody > 64:
    raise NotImplementedError(f"nbody = {model.nbody} > 64: one body per lane in the kinematics sweep")
  if (np.asarray(model.jnt_type) == 1).any():
    raise NotImplementedError("ball joints are not implemented")
  for name in ("neq", "ntendon", "nmocap", "nflex"):
    if int(getattr(model, name, 0)) != 0:
      raise NotImplementedError(f"{name} != 0 is not implemented")
  dims = np.asarray(model.geom_condim)[np.asarray(model.pair_geom).reshape(-1)] if model.npair else np.zeros(0, int)
  if dims.size and not np.isin(dims, (1, 3)).all():
    raise NotImplementedError("only condim 1 and 3 contacts are implemented")
  if model.nsensor:
    spec = np.asarray(model.sensor_intprm)[:, 0]
    if (spec != 1).any():
      raise NotImplementedError("contact sensors support only the 'found' data spec")


class _NumpyView:
  """``wp.array``-like handle: ``.numpy()`` copies the field to the host
  (reference use: src/mjlab/viewer/viser.py:557,620-632)."""

  def __init__(self, t: torch.Tensor) -> None:
    self._t = t

  def numpy(self) -> np.ndarray:
    return self._t.detach().cpu().numpy()

  @property
  def shape(self):
    return tuple(self._t.shape)


class _RawStruct:
  """Stand-in for ``sim.wp_model`` / ``sim.wp_data`` (reference sim/sim.py:152-158)."""

  def __init__(self, bridge: "Bridge", opt: Any | None = None) -> None:
    object.__setattr__(self, "_bridge", bridge)
    if opt is not None:
      object.__setattr__(self, "opt", opt)

  def __getattr__(self, name: str) -> Any:
    v = getattr(object.__getattribute__(self, "_bridge"), name)
    return _NumpyView(v) if isinstance(v, torch.Tensor) else v


class HostData:
  """Host-side ``mjData`` stand-in (qpos0 state) kept for viewers/exporters
  (reference sim/sim.py:106-107,145-150)."""

  def __init__(self, model: Model) -> None:
    self.qpos = model.qpos0.copy()
    self.qvel = np.zeros(model.nv)
    self.ctrl = np.zeros(model.nu)
    self.time = 0.0




class Simulation:
  """Batched physics on one MI355X; one world per wavefront (see csrc/mjlab_amd.hip)."""

  def __init__(self, num_envs: int, cfg: SimulationCfg, model: Model, device: str) -> None:
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
      raise RuntimeError(
        f"mjlab_amd.Simulation needs a ROCm GPU device (got '{device}', "
        f"torch.cuda.is_available()={torch.cuda.is_available()}); there is no CPU fallback"
      )
    # The reference hands over a ``mujoco.MjModel`` (sim/sim.py:97-99).  Anything that is not this
    # package's own host model is read through mjModel's attribute names (from_mujoco.py); the
    # object itself stays available as ``mj_model`` for viewers / exporters, as in the reference.
    self._given_model = model
    if not isinstance(model, Model):
      from .from_mujoco import model_from_mujoco

      model = model_from_mujoco(model)
    check_supported(model)
    self.cfg = cfg
    self.device = device
    self._dev = dev
    self.num_envs = num_envs
    self._mj_model = model
    self._mj_data = HostData(model)
    self._lib = native.lib()
    self.ls_parallel = bool(getattr(cfg, "ls_parallel", True))
    mf, df, MS, DS = native.layouts()
    self._mfields = {f.name: f for f in mf}
    self._dfields = {f.name: f for f in df}
    self.nconmax, self.njmax = _abi.default_capacities(model, cfg.nconmax, cfg.njmax)

    self._m, self._model_base, self._model_view = device_state.upload_model(model, num_envs, self.nconmax, self.njmax, dev)
    # `cfg` may be the reference's own SimulationCfg (sim/sim.py:85-91: nconmax, njmax, ls_parallel, mujoco, nan_guard) when its
    # environment classes construct this Simulation (tools/reference_env.py): this package's extra switches then take their defaults
    ext = SimulationCfg()
    opt = lambda name: getattr(cfg, name, getattr(ext, name))  # noqa: E731
    # the dual solver (MujocoCfg.solver = "pgs") exists as a stage kernel only: the fused launch structures carry the primal solvers
    # ... elliptic friction cones (MujocoCfg.cone = "elliptic": csrc/stage_cone.h) have their own variants of the per-stage and the "step"
    # structures (and of the control kernel), not of "presolve"
    self.fuse = "stage" if (model.opt.solver == SOL_PGS or (model.opt.cone == CONE_ELLIPTIC and opt("fuse") == "presolve")) else opt("fuse")
    self._m.opt.ls_parallel_min_step = float(opt("ls_parallel_min_step"))
    self._m.opt.flags = ((self._m.opt.flags & _abi.OPT_FRICTIONLOSS) | (_abi.OPT_FOLD_FORWARD if opt("fold_forward") else 0)
                         | (_abi.OPT_LITERAL_TERMINATION if opt("literal_termination") else 0)
                         | (_abi.OPT_WARMSTART_AT_ADVANCE if opt("warmstart_at_advance") else 0)
                         | (_abi.OPT_LS_PARALLEL if self.ls_parallel else 0)
                         | (_abi.OPT_LS_LITERAL_COST if opt("ls_literal_cost") else 0)
                         | (0 if (opt("local_frame") and os.environ.get("MJLAB_LOCAL_FRAME", "1") != "0") else _abi.OPT_WORLD_FRAME)
                         | {"stage": 0, "presolve": _abi.OPT_FUSE_PRESOLVE, "step": _abi.OPT_FUSE_STEP}[self.fuse])
    self._d, self._data = device_state.alloc_data(model, num_envs, self.nconmax, self.njmax, dev)

    scalars = {k: int(getattr(model, k)) for k in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "nsensor", "nsensordata")}
    self._expanded: set[str] = set()
    self._model_bridge = Bridge("sim.model", self._model_view, {**scalars, "opt": model.opt, "nworld": num_envs},
                                on_access=self._on_model_access)
    self._data_bridge = Bridge("sim.data", self._data, {"nworld": num_envs, "njmax": self.njmax, "nconmax": self.nconmax})

    # reference sim/sim.py:129 (opt-in; its history ring and non-finite flag live on the device, nan_guard.py)
    self.nan_guard = NanGuard(cfg.nan_guard, num_envs, model)
    self._step_calls = 0
    self.priority_refresh = not os.environ.get("MJLAB_NO_PRIORITY_REFRESH")
    q = os.environ.get("MJLAB_PRIO_Q")
    self._prio_q = torch.tensor([float(x) for x in q.split(",")] if q else list(self.PRIORITY_QUANTILES), device=dev)
    self.post_step_hooks: list = []  # callables(sim) run after every step() (where the reference's NaN guard sits)
    self.use_graph = bool(opt("use_graph")) and not os.environ.get("MJLAB_AMD_NO_GRAPH")
    self.step_graph: torch.cuda.CUDAGraph | None = None
    self.forward_graph: torch.cuda.CUDAGraph | None = None
    # populate derived fields like mjwarp.put_data does from mj_forward; this first pass also
    # writes the poses of the static geoms (world / terrain bodies), which later passes skip
    self._m.size.nstaticgeom = self._m.size.nstaticsite = 0
    flags = self._m.opt.flags
    self._m.opt.flags = flags | _abi.OPT_WORLD_FRAME  # stored world poses must not depend on where the robots are right now
    self.forward()
    self._m.opt.flags = flags
    self._m.size.nstaticgeom, self._m.size.nstaticsite = int(model.nstaticgeom), int(model.nstaticsite)
    if not (flags & _abi.OPT_WORLD_FRAME):
      self.forward()  # the same state in the local frame (xorigin, the hand-over arrays)
    self.create_graph()
    # pay the one-time start-up cost of the device ops of update_priority_thresholds() here, outside any timed loop; the
    # thresholds themselves stay unset (all worlds are identical at this point): the kernels' row-count classes apply
    # until the first refresh
    self.update_priority_thresholds()
    self._data["sched_thr"].zero_()

  # ------------------------------------------------------------------ helpers
  def _on_model_access(self, name: str) -> None:
    """A per-world (expanded) model field handed out may be written through (domain
    randomisation): the last forward pass no longer describes the model, so the next step must
    not reuse it.  Shared fields are read-only broadcasts and do not matter."""
    f = self._mfields.get(name)
    if f is None or f.kind != "r":
      return  # int topology fields are shared and read-only for the kernels' purposes
    if name in self._expanded or self.num_envs == 1:  # writable storage (a 1-world view aliases the base array)
      self._data["fold_valid"].zero_()
      if name == "dof_frictionloss" and not self._m.opt.flags & _abi.OPT_FRICTIONLOSS:
        self._m.opt.flags |= _abi.OPT_FRICTIONLOSS  # values may now be written: the constraint stage builds friction-loss rows
        if self.step_graph is not None or self.forward_graph is not None:  # captured launches hold the old flags
          self.create_graph()

  def _stream(self) -> int:
    return torch.cuda.current_stream(self._dev).cuda_stream

  def _launch_step(self, nsubstep: int = 1) -> None:
    native.check(self._lib.mjlab_step(ctypes.byref(self._m), ctypes.byref(self._d), nsubstep, self._stream()), "mjlab_step")

  def _launch_forward(self) -> None:
    native.check(self._lib.mjlab_forward(ctypes.byref(self._m), ctypes.byref(self._d), self._stream()), "mjlab_forward")

  def forward_stages(self, stages: int) -> None:
    """Run selected pipeline stages once (testing / profiling)."""
    with torch.cuda.device(self._dev):
      native.check(
        self._lib.mjlab_forward_stages(ctypes.byref(self._m), ctypes.byref(self._d), stages, self._stream()),
        "mjlab_forward_stages",
      )

  # ------------------------------------------------------------------ reference surface
  def create_graph(self) -> None:
    """(Re-)capture the step / forward launch sequences into hipGraphs
    (reference sim/sim.py:131-140)."""
    self.step_graph = None
    self.forward_graph = None
    if not self.use_graph:
      return
    with torch.cuda.device(self._dev):
      torch.cuda.synchronize(self._dev)
      # Graph capture re-runs nothing: the captured launches only execute on replay, but
      # capture itself must not mutate state, which holds because launches are recorded.
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g):
        self._launch_step(1)
      self.step_graph = g
      g2 = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g2):
        self._launch_forward()
      self.forward_graph = g2

  @property
  def mj_model(self):
    """The model object the caller passed in (reference sim/sim.py:145-150 keeps the MjModel)."""
    return self._given_model

  @property
  def host_model(self) -> Model:
    """This package's host-side model (mjModel-named numpy arrays + the derived tables)."""
    return self._mj_model

  @property
  def mj_data(self) -> HostData:
    return self._mj_data

  @property
  def wp_model(self) -> _RawStruct:
    return _RawStruct(self._model_bridge, self._mj_model.opt)

  @property
  def wp_data(self) -> _RawStruct:
    return _RawStruct(self._data_bridge)

  @property
  def data(self) -> Bridge:
    return self._data_bridge

  @property
  def model(self) -> Bridge:
    return self._model_bridge

  def expand_model_fields(self, fields: list[str]) -> None:
    """Give each listed model field a per-world copy (reference sim/sim.py:170-176,
    sim/randomization.py:20-55).  Raises ValueError for unknown fields."""
    invalid = [f for f in fields if not hasattr(self._mj_model, f)]
    if invalid:
      raise ValueError(f"Fields not found in model: {invalid}")
    if self.num_envs == 1:
      return
    moves_static = [f for f in fields if f in ("geom_pos", "geom_quat", "body_pos", "body_quat")]
    if moves_static:
      if self._mj_model.nterrain:
        raise NotImplementedError(f"per-world {moves_static} with a box terrain: terrain boxes are static and shared by all worlds")
      self._m.size.nstaticgeom = 0  # static geoms may now differ per world: recompute them every pass
    if any(f in ("site_pos", "site_quat", "body_pos", "body_quat") for f in fields):
      self._m.size.nstaticsite = 0
    with torch.cuda.device(self._dev):
      for name in fields:
        if device_state.expand_field(self._m, self._model_base, self._model_view, self._mj_model, name, self.num_envs, self._dev, self._stream()):
          self._expanded.add(name)
      self._data["fold_valid"].zero_()
      # pointers changed: captured graphs are stale (the reference re-captures too:
      # envs/manager_based_rl_env.py:102-104)
      self.step_graph = None
      self.forward_graph = None

  def reset(self) -> None:
    pass  # reference sim/sim.py:178-180

  def forward(self, env_mask: torch.Tensor | None = None) -> None:
    """``forward()`` is the reference call (all worlds, sim/sim.py:182-187).  ``env_mask`` (bool
    or int tensor of shape ``(num_envs,)``) is an extension: only the marked worlds are
    recomputed; it is enqueued without any host sync."""
    with torch.cuda.device(self._dev):
      if env_mask is not None:
        self._data["world_mask"].copy_(env_mask.to(torch.int32).view(-1))
        native.check(self._lib.mjlab_forward_masked(ctypes.byref(self._m), ctypes.byref(self._d), self._stream()), "mjlab_forward_masked")
        return
      if self.use_graph and self.forward_graph is not None:
        self.forward_graph.replay()
      else:
        self._launch_forward()

  def forward_if(self, flag: torch.Tensor) -> None:
    """``forward()`` on ALL worlds if the device scalar ``flag`` is positive, on none otherwise -- the reference's "forward() iff some
    environment reset" (envs/manager_based_rl_env.py:129-132) decided on the device, no host sync: one launch fills ``world_mask``, then
    ``mjlab_forward_masked``."""
    if flag.dtype != torch.float32 or flag.numel() != 1 or flag.device != self._data["world_mask"].device:
      raise TypeError("forward_if: a float32 device scalar is expected")
    with torch.cuda.device(self._dev):
      native.check(self._lib.mjlab_flag_to_mask(flag.data_ptr(), self.num_envs, self._data["world_mask"].data_ptr(), self._stream()), "mjlab_flag_to_mask")
      native.check(self._lib.mjlab_forward_masked(ctypes.byref(self._m), ctypes.byref(self._d), self._stream()), "mjlab_forward_masked")

  def step(self, nsubstep: int = 1) -> None:
    """``step()`` is the reference call (one physics step, sim/sim.py:189-195).  ``nsubstep`` > 1 is an
    extension: that many steps with the inputs held fixed -- the reference's decimation loop
    (envs/manager_based_rl_env.py:109-114 re-applies the same action before each of them) as one call;
    with ``fuse="step"`` also ONE kernel launch."""
    with torch.cuda.device(self._dev):
      if self.nan_guard.enabled:
        self.nan_guard.capture(self._data_bridge)
      if nsubstep == 1:
        self._step_once()
      else:
        self._launch_step(nsubstep)
      if self.nan_guard.enabled:
        self.nan_guard.check_and_dump(self._data_bridge)
      self._step_calls += 1
      if self._step_calls % 64 == 0:
        self.update_priority_thresholds()
      for hook in self.post_step_hooks:
        hook(self)

  # quantiles of the per-world score nefc x (solver_niter + 2) that separate the wave-priority classes 0 | 1 | 2 | 3
  PRIORITY_QUANTILES = (0.45, 0.85, 0.95)

  def update_priority_thresholds(self) -> None:
    """Refresh ``data.sched_thr`` (include/mjlab_fields.h) from the current batch: a handful of small device ops, no host
    sync.  ``step()`` does it every 64th call, ``PhysicsRollout.step`` every 16th control step; the kernels use their
    built-in row-count thresholds until the first refresh.  A scheduling hint: results do not depend on it."""
    if self.num_envs < 64 or not self.priority_refresh or torch.cuda.is_current_stream_capturing():
      return
    d = self._data
    score = (d["nefc"].view(-1) * (d["solver_niter"].view(-1) + 2)).float()
    q = torch.quantile(score, self._prio_q)
    d["sched_thr"].view(-1)[:3] = q.clamp_(min=1.0).to(torch.int32)

  def invalidate_fold(self) -> None:
    """Forget the last forward pass ("forward folded into the next step").  Needed only by callers
    that write to a model tensor obtained EARLIER (a cached handle): every ``sim.model.<field>``
    access of a writable field does this by itself."""
    self._data["fold_valid"].zero_()

  def overflow_report(self) -> dict[str, int]:
    """Worlds whose last collision / constraint pass dropped work for lack of capacity
    (``data.overflow`` bits, include/mjlab_fields.h); one host sync.  Warns when any did."""
    o = self._data["overflow"].view(-1)
    rep = {"nconmax": int((o & _abi.OVF_NCONMAX).ne(0).sum()), "njmax": int((o & _abi.OVF_NJMAX).ne(0).sum()),
           "terrain_candidates": int((o & _abi.OVF_TCAND).ne(0).sum())}
    if any(rep.values()):
      warnings.warn(f"capacity overflow (worlds affected): {rep}; raise SimulationCfg.njmax / nconmax", stacklevel=2)
    return rep

  def _step_once(self) -> None:
    if self.use_graph and self.step_graph is not None:
      self.step_graph.replay()
    else:
      self._launch_step(1)

  def close(self) -> None:
    pass

  # ------------------------------------------------------------------ extras
  def lds_bytes(self) -> dict[str, int]:
    names = {"position": 1, "collision": 2, "velocity": 4, "constraint": 8, "solve": 16}
    return {k: int(self._lib.mjlab_lds_bytes(ctypes.byref(self._m), v)) for k, v in names.items()}
