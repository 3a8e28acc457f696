"""Run a REGISTERED TASK OF THE REFERENCE -- its own ``ManagerBasedRlEnv``, ``Scene``, ``Entity``, managers, MDP terms and
task configuration, unmodified -- on top of ``mjlab_amd.Simulation`` (VERDICT round 2, row N1 / "do this" item 3).

What is replaced, and only this:
  * ``import mujoco``           -> ``mjlab_amd.mujoco_shim`` (the model-building API over this package's MJCF compiler);
  * ``mjlab.sim.Simulation``    -> ``mjlab_amd.sim.Simulation`` (the three names the reference binds it under);
  * third-party packages that are not installed here and are not on the physics path (``warp``, ``mujoco_warp``: imported
    for type names and the CUDA-only kernel of ``sim/randomization.py``; ``gymnasium``: spaces and the ``Env`` base class;
    ``prettytable``: the managers' ``__str__``; ``tyro`` / ``rsl_rl`` / ``tensordict`` / viewers: never reached) -> inert stubs.

The reference source is read from ``$MJLAB_REFERENCE_SRC``, ``/root/reference/src`` (build container) or ``gpurun_ref/src``
(a copy staged next to the repository for ONE gpurun call by ``tools/stage_reference.sh``; git-ignored, never committed,
removed afterwards): the GPU box has no ``/root/reference``.  ``locate_reference()`` returns None when none exists and the
callers (tests, ``bench.py --full-env``) skip.
"""

from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import math
import os
import sys
import types
from pathlib import Path
from typing import Any
from unittest import mock

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
  sys.path.insert(0, str(ROOT))

TASKS = {
  "Mjlab-Velocity-Flat-Unitree-G1": ("mjlab.tasks.velocity.config.g1.flat_env_cfg", "UnitreeG1FlatEnvCfg"),
  "Mjlab-Velocity-Rough-Unitree-G1": ("mjlab.tasks.velocity.config.g1.rough_env_cfg", "UnitreeG1RoughEnvCfg"),
  "Mjlab-Velocity-Flat-Unitree-Go1": ("mjlab.tasks.velocity.config.go1.flat_env_cfg", "UnitreeGo1FlatEnvCfg"),
  "Mjlab-Velocity-Rough-Unitree-Go1": ("mjlab.tasks.velocity.config.go1.rough_env_cfg", "UnitreeGo1RoughEnvCfg"),
  "Mjlab-Tracking-Flat-Unitree-G1": ("mjlab.tasks.tracking.config.g1.flat_env_cfg", "G1FlatEnvCfg"),
  "Mjlab-Tracking-Flat-Unitree-G1-No-State-Estimation": ("mjlab.tasks.tracking.config.g1.flat_env_cfg", "G1FlatNoStateEstimationEnvCfg"),
}
GENERIC_STUBS = ("mujoco_warp", "tyro", "rsl_rl", "tensordict", "trimesh", "viser", "wandb", "moviepy", "glfw", "OpenGL", "imageio", "mediapy",
                 "onnx", "onnxruntime", "PIL", "cv2")


def locate_reference() -> Path | None:
  for cand in (os.environ.get("MJLAB_REFERENCE_SRC"), "/root/reference/src", str(ROOT / "gpurun_ref" / "src")):
    if cand and (Path(cand) / "mjlab" / "__init__.py").exists():
      return Path(cand)
  return None


# ---------------------------------------------------------------------------------------------------------------- stubs
class _StubModule(types.ModuleType):
  """CamelCase attributes are inert classes (so ``class X(stub.Base)`` and annotations work), the rest MagicMocks."""

  def __getattr__(self, name: str) -> Any:
    if name == "__version__":
      return "0.0.0-stub"
    if name.startswith("__"):
      raise AttributeError(name)
    val = type(name, (), {"__init__": lambda self, *a, **k: None}) if name[:1].isupper() else mock.MagicMock(name=name)
    setattr(self, name, val)
    return val


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
  def find_spec(self, name, path=None, target=None):
    if name.split(".")[0] in GENERIC_STUBS:
      return importlib.machinery.ModuleSpec(name, self, is_package=True)
    return None

  def create_module(self, spec):
    m = _StubModule(spec.name)
    m.__path__ = []
    return m

  def exec_module(self, module):
    pass


def _warp_stub() -> types.ModuleType:
  """``warp`` as the reference's import-time code needs it: ``@wp.kernel`` on ``repeat_array_kernel``
  (sim/randomization.py:9-17), ``wp.array(dtype=...)`` in its annotations, ``wp.config.version``
  (envs/manager_based_rl_env.py:37).  Nothing of it runs: ``Simulation`` is this package's."""
  wp = _StubModule("warp")
  wp.__path__ = []
  wp.kernel = lambda f=None, **k: f if f is not None else (lambda g: g)
  wp.func = wp.kernel
  wp.array = lambda *a, **k: None
  wp.config = types.SimpleNamespace(version="0.0.0-stub", quiet=True)
  for name in ("float32", "int32", "vec3", "quat", "mat33"):
    setattr(wp, name, type(name, (), {"__init__": lambda self, *a, **k: None}))
  wp.rand_init = lambda *a, **k: 0  # utils/random.py:20 seeds warp's RNG too
  return wp


def _gymnasium_stub() -> types.ModuleType:
  """The part of gymnasium the env classes touch: ``Env`` as a base class, ``spaces.Box`` / ``spaces.Dict``,
  ``vector.utils.batch_space`` (envs/manager_based_rl_env.py:33,196-232), ``register`` (task packages)."""
  gym = types.ModuleType("gymnasium")
  gym.__path__ = []

  class Env:
    metadata: dict = {}

  class Box:
    def __init__(self, low=-math.inf, high=math.inf, shape=(), dtype=None) -> None:
      self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

  class Dict(dict):
    def __init__(self, spaces=None, **kw) -> None:
      super().__init__(spaces or {}, **kw)

  def batch_space(space, n: int):
    if isinstance(space, Dict):
      return Dict({k: batch_space(v, n) for k, v in space.items()})
    return Box(space.low, space.high, (n, *space.shape), space.dtype)

  spaces = types.ModuleType("gymnasium.spaces")
  spaces.Box, spaces.Dict, spaces.Space = Box, Dict, object
  vector = types.ModuleType("gymnasium.vector")
  vutils = types.ModuleType("gymnasium.vector.utils")
  vutils.batch_space = batch_space
  vector.utils = vutils
  registry: dict[str, dict] = {}
  gym.Env, gym.spaces, gym.vector, gym.registry, gym.Space = Env, spaces, vector, registry, object
  gym.register = lambda id, **kw: registry.__setitem__(id, kw)

  def spec(id: str):  # gymnasium.spec(): the registered EnvSpec; the reference reads .kwargs of it (isaaclab_tasks/utils/parse_cfg.py:54)
    if id not in registry:
      raise KeyError(f"No registered env with id: {id}")
    return types.SimpleNamespace(id=id, entry_point=registry[id].get("entry_point"), kwargs=dict(registry[id].get("kwargs", {})))

  gym.spec = spec
  gym.Wrapper = type("Wrapper", (), {})
  sys.modules.update({"gymnasium.spaces": spaces, "gymnasium.vector": vector, "gymnasium.vector.utils": vutils})
  return gym


def _prettytable_stub() -> types.ModuleType:
  pt = types.ModuleType("prettytable")

  class PrettyTable:
    def __init__(self, *a, **k) -> None:
      self.title, self.field_names, self.align, self._rows = "", [], {}, []

    def add_row(self, row) -> None:
      self._rows.append(list(row))

    def get_string(self, *a, **k) -> str:
      return "\n".join([str(self.title), " | ".join(map(str, self.field_names))] + [" | ".join(map(str, r)) for r in self._rows])

    __str__ = get_string

  pt.PrettyTable = PrettyTable
  return pt


def install_stubs(reference_src: Path) -> None:
  """Idempotent: the shim as ``mujoco``, the stubs, the reference source on sys.path, no ``__pycache__`` in it."""
  sys.dont_write_bytecode = True  # the reference tree is read-only by contract
  from mjlab_amd import mujoco_shim

  mujoco_shim.install(force=not _real_module("mujoco"))
  for name, make in (("warp", _warp_stub), ("gymnasium", _gymnasium_stub), ("prettytable", _prettytable_stub)):
    if _real_module(name) or getattr(sys.modules.get(name), "_mjlab_amd_stub", False):
      continue
    sys.modules[name] = make()
    sys.modules[name]._mjlab_amd_stub = True
  if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _StubFinder())
  if str(reference_src) not in sys.path:
    sys.path.insert(0, str(reference_src))


def _real_module(name: str) -> bool:
  """Is an installed (file-backed) distribution of `name` importable?  (None of them is on the MI355X image.)"""
  mod = sys.modules.get(name)
  if mod is not None:
    return getattr(mod, "__file__", None) is not None and "mjlab_amd" not in str(getattr(mod, "__file__", ""))
  try:
    return importlib.util.find_spec(name) is not None
  except (ImportError, ValueError):
    return False


# ------------------------------------------------------------------------------------------------------------------ env
def make_env(task: str = "Mjlab-Velocity-Flat-Unitree-G1", num_envs: int = 256, device: str = "cuda:0", sim_cls: Any = None, seed: int = 42,
             cfg_edit: Any = None):
  """-> the reference's ``ManagerBasedRlEnv`` for `task` with `num_envs` worlds over ``sim_cls`` (default:
  ``mjlab_amd.sim.Simulation``).  ``cfg_edit(cfg)`` may adjust the task config before construction."""
  src = locate_reference()
  if src is None:
    raise FileNotFoundError("reference source not found (MJLAB_REFERENCE_SRC, /root/reference/src, gpurun_ref/src)")
  install_stubs(src)
  if sim_cls is None:
    from mjlab_amd.sim import Simulation as sim_cls
  import mjlab.envs.manager_based_env as mbe
  import mjlab.sim as msim
  import mjlab.sim.sim as msimsim

  # THE substitution (everything else of mjlab.* runs as it is)
  mbe.Simulation = msim.Simulation = msimsim.Simulation = sim_cls
  from mjlab.envs import ManagerBasedRlEnv

  modname, clsname = TASKS[task]
  import copy

  # a private copy: the reference's task configs hold shared default objects (SceneEntityCfg, term params) that the managers
  # resolve IN PLACE while an environment is built -- a second environment of the task in the same process would otherwise
  # receive half-resolved ones
  cfg = copy.deepcopy(getattr(importlib.import_module(modname), clsname)())
  cfg.scene.num_envs = num_envs
  cfg.seed = seed
  if cfg_edit is not None:
    cfg_edit(cfg)
  return ManagerBasedRlEnv(cfg, device)


def random_rollout(env, steps: int, seed: int = 0, on_step: Any = None) -> dict:
  """`steps` calls of ``env.step`` with the reference's "random" policy (scripts/play.py:159-172: 2 U(0,1) - 1)."""
  import torch

  gen = torch.Generator(device=env.device)
  gen.manual_seed(seed)
  na = sum(env.action_manager.action_term_dim)
  obs, _ = env.reset()
  nreset, rew_sum = 0, 0.0
  for k in range(steps):
    action = 2.0 * torch.rand((env.num_envs, na), device=env.device, generator=gen) - 1.0
    obs, rew, terminated, time_out, _ = env.step(action)
    nreset += int((terminated | time_out).sum())
    rew_sum += float(rew.mean())
    if on_step is not None:
      on_step(k, obs, rew, terminated, time_out)
  return {"obs": obs, "resets": nreset, "mean_reward": rew_sum / max(steps, 1)}
