"""C-ABI library: loads, exports every symbol include/mjlab_amd.h declares, layouts agree."""

import ctypes
import re
from pathlib import Path

import pytest

from mjlab_amd import _abi, native, robots

ROOT = Path(__file__).resolve().parents[1]


def test_exports_every_declared_symbol():
  L = native.lib()
  header = (ROOT / "include" / "mjlab_amd.h").read_text()
  declared = set(re.findall(r"\b(mjlab_[a-z_]+)\s*\(", header))
  assert declared == set(native.EXPORTED_SYMBOLS)
  for sym in declared:
    assert hasattr(L, sym), sym
  assert L.mjlab_abi_version() == native.ABI_VERSION == 4


def test_layout_matches_struct_sizes():
  mf, df, MS, DS = native.layouts()
  L = native.lib()
  assert ctypes.sizeof(MS) == L.mjlab_sizeof_model()
  assert ctypes.sizeof(DS) == L.mjlab_sizeof_data()
  assert {"qpos", "qvel", "ctrl", "xpos", "xquat", "cvel", "subtree_com", "sensordata", "actuator_force"} <= {f.name for f in df}
  m = robots.load_model("g1_velocity_flat")
  for f in mf:  # every model field the kernels need exists on the host model
    assert hasattr(m, f.name), f.name


def test_oracle_and_product_share_layout():
  from oracle.oracle import _load

  o = _load("f64")
  L = native.lib()
  assert o.mjo_model_layout() == L.mjlab_model_layout()
  assert o.mjo_data_layout() == L.mjlab_data_layout()


def test_no_cpu_fallback():
  from mjlab_amd.sim import Simulation, SimulationCfg

  with pytest.raises(RuntimeError, match="no CPU fallback"):
    Simulation(2, SimulationCfg(), robots.pendulum_model(), "cpu")


def test_control_struct_mirror_matches_the_library():
  import ctypes

  from mjlab_amd.entity_data import _View
  from mjlab_amd.rollout import _Control, _PushRange

  L = native.lib()
  assert ctypes.sizeof(_Control) == L.mjlab_sizeof_control()
  from mjlab_amd.rollout import _MotionReset

  assert ctypes.sizeof(_MotionReset) == L.mjlab_sizeof_motion_reset()  # mjlab_control_t.motion points to one of these on the device
  assert ctypes.sizeof(_PushRange) == 48 and ctypes.sizeof(_View) % 8 == 0


def test_product_does_not_import_oracle():
  for p in (ROOT / "mjlab_amd").rglob("*.py"):
    txt = p.read_text()
    assert "import oracle" not in txt and "from oracle" not in txt, p
  for p in (ROOT / "mjlab_amd" / "csrc").glob("*"):
    if p.suffix in (".hip", ".h", ".cpp"):
      assert "oracle" not in p.read_text().lower(), p


def test_default_capacities():
  m = robots.load_model("g1_velocity_flat")
  ncon, njmax = _abi.default_capacities(m, 140_000, 300)
  assert njmax == 300 and ncon == 300


def test_every_bundled_scene_keeps_16_waves_per_cu_in_lds():
  """One world per wave, all 4096 worlds of a 4096-env batch resident at once: 16 waves per CU, i.e. at most 10 KB of
  LDS per wave in every stage (160 KB per CU).  The solve stage of the G1 is the tight one: H / factor, packed M,
  128 rows of the three per-row arrays, scratch = 10 104 B."""
  import ctypes

  _, _, MS, _ = native.layouts()
  L = native.lib()
  stages = {"position": 1, "collision": 2, "velocity": 4, "constraint": 8, "solve": 16}
  for scene in robots.SCENES:
    model = robots.load_model(scene)
    ms = MS()
    ms.size = _abi.fill_sizes(model, 4096, *_abi.default_capacities(model, None, 300))
    ms.opt = _abi.fill_option(model)  # (the solve stage's LDS block depends on the solver: the dual solver has its own layout)
    lds = {k: L.mjlab_lds_bytes(ctypes.byref(ms), v) for k, v in stages.items()}
    assert all(0 < b <= 10240 for b in lds.values()), (scene, lds)
    if scene.startswith("g1"):
      assert lds["solve"] == 10104, lds
      # elliptic cones (stage_cone.h): their own kernels and LDS layout -- the factor's block, packed M, 168 rows of the three per-row arrays,
      # role bits: 10 KB per wave like the pyramid's solve
      ms.opt.cone = 1
      ell = L.mjlab_lds_bytes(ctypes.byref(ms), 16)
      assert ell == 10228 and ell <= 10240, ell
