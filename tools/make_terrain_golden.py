"""Record the box terrains the REFERENCE'S OWN generator produces into
tests/golden/terrain_reference.npz.

The reference's terrain code (src/mjlab/terrains/*.py) is plain Python + numpy on top of
``mujoco.MjSpec``; the only thing it asks of the spec is ``add_body`` / ``add_geom`` and a few
geom attributes.  This script imports those modules with ``mujoco`` (absent here) replaced by a
stub, hands them a recording fake spec and stores, for seeded configurations of the reference's
``ROUGH_TERRAINS_CFG`` (terrains/config.py:7-57): every box (centre, half size) in geom order,
the sub-terrain origins and the environment origins of ``TerrainImporter``
(terrain_importer.py:196-229, for the level/type assignment given explicitly).
tests/test_terrains.py checks mjlab_amd/terrains.py against the file.

Run in the build container (needs /root/reference):  python tools/make_terrain_golden.py
"""

from __future__ import annotations

import copy
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
from make_reference_pins import REF, _StubFinder  # noqa: E402


class _Geom:
  def __init__(self, type=None, size=(0, 0, 0), pos=(0, 0, 0), **kw):
    self.type, self.size, self.pos = type, np.array(size, dtype=np.float64), np.array(pos, dtype=np.float64)
    self.material = None
    self.rgba = np.zeros(4)
    self.name = ""


class _Body:
  def __init__(self, name=""):
    self.name, self.geoms, self.bodies = name, [], []

  def add_geom(self, **kw):
    g = _Geom(**kw)
    self.geoms.append(g)
    return g

  def add_body(self, name=""):
    b = _Body(name)
    self.bodies.append(b)
    return b

  def add_light(self, **kw):
    return None

  def add_site(self, **kw):
    return None


class _Spec:
  def __init__(self):
    self.worldbody = _Body("world")

  def body(self, name):
    return next(b for b in self.worldbody.bodies if b.name == name)


def main() -> None:
  sys.meta_path.insert(0, _StubFinder())
  sys.path.insert(0, str(REF / "src"))
  from mjlab.terrains.config import ROUGH_TERRAINS_CFG
  from mjlab.terrains.terrain_generator import TerrainGenerator

  out: dict[str, np.ndarray] = {}
  cases = {
    "curriculum_10x20_seed0": dict(seed=0, curriculum=True, num_rows=10, num_cols=20),
    "random_3x5_seed7": dict(seed=7, curriculum=False, num_rows=3, num_cols=5),
    "curriculum_2x3_seed1": dict(seed=1, curriculum=True, num_rows=2, num_cols=3),
  }
  for name, kw in cases.items():
    cfg = copy.deepcopy(ROUGH_TERRAINS_CFG)
    for k, v in kw.items():
      setattr(cfg, k, v)
    gen = TerrainGenerator(cfg, device="cpu")
    spec = _Spec()
    gen.compile(spec)
    geoms = spec.body("terrain").geoms
    assert [g.name for g in geoms] == [f"terrain_{i}" for i in range(len(geoms))]
    out[name + "/boxes"] = np.array([np.concatenate([g.pos, g.size]) for g in geoms])
    out[name + "/origins"] = np.array(gen.terrain_origins)
    out[name + "/cfg"] = np.array([kw["seed"], int(kw["curriculum"]), kw["num_rows"], kw["num_cols"]])
    print(name, len(geoms), "boxes")
  # the fourth box terrain of the reference (not part of ROUGH_TERRAINS_CFG): random grid, in its
  # three modes, through the same generator
  import mjlab.terrains as tg
  from mjlab.terrains.terrain_generator import TerrainGeneratorCfg

  grid_modes = {"plain": {}, "merged": {"merge_similar_heights": True}, "holes": {"holes": True}}
  for mode, kw in grid_modes.items():
    cfg = TerrainGeneratorCfg(
      size=(8.0, 8.0), seed=11, num_rows=2, num_cols=2, border_width=1.0,
      sub_terrains={"grid": tg.BoxRandomGridTerrainCfg(grid_width=0.45, grid_height_range=(0.05, 0.2), platform_width=2.0, **kw)},
    )
    gen = TerrainGenerator(cfg, device="cpu")
    spec = _Spec()
    gen.compile(spec)
    geoms = spec.body("terrain").geoms
    out[f"grid_{mode}/boxes"] = np.array([np.concatenate([g.pos, g.size]) for g in geoms])
    out[f"grid_{mode}/origins"] = np.array(gen.terrain_origins)
    print("grid", mode, len(geoms), "boxes")
  path = ROOT / "tests" / "golden" / "terrain_reference.npz"
  np.savez_compressed(path, **out)
  print("wrote", path)


if __name__ == "__main__":
  main()
