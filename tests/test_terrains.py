"""Box-terrain generator vs the boxes recorded from the reference's own generator code
(tests/golden/terrain_reference.npz, made by tools/make_terrain_golden.py)."""

from pathlib import Path

import numpy as np
import pytest

from mjlab_amd import terrains

GOLD = Path(__file__).parent / "golden" / "terrain_reference.npz"


@pytest.fixture(scope="module")
def gold():
  with np.load(GOLD) as z:
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("case", ["curriculum_10x20_seed0", "random_3x5_seed7", "curriculum_2x3_seed1"])
def test_boxes_and_origins_match_reference_generator(gold, case):
  seed, curriculum, rows, cols = (int(v) for v in gold[case + "/cfg"])
  cfg = terrains.rough_terrains_cfg(seed=seed, curriculum=bool(curriculum), num_rows=rows, num_cols=cols)
  t = terrains.TerrainGenerator(cfg).generate()
  ref = gold[case + "/boxes"]
  assert t.boxes.shape == ref.shape
  np.testing.assert_allclose(t.boxes, ref, rtol=0, atol=1e-12)
  np.testing.assert_allclose(t.origins, gold[case + "/origins"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("mode,kw", [("plain", {}), ("merged", {"merge_similar_heights": True}), ("holes", {"holes": True})])
def test_random_grid_matches_reference_generator(gold, mode, kw):
  sub = terrains.BoxRandomGridTerrainCfg(grid_width=0.45, grid_height_range=(0.05, 0.2), platform_width=2.0, **kw)
  cfg = terrains.TerrainGeneratorCfg(size=(8.0, 8.0), seed=11, num_rows=2, num_cols=2, border_width=1.0, sub_terrains={"grid": sub})
  t = terrains.TerrainGenerator(cfg).generate()
  ref = gold[f"grid_{mode}/boxes"]
  assert t.boxes.shape == ref.shape
  np.testing.assert_allclose(t.boxes, ref, rtol=0, atol=1e-12)
  np.testing.assert_allclose(t.origins, gold[f"grid_{mode}/origins"], rtol=0, atol=1e-12)


def test_rough_cfg_counts():
  t = terrains.TerrainGenerator(terrains.rough_terrains_cfg(seed=0)).generate()
  # 8 flat columns x 10 rows (1 slab each) + 12 stair columns x 10 rows x (4 border + 6 rings x 4 + 1) + 4 outer border
  assert len(t.boxes) == 80 + 120 * 29 + 4
  assert (t.boxes[:, 3:] > 0).all()
  # the outer border encloses the grid: 80 x 160 m of sub-terrains + 20 m on every side
  lo = (t.boxes[:, :3] - t.boxes[:, 3:]).min(axis=0)
  hi = (t.boxes[:, :3] + t.boxes[:, 3:]).max(axis=0)
  np.testing.assert_allclose(lo[:2], [-60.0, -100.0])
  np.testing.assert_allclose(hi[:2], [60.0, 100.0])


def test_spawn_origin_sits_on_top_of_a_box():
  t = terrains.TerrainGenerator(terrains.rough_terrains_cfg(seed=3, num_rows=4, num_cols=6)).generate()
  for o in t.origins.reshape(-1, 3):
    inside = np.all(np.abs(t.boxes[:, :2] - o[:2]) <= t.boxes[:, 3:5], axis=1)
    top = (t.boxes[inside, 2] + t.boxes[inside, 5]).max()
    assert abs(top - o[2]) < 1e-9


def test_env_origins():
  rng = np.random.default_rng(0)
  origins = np.arange(10 * 20 * 3, dtype=np.float64).reshape(10, 20, 3)
  eo, levels, types = terrains.env_origins_curriculum(4096, origins, 5, rng)
  assert eo.shape == (4096, 3) and levels.max() <= 5 and levels.min() == 0
  # types are spread evenly over the columns in env order (terrain_importer.py:222-226)
  import torch

  assert np.array_equal(types, torch.div(torch.arange(4096), 4096 / 20, rounding_mode="floor").long().numpy())
  np.testing.assert_array_equal(eo, origins[levels, types])
  g = terrains.env_origins_grid(16, 2.0)
  assert g.shape == (16, 3) and np.allclose(g.mean(axis=0), 0) and np.isclose(np.ptp(g[:, 0]), 6.0)


def test_seed_required():
  cfg = terrains.rough_terrains_cfg(seed=0)
  cfg.seed = None
  with pytest.raises(ValueError):
    terrains.TerrainGenerator(cfg)


def test_terrain_importer_origins_and_level_updates():
  import torch

  cfg = terrains.TerrainImporterCfg(terrain_type="generator", terrain_generator=terrains.rough_terrains_cfg(seed=0), max_init_terrain_level=5, num_envs=64)
  imp = terrains.TerrainImporter(cfg, device="cpu", seed=1)
  assert imp.env_origins.shape == (64, 3) and int(imp.terrain_levels.max()) <= 5
  assert torch.equal(imp.terrain_types, torch.div(torch.arange(64), 64 / 20, rounding_mode="floor").long())
  _, _, types_np = terrains.env_origins_curriculum(64, np.zeros((10, 20, 3)), 5, np.random.default_rng(0))
  assert np.array_equal(types_np, imp.terrain_types.numpy())  # the numpy helper floors like the reference's float32 division
  assert torch.equal(imp.env_origins, imp.terrain_origins[imp.terrain_levels, imp.terrain_types])
  assert len(imp.spec.body("terrain").geoms) == 3564
  # promote env 0..3, demote 4..7 (reference terrain_importer.py:196-209)
  ids = torch.arange(8)
  before = imp.terrain_levels.clone()
  up = torch.tensor([1, 1, 1, 1, 0, 0, 0, 0], dtype=torch.bool)
  imp.update_env_origins(ids, up, ~up)
  assert torch.equal(imp.terrain_levels[:4], before[:4] + 1)
  assert torch.equal(imp.terrain_levels[4:8], torch.clip(before[4:8] - 1, 0))
  assert torch.equal(imp.env_origins[:8], imp.terrain_origins[imp.terrain_levels[:8], imp.terrain_types[:8]])
  # outgrowing the last level restarts at a random valid level
  imp.terrain_levels[:] = 9
  imp.update_env_origins(ids, torch.ones(8, dtype=torch.bool), torch.zeros(8, dtype=torch.bool))
  assert (imp.terrain_levels[:8] >= 0).all() and (imp.terrain_levels[:8] <= 9).all()
  # plane: grid origins
  pl = terrains.TerrainImporter(terrains.TerrainImporterCfg(terrain_type="plane", num_envs=9, env_spacing=2.0), device="cpu")
  assert pl.env_origins.shape == (9, 3) and pl.terrain_origins is None
  pl.update_env_origins(ids[:2], up[:2], ~up[:2])  # no-op without sub-terrains
