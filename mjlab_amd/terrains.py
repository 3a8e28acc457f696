"""Procedural box terrains and environment origins (host side, numpy only).

Restates what the rough-terrain tasks of the reference put under the ``terrain`` body
(reference src/mjlab/terrains/terrain_generator.py:62-250, primitive_terrains.py:52-639,
utils.py:11-108, config.py:7-57) and how environments are placed on it
(terrain_importer.py:196-240).  Every terrain piece is an axis-aligned static box; the
physics step collides robot geoms against them through a uniform-grid broadphase
(``mjcf._compile`` builds the grid, ``k_collision`` walks it).

A sub-terrain is described by ``(boxes, origin)``: ``boxes`` is an ``(n, 6)`` array of
``[centre xyz, half-size xyz]`` rows in the sub-terrain's own frame (corner at 0, 0) and
``origin`` the spawn point.  Box order and random-number consumption follow the reference so
that a seeded generator yields the same geoms (``terrain_0 .. terrain_{n-1}``) as upstream;
tests/golden/terrain_reference.npz pins that against the reference's own code.
Heightfield sub-terrains are disabled upstream (config.py:28-55) and are not provided.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .mjcf import GEOM_BOX, Spec, SpecBody


def _box(cx, cy, cz, lx, ly, lz) -> list[float]:
  """Centre + full edge lengths -> centre + half sizes."""
  return [cx, cy, cz, lx / 2.0, ly / 2.0, lz / 2.0]


def border_boxes(size, inner_size, height, position) -> list[list[float]]:
  """Hollow rectangle out of four boxes: +y, -y, -x, +x strips (utils.py:43-108)."""
  tx = (size[0] - inner_size[0]) / 2.0
  ty = (size[1] - inner_size[1]) / 2.0
  px, py, pz = position
  dy = inner_size[1] / 2.0 + ty / 2.0
  dx = inner_size[0] / 2.0 + tx / 2.0
  return [
    _box(px, py + dy, pz, size[0], ty, height),
    _box(px, py - dy, pz, size[0], ty, height),
    _box(px - dx, py, pz, tx, inner_size[1], height),
    _box(px + dx, py, pz, tx, inner_size[1], height),
  ]


@dataclass
class SubTerrainCfg:
  proportion: float = 1.0
  size: tuple[float, float] = (10.0, 10.0)

  def function(self, difficulty: float, rng: np.random.Generator) -> tuple[np.ndarray, np.ndarray]:
    raise NotImplementedError


@dataclass(kw_only=True)
class BoxFlatTerrainCfg(SubTerrainCfg):
  """One 1 m thick slab whose top is z = 0 (primitive_terrains.py:52-63, utils.py:11-35)."""

  def function(self, difficulty, rng):
    sx, sy = self.size
    return np.array([_box(sx / 2.0, sy / 2.0, -0.5, sx, sy, 1.0)]), np.array([sx / 2.0, sy / 2.0, 0.0])


@dataclass(kw_only=True)
class BoxPyramidStairsTerrainCfg(SubTerrainCfg):
  """Concentric rings of steps rising to a central platform (primitive_terrains.py:66-222)."""

  border_width: float = 0.0
  step_height_range: tuple[float, float] = (0.0, 0.1)
  step_width: float = 0.3
  platform_width: float = 1.0
  holes: bool = False

  def _layout(self, difficulty):
    sx, sy = self.size
    h = self.step_height_range[0] + difficulty * (self.step_height_range[1] - self.step_height_range[0])
    nx = (sx - 2 * self.border_width - self.platform_width) // (2 * self.step_width) + 1
    ny = (sy - 2 * self.border_width - self.platform_width) // (2 * self.step_width) + 1
    return h, int(min(nx, ny)), (sx - 2 * self.border_width, sy - 2 * self.border_width)

  def _ring(self, k, inner, z, height):
    """The four strips of ring k: +y, -y, +x, -x."""
    sx, sy = self.size
    cx, cy = sx / 2.0, sy / 2.0
    w = self.step_width
    if self.holes:
      bx, by = self.platform_width, self.platform_width
      side_y = by
    else:
      bx, by = inner[0] - 2 * k * w, inner[1] - 2 * k * w
      side_y = by - 2 * w
    off = (k + 0.5) * w
    return [
      _box(cx, cy + inner[1] / 2.0 - off, z, bx, w, height),
      _box(cx, cy - inner[1] / 2.0 + off, z, bx, w, height),
      _box(cx + inner[0] / 2.0 - off, cy, z, w, side_y, height),
      _box(cx - inner[0] / 2.0 + off, cy, z, w, side_y, height),
    ]

  def function(self, difficulty, rng):
    sx, sy = self.size
    h, nsteps, inner = self._layout(difficulty)
    boxes: list[list[float]] = []
    if self.border_width > 0.0 and not self.holes:
      boxes += border_boxes(self.size, inner, h, (sx / 2.0, sy / 2.0, -h / 2.0))
    for k in range(nsteps):
      boxes += self._ring(k, inner, k * h / 2.0, (k + 2) * h)
    w = self.step_width
    boxes.append(_box(sx / 2.0, sy / 2.0, nsteps * h / 2.0, inner[0] - 2 * nsteps * w, inner[1] - 2 * nsteps * w, (nsteps + 2) * h))
    return np.array(boxes), np.array([sx / 2.0, sy / 2.0, (nsteps + 1) * h])


@dataclass(kw_only=True)
class BoxInvertedPyramidStairsTerrainCfg(BoxPyramidStairsTerrainCfg):
  """Rings descending to a central pit (primitive_terrains.py:225-377)."""

  def function(self, difficulty, rng):
    sx, sy = self.size
    h, nsteps, inner = self._layout(difficulty)
    total = (nsteps + 1) * h
    boxes: list[list[float]] = []
    if self.border_width > 0.0 and not self.holes:
      boxes += border_boxes(self.size, inner, h, (sx / 2.0, sy / 2.0, -0.5 * h))
    for k in range(nsteps):
      boxes += self._ring(k, inner, -total / 2.0 - (k + 1) * h / 2.0, total - (k + 1) * h)
    w = self.step_width
    boxes.append(_box(sx / 2.0, sy / 2.0, -total - h / 2.0, inner[0] - 2 * nsteps * w, inner[1] - 2 * nsteps * w, h))
    return np.array(boxes), np.array([sx / 2.0, sy / 2.0, -(nsteps + 1) * h])


@dataclass(kw_only=True)
class BoxRandomGridTerrainCfg(SubTerrainCfg):
  """Grid of columns with random heights around a central platform
  (primitive_terrains.py:379-639): four border strips, then the columns -- individually, or
  greedily merged into rectangles of equal quantised height -- then the platform."""

  grid_width: float
  grid_height_range: tuple[float, float]
  platform_width: float = 1.0
  holes: bool = False
  merge_similar_heights: bool = False
  height_merge_threshold: float = 0.05
  max_merge_distance: int = 3

  def function(self, difficulty, rng):
    sx, sy = self.size
    if sx != sy:
      raise ValueError(f"The terrain must be square. Received size: {self.size}.")
    gh = self.grid_height_range[0] + difficulty * (self.grid_height_range[1] - self.grid_height_range[0])
    nx, ny = int(sx / self.grid_width), int(sy / self.grid_width)
    depth = 1.0  # the columns reach 1 m below z = 0
    border = sx - min(nx, ny) * self.grid_width
    if border <= 0:
      raise RuntimeError("Border width must be greater than 0! Adjust the parameter 'self.grid_width'.")
    bt = border / 2
    zc = -depth / 2
    inner_y = sy - 2 * bt
    boxes = [
      _box(sx / 2, sy - bt / 2, zc, sx, bt, depth),
      _box(sx / 2, bt / 2, zc, sx, bt, depth),
      _box(bt / 2, sx / 2, zc, bt, inner_y, depth),
      _box(sx - bt / 2, sx / 2, zc, bt, inner_y, depth),
    ]
    hmap = rng.uniform(-gh, gh, (nx, ny))
    gw = self.grid_width
    if self.merge_similar_heights and not self.holes:
      q = np.round(hmap / self.height_merge_threshold) * self.height_merge_threshold
      done = np.zeros((nx, ny), dtype=bool)
      for i in range(nx):
        for j in range(ny):
          if done[i, j]:
            continue
          h = q[i, j]
          i1 = i + 1  # grow along x first, then along y over the whole x run
          while i1 < min(i + self.max_merge_distance, nx) and not done[i1, j] and abs(q[i1, j] - h) < 1e-6:
            i1 += 1
          j1 = j + 1
          while j1 < min(j + self.max_merge_distance, ny) and not done[i:i1, j1].any() and (np.abs(q[i:i1, j1] - h) <= 1e-6).all():
            j1 += 1
          done[i:i1, j:j1] = True
          boxes.append(_box(bt + (i + (i1 - i) / 2) * gw, bt + (j + (j1 - j) / 2) * gw, zc + h / 2, (i1 - i) * gw, (j1 - j) * gw, depth + h))
    else:
      lo, hi = sx / 2 - self.platform_width / 2, sx / 2 + self.platform_width / 2
      for i in range(nx):
        cx = bt + (i + 0.5) * gw
        for j in range(ny):
          cy = bt + (j + 0.5) * gw
          if self.holes and not (lo <= cx <= hi or lo <= cy <= hi):
            continue  # holes: only the cross through the platform is filled
          boxes.append(_box(cx, cy, zc + hmap[i, j] / 2, gw, gw, depth + hmap[i, j]))
    boxes.append(_box(sx / 2, sy / 2, zc + gh / 2, self.platform_width, self.platform_width, depth + gh))
    return np.array(boxes), np.array([sx / 2, sy / 2, gh])


@dataclass(kw_only=True)
class TerrainGeneratorCfg:
  size: tuple[float, float]
  sub_terrains: dict[str, SubTerrainCfg]
  seed: int | None = None
  curriculum: bool = False
  border_width: float = 0.0
  border_height: float = 1.0
  num_rows: int = 1
  num_cols: int = 1
  difficulty_range: tuple[float, float] = (0.0, 1.0)


def rough_terrains_cfg(seed: int = 0, curriculum: bool = True, num_rows: int = 10, num_cols: int = 20) -> TerrainGeneratorCfg:
  """``ROUGH_TERRAINS_CFG`` of the reference (terrains/config.py:7-57) as the velocity tasks
  use it (curriculum switched on, velocity_env_cfg.py:275-278)."""
  stairs = dict(proportion=0.3, step_height_range=(0.0, 0.1), step_width=0.3, platform_width=3.0, border_width=1.0)
  return TerrainGeneratorCfg(
    size=(8.0, 8.0),
    border_width=20.0,
    num_rows=num_rows,
    num_cols=num_cols,
    seed=seed,
    curriculum=curriculum,
    sub_terrains={
      "flat": BoxFlatTerrainCfg(proportion=0.4),
      "pyramid_stairs": BoxPyramidStairsTerrainCfg(**stairs),
      "pyramid_stairs_inv": BoxInvertedPyramidStairsTerrainCfg(**stairs),
    },
  )


@dataclass
class Terrain:
  """Generated terrain: world-frame boxes and the spawn origin of every sub-terrain."""

  boxes: np.ndarray  # (n, 6) centre + half size, world frame, axis aligned
  origins: np.ndarray  # (num_rows, num_cols, 3)
  sub_index: np.ndarray = field(default_factory=lambda: np.zeros((0, 0), np.int32))  # which sub-terrain type


class TerrainGenerator:
  """Grid of sub-terrains centred on the world origin plus an outer border
  (terrain_generator.py:62-250).  A seed is required: the product must be reproducible."""

  def __init__(self, cfg: TerrainGeneratorCfg) -> None:
    if len(cfg.sub_terrains) == 0:
      raise ValueError("At least one sub_terrain must be specified.")
    if cfg.seed is None:
      raise ValueError("TerrainGeneratorCfg.seed must be set")
    self.cfg = cfg
    for sub in cfg.sub_terrains.values():
      sub.size = cfg.size
    self.rng = np.random.default_rng(cfg.seed)

  def _corner(self, row: int, col: int) -> np.ndarray:
    c = self.cfg
    return np.array([(row - c.num_rows * 0.5) * c.size[0], (col - c.num_cols * 0.5) * c.size[1], 0.0])

  def generate(self) -> Terrain:
    c = self.cfg
    subs = list(c.sub_terrains.values())
    prop = np.array([s.proportion for s in subs], dtype=np.float64)
    prop /= prop.sum()
    origins = np.zeros((c.num_rows, c.num_cols, 3))
    kind = np.zeros((c.num_rows, c.num_cols), np.int32)
    out: list[np.ndarray] = []

    def place(row, col, sub_i, difficulty):
      boxes, origin = subs[sub_i].function(difficulty, self.rng)
      corner = self._corner(row, col)
      boxes = boxes.copy()
      boxes[:, :3] += corner
      out.append(boxes)
      origins[row, col] = origin + corner
      kind[row, col] = sub_i

    if c.curriculum:
      # column -> sub-terrain type by cumulative proportion, row -> difficulty
      cum = np.cumsum(prop)
      col_type = [int(np.min(np.where(col / c.num_cols + 0.001 < cum)[0])) for col in range(c.num_cols)]
      lo, hi = c.difficulty_range
      for col in range(c.num_cols):
        for row in range(c.num_rows):
          difficulty = lo + (hi - lo) * (row + self.rng.uniform()) / c.num_rows
          place(row, col, col_type[col], difficulty)
    else:
      for index in range(c.num_rows * c.num_cols):
        row, col = divmod(index, c.num_cols)
        sub_i = int(self.rng.choice(len(prop), p=prop))
        difficulty = self.rng.uniform(*c.difficulty_range)
        place(row, col, sub_i, difficulty)

    inner = (c.num_rows * c.size[0], c.num_cols * c.size[1])
    outer = (inner[0] + 2 * c.border_width, inner[1] + 2 * c.border_width)
    if c.border_width > 0.0:
      out.append(np.array(border_boxes(outer, inner, abs(c.border_height), (0.0, 0.0, -c.border_height / 2.0))))
    return Terrain(np.concatenate(out, axis=0), origins, kind)

  def compile(self, spec: Spec) -> Terrain:
    """Add a ``terrain`` body holding geoms ``terrain_0 ..`` to ``spec``."""
    terrain = self.generate()
    body = spec.add_body("terrain")
    add_boxes(spec, body, terrain.boxes)
    return terrain


def add_boxes(spec: Spec, body: SpecBody, boxes: np.ndarray, prefix: str = "terrain_") -> None:
  for i, b in enumerate(boxes):
    if np.any(b[3:] <= 0.0):
      # zero-height steps (difficulty 0) are legal upstream; keep the geom count, make it inert
      spec.add_geom(body, f"{prefix}{i}", GEOM_BOX, np.maximum(b[3:], 1e-6), pos=b[:3], contype=0, conaffinity=0)
    else:
      spec.add_geom(body, f"{prefix}{i}", GEOM_BOX, b[3:], pos=b[:3])


def env_origins_curriculum(num_envs: int, origins: np.ndarray, max_init_level: int | None, rng: np.random.Generator):
  """Environment -> (level, type) assignment and origins (terrain_importer.py:211-229):
  types are spread evenly over the columns, levels drawn uniformly from ``[0, max_init_level]``."""
  num_rows, num_cols = origins.shape[:2]
  top = num_rows - 1 if max_init_level is None else min(max_init_level, num_rows - 1)
  levels = rng.integers(0, top + 1, size=num_envs)
  # the reference floors a float32 division (torch.div(arange, num_envs / num_cols, rounding_mode="floor")):
  # with 4096 envs over 20 columns that differs from exact arithmetic at the column boundaries
  types = np.floor_divide(np.arange(num_envs, dtype=np.float32), np.float32(num_envs / num_cols)).astype(np.int64)
  return origins[levels, types].astype(np.float64), levels, types


def env_origins_grid(num_envs: int, env_spacing: float) -> np.ndarray:
  """Square-ish grid of origins for plane terrains (terrain_importer.py:231-247)."""
  num_rows = np.ceil(num_envs / int(np.sqrt(num_envs)))
  num_cols = np.ceil(num_envs / num_rows)
  ii, jj = np.meshgrid(np.arange(num_rows), np.arange(num_cols), indexing="ij")
  out = np.zeros((num_envs, 3))
  out[:, 0] = -(ii.flatten()[:num_envs] - (num_rows - 1) / 2) * env_spacing
  out[:, 1] = (jj.flatten()[:num_envs] - (num_cols - 1) / 2) * env_spacing
  return out


@dataclass
class TerrainImporterCfg:
  """Reference terrains/terrain_importer.py:36-54."""

  terrain_type: str = "plane"  # "generator" | "plane"
  terrain_generator: TerrainGeneratorCfg | None = None
  env_spacing: float | None = 2.0
  max_init_terrain_level: int | None = None
  num_envs: int = 1


class TerrainImporter:
  """Terrain geometry + environment placement (reference terrains/terrain_importer.py:57-240): adds
  the terrain to a scene ``Spec``, computes ``env_origins`` (sub-terrain origins by level / type for
  generated terrains, a square grid for the plane) and moves environments between difficulty levels
  (``update_env_origins``, driven by the reference's ``terrain_levels_vel`` curriculum term).
  Origins are torch tensors on ``device`` like upstream; the random draws use a seeded generator."""

  def __init__(self, cfg: TerrainImporterCfg, device: str, spec: Spec | None = None, seed: int = 0) -> None:
    import torch

    self.cfg, self.device = cfg, device
    self.spec = spec if spec is not None else Spec()
    self._gen = torch.Generator(device="cpu")
    self._gen.manual_seed(seed)
    self.terrain_origins = None
    if cfg.terrain_type == "generator":
      if cfg.terrain_generator is None:
        raise ValueError("terrain_generator must be specified for terrain_type 'generator'")
      self.terrain = TerrainGenerator(cfg.terrain_generator).compile(self.spec)
      self.terrain_origins = torch.tensor(self.terrain.origins, dtype=torch.float, device=device)
      num_rows, num_cols = self.terrain_origins.shape[:2]
      top = num_rows - 1 if cfg.max_init_terrain_level is None else min(cfg.max_init_terrain_level, num_rows - 1)
      self.max_terrain_level = num_rows
      self.terrain_levels = torch.randint(0, top + 1, (cfg.num_envs,), generator=self._gen).to(device)
      self.terrain_types = torch.div(torch.arange(cfg.num_envs, device=device), cfg.num_envs / num_cols, rounding_mode="floor").to(torch.long)
      self.env_origins = self.terrain_origins[self.terrain_levels, self.terrain_types].clone()
    elif cfg.terrain_type == "plane":
      from .mjcf import GEOM_PLANE

      self.spec.add_geom(self.spec.add_body("terrain"), "terrain", GEOM_PLANE, (0, 0, 0.01))
      if cfg.env_spacing is None:
        raise ValueError("Environment spacing must be specified for configuring grid-like origins.")
      self.env_origins = torch.tensor(env_origins_grid(cfg.num_envs, cfg.env_spacing), dtype=torch.float, device=device)
    else:
      raise ValueError(f"Unknown terrain type: {cfg.terrain_type}")

  def update_env_origins(self, env_ids, move_up, move_down) -> None:
    """Promote / demote environments; one that outgrows the last level restarts at a random one."""
    import torch

    if self.terrain_origins is None:
      return
    lv = self.terrain_levels[env_ids] + 1 * move_up - 1 * move_down
    rnd = torch.randint(0, self.max_terrain_level, (len(lv),), generator=self._gen).to(lv.device)
    self.terrain_levels[env_ids] = torch.where(lv >= self.max_terrain_level, rnd, torch.clip(lv, 0))
    self.env_origins[env_ids] = self.terrain_origins[self.terrain_levels[env_ids], self.terrain_types[env_ids]]
