"""Time upstream MuJoCo's CPU `mj_step` on the benchmark workload -- the baseline the north_star
asks for.  It needs the `mujoco` wheel and a checkout of mjlab with its dependencies (to build the
scene exactly like the task does); neither exists in the build container, where this script exits
with a message and `bench.py` reports the C restatement instead (`cpu_baseline.kind = "port"`).

  python tools/cpu_mujoco.py --reference /path/to/mjlab [--task Mjlab-Velocity-Flat-Unitree-G1]
"""

from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import sys
import time

import numpy as np


def main() -> None:
  ap = argparse.ArgumentParser()
  ap.add_argument("--reference", required=True)
  ap.add_argument("--task", default="Mjlab-Velocity-Flat-Unitree-G1")
  ap.add_argument("--envs", type=int, default=4096)
  ap.add_argument("--env-steps", type=int, default=20)
  ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
  args = ap.parse_args()
  try:
    import mujoco
  except ImportError as e:
    raise SystemExit(f"upstream mujoco is not importable here ({e}); bench.py reports the C restatement instead")
  sys.path.insert(0, os.path.join(args.reference, "src"))
  from mjlab.scene import Scene  # type: ignore
  from mjlab.tasks.registry import load_cfg_from_registry  # type: ignore

  cfg = load_cfg_from_registry(args.task, "env_cfg_entry_point")
  cfg.scene.num_envs = 1
  scene = Scene(cfg.scene, device="cpu")
  cfg.sim.mujoco.edit_spec(scene.spec)
  model = scene.compile()
  key = model.key("init_state").id
  rng = np.random.default_rng(42)
  datas = [mujoco.MjData(model) for _ in range(args.envs)]
  for d in datas:
    mujoco.mj_resetDataKeyframe(model, d, key)
  default = model.key_qpos[key][7:]
  chunks = np.array_split(np.arange(args.envs), args.threads)

  def work(idx, actions):
    for i in idx:
      d = datas[i]
      d.ctrl[:] = default + 0.25 * actions[i]
      for _ in range(cfg.decimation):
        mujoco.mj_step(model, d)
      mujoco.mj_forward(model, d)

  t0 = time.perf_counter()
  with cf.ThreadPoolExecutor(args.threads) as pool:  # mj_step releases the GIL
    for _ in range(args.env_steps):
      act = rng.uniform(-1, 1, (args.envs, model.nu))
      list(pool.map(lambda idx: work(idx, act), chunks))
  dt = time.perf_counter() - t0
  print(f"upstream mj_step: {args.envs * args.env_steps / dt:.0f} env-steps/s on {args.threads} threads "
        f"({args.task}, {args.envs} envs, {cfg.decimation} substeps + 1 forward per env-step)")


if __name__ == "__main__":
  main()
