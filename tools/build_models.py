"""Compile the BASELINE.json scenes from the reference MJCF into mjlab_amd/assets/*.npz.

Run in a container that has the reference checkout (``/root/reference`` or
``$MJLAB_REFERENCE_ROOT``).  The GPU box has no reference tree: bench.py, smoke()
and the ``-m gpu`` tests load the committed ``.npz`` files instead.
"""

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from mjlab_amd import robots  # noqa: E402


def main() -> None:
  robots.ASSET_DIR.mkdir(exist_ok=True)
  for name in robots.SCENES:
    m = robots.compile_scene(name)
    out = robots.ASSET_DIR / f"{name}.npz"
    m.save(out)
    print(
      f"{name}: nq={m.nq} nv={m.nv} nu={m.nu} nbody={m.nbody} ngeom={m.ngeom} nsite={m.nsite} "
      f"npair={m.npair} nsensordata={m.nsensordata} meaninertia={m.meaninertia:.6g} -> {out}"
    )


if __name__ == "__main__":
  main()
