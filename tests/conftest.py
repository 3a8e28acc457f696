import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
  sys.path.insert(0, str(ROOT))


# The parity suites compare with the oracle's default search, MuJoCo's exact iterative one (mj_solPrimal); mujoco_warp's parallel
# grid search -- what SimulationCfg.ls_parallel=True, the reference's default, selects -- has its own tests, which lift this override.
import os  # noqa: E402

os.environ.setdefault("MJLAB_LS_PARALLEL", "0")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def reference_root():
  p = Path("/root/reference")
  if not p.exists():
    pytest.skip("reference checkout not present")
  return p
