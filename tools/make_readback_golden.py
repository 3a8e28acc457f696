"""Golden vectors for the fused entity read-back, computed by THE REFERENCE'S OWN torch code
(src/mjlab/entity/data.py compute_velocity_from_cvel and the quaternion helpers of
third_party/isaaclab/isaaclab/utils/math.py), imported here with the unavailable third-party
packages stubbed (see tools/make_reference_pins.py).  Inputs are random mjData-shaped arrays for
the G1 entity; tests/test_gpu_readback.py writes them into sim.data and compares
mjlab_entity_readback's outputs.  Run in the build container:  python tools/make_readback_golden.py
"""

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

from make_reference_pins import REF, _StubFinder  # noqa: E402

from mjlab_amd import robots  # noqa: E402


def main() -> None:
  sys.meta_path.insert(0, _StubFinder())
  sys.path.insert(0, str(REF / "src"))
  from mjlab.entity.data import compute_velocity_from_cvel
  from mjlab.third_party.isaaclab.isaaclab.utils.math import quat_apply, quat_apply_inverse, quat_mul

  model = robots.load_model("g1_velocity_flat")
  root = int(np.nonzero(model.body_parentid == 0)[0][-1])
  ids = np.arange(root, root + int(model.body_subtreenum[root]))
  n, nb = 16, model.nbody
  g = torch.Generator().manual_seed(0)
  xpos = torch.randn((n, nb, 3), generator=g)
  xipos = xpos + 0.05 * torch.randn((n, nb, 3), generator=g)
  xquat = torch.nn.functional.normalize(torch.randn((n, nb, 4), generator=g), dim=-1)
  cvel = torch.randn((n, nb, 6), generator=g)
  sub = torch.randn((n, nb, 3), generator=g)
  iquat = torch.tensor(model.body_iquat, dtype=torch.float32)
  idt = torch.from_numpy(ids).long()
  pos, quat, ipos, cv = xpos[:, idt], xquat[:, idt], xipos[:, idt], cvel[:, idt]
  sc = sub[:, root].unsqueeze(1)
  out = {
    "in_xpos": xpos, "in_xipos": xipos, "in_xquat": xquat, "in_cvel": cvel, "in_subtree_com": sub,
    "body_link_vel_w": compute_velocity_from_cvel(pos, sc, cv),
    "body_com_vel_w": compute_velocity_from_cvel(ipos, sc, cv),
    "body_com_quat_w": quat_mul(quat, iquat[idt].unsqueeze(0).expand(n, -1, -1)),
  }  # fmt: skip
  rq = quat[:, 0]
  grav = torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1)
  fwd = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1)
  out["projected_gravity_b"] = quat_apply_inverse(rq, grav)
  f = quat_apply(rq, fwd)
  out["heading_w"] = torch.atan2(f[:, 1], f[:, 0])
  lv = out["body_link_vel_w"][:, 0]
  cvw = out["body_com_vel_w"][:, 0]
  out["root_link_lin_vel_b"] = quat_apply_inverse(rq, lv[:, :3])
  out["root_link_ang_vel_b"] = quat_apply_inverse(rq, lv[:, 3:])
  out["root_com_lin_vel_b"] = quat_apply_inverse(rq, cvw[:, :3])
  out["root_com_ang_vel_b"] = quat_apply_inverse(rq, cvw[:, 3:])
  dst = ROOT / "tests" / "golden" / "readback_reference.npz"
  np.savez_compressed(dst, **{k: v.numpy() for k, v in out.items()})
  print("wrote", dst)


if __name__ == "__main__":
  main()
