"""GPU box, reference staged: MotionCommand._update_command's relative body poses by the reference's own (jit-scripted, NNC-fused) helpers
against tools/experiments/rel_probe.hip under every combination of its contraction switches -- which variant, if any, reproduces the
reference's bits?  Prints the number of differing elements per variant (quaternion, position) after the jit's profiling runs.
  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o gpurun_aux/rel_probe.so tools/experiments/rel_probe.hip
  python tools/experiments/rel_probe.py"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import reference_env  # noqa: E402

reference_env.install_stubs(reference_env.locate_reference())
from mjlab.third_party.isaaclab.isaaclab.utils.math import quat_apply, quat_inv, quat_mul, yaw_quat  # noqa: E402

lib = ctypes.CDLL(str(ROOT / "gpurun_aux" / "rel_probe.so"))
vp = ctypes.c_void_p
lib.rel_probe.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int, vp, vp]
dev, n, nb = "cuda:0", 4096, 14
g = torch.Generator(device=dev)
g.manual_seed(3)
r = lambda *s: torch.randn(s, device=dev, generator=g)  # noqa: E731
uq = lambda *s: torch.nn.functional.normalize(r(*s, 4), dim=-1)  # noqa: E731


def reference(apos, aquat, rpos, rquat, bpos, bquat):
  ap, aq = apos[:, None, :].repeat(1, nb, 1), aquat[:, None, :].repeat(1, nb, 1)
  rp, rq = rpos[:, None, :].repeat(1, nb, 1), rquat[:, None, :].repeat(1, nb, 1)
  delta_pos = rp
  delta_pos[..., 2] = ap[..., 2]
  inv = quat_inv(aq)
  d = quat_mul(rq, inv)
  delta_ori = yaw_quat(d)
  return delta_pos + quat_apply(delta_ori, bpos - ap), quat_mul(delta_ori, bquat), torch.cat([inv, d, delta_ori], dim=-1)


for rep in range(4):  # (the profiling executor fuses after its profiling runs: the last repetition is what a long run computes)
  apos, rpos, bpos = r(n, 3) * 2, r(n, 3) * 2, r(n, nb, 3) * 2
  aquat, rquat, bquat = uq(n), uq(n), uq(n, nb)
  want_p, want_q, want_dbg = reference(apos, aquat, rpos, rquat, bpos, bquat)
  torch.cuda.synchronize()
  if rep in (0, 3):
    print(f"repetition {rep}:")
    best = None
    for variant in range(64):
      if (variant & 8) and (variant & 32):
        continue
      out_p, out_q, dbg = torch.zeros_like(want_p), torch.zeros_like(want_q), torch.zeros_like(want_dbg)
      rc = lib.rel_probe(n, nb, apos.data_ptr(), aquat.data_ptr(), rpos.data_ptr(), rquat.data_ptr(), bpos.contiguous().data_ptr(), bquat.contiguous().data_ptr(),
                         out_p.data_ptr(), out_q.data_ptr(), variant, dbg.data_ptr(), torch.cuda.current_stream().cuda_stream)
      assert rc == 0
      torch.cuda.synchronize()
      dq, dp = int((out_q != want_q).sum()), int((out_p != want_p).sum())
      if best is None or dq + dp < best[0]:
        best = (dq + dp, variant)
      if dq == 0 or dp == 0 or variant in (0, 1, 16, 17, 63):
        parts = [int((dbg[..., a:a + 4] != want_dbg[..., a:a + 4]).sum()) for a in (0, 4, 8)]
        comp = [int((out_q[..., k] != want_q[..., k]).sum()) for k in range(4)]
        print(f"  variant {variant:2d} ({variant:06b}): {dq:7d} of {want_q.numel()} quaternion elements differ (w x y z: {comp}), {dp:7d} of {want_p.numel()} position elements; "
              f"inv / d / delta_ori differing: {parts}")
    print("  best:", best)
