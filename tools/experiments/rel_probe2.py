"""GPU box, reference staged: the reference's jit-scripted quat_mul / yaw_quat / quat_apply (after the profiling executor's warm-up: the
NNC-fused kernels a long run executes) against tools/experiments/rel_probe2.hip's renderings, stage by stage on independent inputs.
  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o gpurun_aux/rel_probe2.so tools/experiments/rel_probe2.hip; python tools/experiments/rel_probe2.py"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import reference_env  # noqa: E402

reference_env.install_stubs(reference_env.locate_reference())
from mjlab.third_party.isaaclab.isaaclab.utils.math import quat_apply, quat_mul, yaw_quat  # noqa: E402

lib = ctypes.CDLL(str(ROOT / "gpurun_aux" / "rel_probe2.so"))
vp = ctypes.c_void_p
lib.probe.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp]
dev, n = "cuda:0", 57344
g = torch.Generator(device=dev)
g.manual_seed(3)
r = lambda *s: torch.randn(s, device=dev, generator=g)  # noqa: E731
uq = lambda *s: torch.nn.functional.normalize(r(*s, 4), dim=-1)  # noqa: E731
st = torch.cuda.current_stream().cuda_stream
for rep in range(5):
  a, b, v = uq(n), uq(n), r(n, 3)
  yq = yaw_quat(a)
  want = {"quat_mul": quat_mul(a, b), "yaw_quat": yq, "quat_apply": quat_apply(a, v), "quat_mul (first operand a yaw quaternion)": quat_mul(yq, b),
          "quat_apply (yaw quaternion)": quat_apply(yq, v)}
  torch.cuda.synchronize()
  if rep in (0, 4):
    print(f"repetition {rep} ({'first call: unfused' if rep == 0 else 'fused'}):")
    for name, w in want.items():
      which = 0 if name.startswith("quat_mul") else 1 if name == "yaw_quat" else 2
      q = yq if "yaw quaternion" in name else a
      res = []
      for mode in range(3 if which < 2 else 9):
        o = torch.zeros_like(w)
        rc = lib.probe(which, n, q.data_ptr(), (b if which == 0 else v).data_ptr(), o.data_ptr(), mode, st)
        assert rc == 0
        torch.cuda.synchronize()
        res.append(int((o != w).sum()))
      print(f"  {name:45s} differing elements of {w.numel()} by mode: {res}")

    # enumeration of the contraction sites (rel_probe2.hip: k_quat_mul_enum bits: 1 qq, 2 outer products, 4 inner differences, 8 * {0, 1, 2} inner sum of xx, 24 outer sum of xx)
    lib.probe_enum.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp]
    for name, which, q, w, nv in (("quat_mul", 0, a, want["quat_mul"], 48), ("quat_mul (yaw first)", 0, yq, want["quat_mul (first operand a yaw quaternion)"], 48), ("yaw_quat", 1, a, want["yaw_quat"], 9)):
      rows = []
      for variant in list(range(nv)) + ([99] if which == 0 else []):
        o = torch.zeros_like(w)
        assert lib.probe_enum(which, n, q.data_ptr(), b.data_ptr(), o.data_ptr(), variant, st) == 0
        torch.cuda.synchronize()
        rows.append((int((o != w).sum()), variant, [int((o[:, k] != w[:, k]).sum()) for k in range(4)]))
      rows.sort()
      print(f"  {name}: best variants (differing, variant, per component):", rows[:4])
