"""GPU box, reference staged: tools/experiments/rel_probe2.py's stage-wise comparison INSIDE the tracking environment -- the helpers called with
the tensors MotionCommand._update_command hands them, after the process has built and stepped the environment (the jit's history as a run has it).
  python tools/experiments/rel_probe_env.py"""
import ctypes
import sys
import tempfile
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import reference_env  # noqa: E402

from mjlab_amd import robots  # noqa: E402
from mjlab_amd.graphed_env import GraphedRlEnv  # noqa: E402
from mjlab_amd.rollout import write_motion_npz  # noqa: E402

path = str(Path(tempfile.mkdtemp()) / "motion.npz")
write_motion_npz(path, robots.load_model("g1_tracking_flat"), "cuda:0")


def edit(cfg):
  cfg.commands.motion.motion_file = path


n = 256
env = reference_env.make_env("Mjlab-Tracking-Flat-Unitree-G1", num_envs=n, device="cuda:0", cfg_edit=edit)
env.reset()
g = GraphedRlEnv(env, capture=False, fused_relative_poses=False)
for _ in range(4):
  g.step(torch.rand((n, 29), device="cuda:0") * 2 - 1)
from mjlab.third_party.isaaclab.isaaclab.utils.math import quat_apply, quat_inv, quat_mul, yaw_quat  # noqa: E402

lib = ctypes.CDLL(str(ROOT / "gpurun_aux" / "rel_probe2.so"))
vp = ctypes.c_void_p
lib.probe.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp]
lib.probe_enum.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp]
st = torch.cuda.current_stream().cuda_stream
term = env.command_manager.get_term("motion")
nb = len(term.cfg.body_names)
g._caching[0] = False
for rep in range(3):
  ap = term.anchor_pos_w[:, None, :].repeat(1, nb, 1)
  aq = term.anchor_quat_w[:, None, :].repeat(1, nb, 1)
  rp = term.robot_anchor_pos_w[:, None, :].repeat(1, nb, 1)
  rq = term.robot_anchor_quat_w[:, None, :].repeat(1, nb, 1)
  inv = quat_inv(aq)
  d = quat_mul(rq, inv)
  dori = yaw_quat(d)
  bq = term.body_quat_w
  oq = quat_mul(dori, bq)
  rel = term.body_pos_w - ap
  rot = quat_apply(dori, rel)
  torch.cuda.synchronize()
print("input layouts: aq", aq.stride(), "rq", rq.stride(), "inv", inv.stride(), "d", d.stride(), "dori", dori.stride(), "bq", bq.stride(), bq.is_contiguous(), "rel", rel.stride())
N = n * nb
s2 = (aq[..., 0] ** 2 + aq[..., 1] ** 2) + (aq[..., 2] ** 2 + aq[..., 3] ** 2)
s1 = ((aq[..., 0] ** 2 + aq[..., 1] ** 2) + aq[..., 2] ** 2) + aq[..., 3] ** 2
conj = torch.cat([aq[..., :1], -aq[..., 1:]], dim=-1)
print("quat_inv: differing with the pairwise sum", int((conj / s2.clamp(min=1e-9)[..., None] != inv).sum()), "sequential", int((conj / s1.clamp(min=1e-9)[..., None] != inv).sum()), "of", inv.numel())


def run(which, a, b, want, variants, enum):
  res = []
  a, b = a.reshape(N, -1).contiguous(), (b.reshape(N, -1).contiguous() if b is not None else None)
  for v in variants:
    o = torch.zeros_like(want.reshape(N, -1))
    f = lib.probe_enum if enum else lib.probe
    assert f(which, N, a.data_ptr(), b.data_ptr() if b is not None else 0, o.data_ptr(), v, st) == 0
    torch.cuda.synchronize()
    res.append((int((o != want.reshape(N, -1)).sum()), v))
  return sorted(res)[:4]


print("quat_mul(rq, inv)     best (differing, variant):", run(0, rq, inv, d, list(range(48)) + [99], True))
print("yaw_quat(d)           best:", run(1, d, d, dori, range(9), True))
print("quat_mul(dori, bq)    best:", run(0, dori, bq, oq, list(range(48)) + [99], True))
print("quat_apply(dori, rel) best (mode = cross + 3 * sum):", run(2, dori, rel, rot, range(9), False))

# ---- the wider enumeration of quat_mul, component by component (rel_probe2.hip: k_quat_mul_wide)
ci = ctypes.c_int
lib.probe_wide.argtypes = [ci, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]


def wide(name, a, b, want):
  a, b, want = a.reshape(N, 4).contiguous(), b.reshape(N, 4).contiguous(), want.reshape(N, 4)
  o = torch.zeros_like(want)

  def diff(comp, *args):
    assert lib.probe_wide(N, a.data_ptr(), b.data_ptr(), o.data_ptr(), *args, st) == 0
    torch.cuda.synchronize()
    return int((o[:, comp] != want[:, comp]).sum())

  xs = sorted((diff(1, sx, qf, xo, 0, 0, 0), sx, qf, xo) for sx in range(18) for qf in range(2) for xo in range(2))
  print(f"{name}: x component, best (differing, sx, qf, xo):", xs[:3])
  _, sx, qf, xo = xs[0]
  for comp, nm in ((0, "w"), (2, "y"), (3, "z")):
    ms = sorted((diff(comp, sx, qf, xo, m, m, m), m) for m in range(4))
    print(f"{name}: {nm} component given those, best (differing, mode):", ms[:2])


wide("quat_mul(rq, inv)", rq, inv, d)
wide("quat_mul(dori, bq)", dori, bq, oq)
