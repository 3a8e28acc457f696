"""Run-to-run spread of the parity gate's statistics (GPU box): the same scene / rollout length under several seeds, for the library
named by MJLAB_AMD_LIB.  Used in round 5 to tell a change of arithmetic (another elimination order of the same LDL^T) from noise.
  MJLAB_AMD_LIB=gpurun_prof/ab_t1.so python tools/parity_seeds.py g1_tracking_flat 250 5"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from parity_report import scene_report  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "g1_tracking_flat"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 250
nseed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
expand = ("geom_friction", "body_ipos", "qpos0") if "tracking" in scene else ("geom_friction",)
for s in range(nseed):
  r = scene_report(scene, 1024, steps, "f64", seed=1000 + 17 * s, expand=expand)
  f = r["fields"]
  print(f"{scene} {steps} steps seed {1000 + 17 * s}: " + "  ".join(f"{k} med {f[k][0]:.2e} p99 {f[k][1]:.2e} max {f[k][2]:.2e}" for k in ("qacc_smooth", "qacc", "qfrc_constraint", "step_qvel"))
        + f"  elem qacc {r['elem']['qacc'][3]:.3f}", flush=True)
