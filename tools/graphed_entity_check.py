"""The teacher-forced comparisons of tests/_graphed_check.py (graphed step against the EAGER reference step, bit for bit) with an explicit
``fused_entity_data=True``, one task per call (GPU box, reference staged):  python tools/graphed_entity_check.py g1|go1|rough|tracking"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools")); sys.path.insert(0, str(ROOT / "tests"))
import reference_env, _graphed_check
task = sys.argv[1]
kw = {"fused_entity_data": True}
if task == "tracking":
  from mjlab_amd import robots
  from mjlab_amd.rollout import write_motion_npz
  import tempfile
  path = str(Path(tempfile.mkdtemp()) / "motion.npz")
  write_motion_npz(path, robots.load_model("g1_tracking_flat"), "cuda:0")
  def make(n, device, edit):
    def both(cfg):
      cfg.commands.motion.motion_file = path
      edit(cfg)
    return reference_env.make_env("Mjlab-Tracking-Flat-Unitree-G1", num_envs=n, device=device, seed=7, cfg_edit=both)
  st = _graphed_check.run_tracking(make, "cuda:0", num_envs=128, steps=40, capture=True, g_kwargs=kw)
else:
  name = {"g1": "Mjlab-Velocity-Flat-Unitree-G1", "go1": "Mjlab-Velocity-Flat-Unitree-Go1", "rough": "Mjlab-Velocity-Rough-Unitree-G1"}[task]
  def make(n, device, edit):
    return reference_env.make_env(name, num_envs=n, device=device, seed=11, cfg_edit=edit)
  st = _graphed_check.run(make, "cuda:0", num_envs=256, steps=70, capture=True, g_kwargs=kw)
print("FUSED_ENTITY", task, "bit for bit against the eager reference:", json.dumps(st))
