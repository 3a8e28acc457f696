"""A complete stand-in for the tracking task's ``motion.npz`` (reference tasks/tracking/mdp/commands.py:30-65 MotionLoader; none is in
the reference tree): mjlab_amd.rollout.synthetic_motion's joint trajectory and root pose, with the pose and velocity of EVERY
robot body from the CPU oracle's forward kinematics (test infrastructure) -- what MotionCommand needs to build its body-tracking
commands, rewards and terminations.  Keys and shapes follow scripts/csv_to_npz.py:298-309."""

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
  sys.path.insert(0, str(ROOT))


def write_full_motion(path: str, scene: str = "g1_tracking_flat") -> tuple:
  from mjlab_amd import robots
  from mjlab_amd.rollout import synthetic_motion
  from oracle.oracle import OracleSim

  model = robots.load_model(scene)
  mo = synthetic_motion(model)
  nframe = mo["joint_pos"].shape[0]
  o = OracleSim(model, nframe, njmax=250)
  o.qpos[:, :3], o.qpos[:, 3:7], o.qpos[:, 7:] = mo["body_pos_w"][:, 0], mo["body_quat_w"][:, 0], mo["joint_pos"]
  o.qvel[:] = 0.0
  o.qvel[:, 6:] = mo["joint_vel"]
  o.forward(nthread=8)
  root, nb = int(model.jnt_bodyid[0]), model.nbody
  ids = np.arange(root, nb)  # the robot's bodies (the world and the terrain body come first)
  pos, quat, cv = o.xpos.reshape(nframe, nb, 3)[:, ids], o.xquat.reshape(nframe, nb, 4)[:, ids], o.cvel.reshape(nframe, nb, 6)[:, ids]
  sub = o.subtree_com.reshape(nframe, nb, 3)[:, root][:, None, :]
  lin = cv[..., 3:] - np.cross(cv[..., :3], sub - pos)  # velocity of the body origin from the com-based spatial velocity (entity/data.py:20-31)
  np.savez(path, fps=np.array([50.0]), joint_pos=mo["joint_pos"], joint_vel=mo["joint_vel"], body_pos_w=pos.astype(np.float32),
           body_quat_w=quat.astype(np.float32), body_lin_vel_w=lin.astype(np.float32), body_ang_vel_w=cv[..., :3].astype(np.float32))
  return pos.shape
