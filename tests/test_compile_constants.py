"""Independent check of the compile-time constants BOTH the HIP path and the oracle consume
(the parity tests cannot see through a shared input): `dof_invweight0`, `body_invweight0`,
`stat.meaninertia` and the static candidate pair list.

The model compiler derives them from its own numpy kinematics / Jacobian mass matrix
(mjlab_amd/mjcf.py::_set_const, finalize_topology).  Here nothing of that code is used:

  * M(qpos0) is the C oracle's composite-rigid-body qM (oracle/mjoracle.c crb_factor);
  * body Jacobians are FINITE DIFFERENCES of the oracle's forward kinematics (xipos / ximat at
    qpos0 (+) eps e_dof on the configuration manifold, 4th-order central stencil);
  * the definitions are MuJoCo's set0: meaninertia = mean diag M;
    dof_invweight0 = diag(M^-1), averaged over the 3 translational / 3 rotational dofs of a free
    joint; body_invweight0[b] = (tr(Jp M^-1 Jp^T) / 3, tr(Jr M^-1 Jr^T) / 3) at the body's com;
  * the pair list is rebuilt by a brute-force filter written from Appendix B's rule.

Extends what the reference pins itself (tests/test_g1_constants.py:37-124: actuator / keyframe /
collision-attribute constants), which says nothing about these derived arrays.
"""

import math

import numpy as np
import pytest

from mjlab_amd import mjcf, robots
from oracle.oracle import OracleSim


def _perturb(model, qpos, dof, eps):
  q = qpos.copy()
  j = int(model.dof_jntid[dof])
  k = dof - int(model.jnt_dofadr[j])
  qa = int(model.jnt_qposadr[j])
  if model.jnt_type[j] == mjcf.JNT_FREE and k >= 3:
    ax = np.zeros(3)
    ax[k - 3] = 1.0
    q[qa + 3 : qa + 7] = mjcf.quat_mul(q[qa + 3 : qa + 7], np.concatenate([[math.cos(eps / 2)], math.sin(eps / 2) * ax]))
  else:
    q[qa + (k if model.jnt_type[j] == mjcf.JNT_FREE else 0)] += eps
  return q


def _fd_jacobians(model, s, q0, eps=1e-4):
  """(jp, jr)[body] = d xipos / d dof, angular velocity of the inertial frame per unit dof rate (world frame)."""
  nb, nv = model.nbody, model.nv
  jp, jr = np.zeros((nb, 3, nv)), np.zeros((nb, 3, nv))

  def frames(q):
    s.qpos[0] = q
    s.forward()
    return s.xipos[0].copy(), s.ximat[0].reshape(nb, 3, 3).copy()

  for d in range(nv):
    P, R = {}, {}
    for c in (-2, -1, 1, 2):
      P[c], R[c] = frames(_perturb(model, q0, d, c * eps))
    jp[:, :, d] = (P[-2] - 8 * P[-1] + 8 * P[1] - P[2]) / (12 * eps)
    dR = (R[-2] - 8 * R[-1] + 8 * R[1] - R[2]) / (12 * eps)
    _, R0 = frames(q0)
    W = np.einsum("bij,bkj->bik", dR, R0)  # dR R^T = [w]x
    jr[:, 0, d], jr[:, 1, d], jr[:, 2, d] = W[:, 2, 1], W[:, 0, 2], W[:, 1, 0]
  return jp, jr


@pytest.mark.parametrize("name", ["go1_velocity_flat", "g1_velocity_flat"])
def test_invweight0_and_meaninertia_from_first_principles(name):
  model = robots.load_model(name)
  s = OracleSim(model, 1)
  q0 = np.asarray(model.qpos0, dtype=np.float64)
  s.qpos[0] = q0
  s.forward()
  nv = model.nv
  M = s.qM[0].reshape(nv, nv).copy()
  assert np.allclose(M, M.T, atol=1e-13)
  Minv = np.linalg.inv(M)
  assert abs(np.mean(np.diag(M)) - model.meaninertia) <= 1e-10 * model.meaninertia
  want = np.diag(Minv).copy()
  for j in range(model.njnt):
    if model.jnt_type[j] == mjcf.JNT_FREE:
      da = int(model.jnt_dofadr[j])
      want[da : da + 3] = want[da : da + 3].mean()
      want[da + 3 : da + 6] = want[da + 3 : da + 6].mean()
  assert np.abs(want - model.dof_invweight0).max() <= 1e-10 * np.abs(want).max()
  jp, jr = _fd_jacobians(model, s, q0)
  moving = np.asarray(model.body_weldid) != 0
  tran = np.einsum("bid,de,bie->b", jp, Minv, jp) / 3.0
  rot = np.einsum("bid,de,bie->b", jr, Minv, jr) / 3.0
  got = np.asarray(model.body_invweight0)
  assert np.abs(tran[moving] - got[moving, 0]).max() <= 1e-9 * np.abs(tran[moving]).max()
  assert np.abs(rot[moving] - got[moving, 1]).max() <= 1e-9 * np.abs(rot[moving]).max()
  assert np.all(got[~moving] == 0)


@pytest.mark.parametrize("name,npair", [("g1_velocity_flat", 502), ("go1_velocity_flat", 30), ("g1_tracking_flat", 502)])
def test_candidate_pair_list_equals_a_brute_force_filter(name, npair):
  """Appendix B: skip same weld body; skip parent-child only if BOTH weld bodies are not the world;
  skip <exclude> pairs; require (contype1 & conaffinity2) | (contype2 & conaffinity1)."""
  m = robots.load_model(name)
  weld, parent, gbody = np.asarray(m.body_weldid), np.asarray(m.body_parentid), np.asarray(m.geom_bodyid)
  ct, ca = np.asarray(m.geom_contype), np.asarray(m.geom_conaffinity)
  excluded = {(int(sig) >> 16, int(sig) & 0xFFFF) for sig in np.asarray(m.exclude_signature)}
  want = set()
  for g1 in range(m.ngeom):
    for g2 in range(g1 + 1, m.ngeom):
      if not ((ct[g1] & ca[g2]) | (ct[g2] & ca[g1])):
        continue
      b1, b2 = int(gbody[g1]), int(gbody[g2])
      w1, w2 = int(weld[b1]), int(weld[b2])
      if w1 == w2:
        continue
      if w1 != 0 and w2 != 0 and (int(weld[parent[w1]]) == w2 or int(weld[parent[w2]]) == w1):
        continue
      if (min(b1, b2), max(b1, b2)) in excluded:
        continue
      want.add((g1, g2))
  have = {(int(min(a, b)), int(max(a, b))) for a, b in np.asarray(m.pair_geom)}
  assert len(np.asarray(m.pair_geom)) == len(have) == npair  # no duplicates, the survey's count
  assert have == want
  # collision functions are defined for type1 <= type2, and that is how the list stores a pair
  t = np.asarray(m.geom_type)
  assert all(t[a] <= t[b] for a, b in np.asarray(m.pair_geom))
