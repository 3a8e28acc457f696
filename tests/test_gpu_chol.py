"""The solve stage's factor + substitution pair, alone, against an fp64 factorization (the gate's tail literals cannot
tell a regression of ``chol_factor_tiles`` / ``chol_solve_tiles`` from the spread of a capped Newton solve -- this test can).

``mjlab_chol_selftest`` (include/mjlab_amd.h) runs the very device functions ``stage_solve.h`` inlines -- the register-resident blocked LDL^T on
MFMA accumulator tiles for the padded sizes 32 / 36 / 48 / 64, the LDS-broadcast column sweep for the others -- on caller-supplied matrices,
one wave per matrix.  Matrices: (i) the Newton Hessians ``H = M + J_a^T D_a J_a`` and the mass matrices of the parity gate's 256 rollout
states (tests/golden/rollout_states_*.npz: fallen, self-colliding robots; condition numbers 1e3..1e6), built in fp64 from the restatement's
arrays; (ii) random SPD matrices of every padded size.  Bounds = measured x 3 (profiles/r06_chol.txt): normwise backward error <= 2e-7
(measured: Hessians 1.5e-8, random 6.8e-8 -- at float32's unit roundoff 6e-8), forward error <= 1.5 eps x condition number (measured 0.44)."""

import ctypes
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
pytestmark = pytest.mark.gpu
EPS = 2.0**-24


def _device_solve(A64: np.ndarray, b64: np.ndarray) -> np.ndarray:
  import torch

  from mjlab_amd import native

  nb, n, _ = A64.shape
  A = torch.from_numpy(np.ascontiguousarray(A64, dtype=np.float32)).to("cuda:0")
  b = torch.from_numpy(np.ascontiguousarray(b64, dtype=np.float32)).to("cuda:0")
  x = torch.zeros_like(b)
  native.check(native.lib().mjlab_chol_selftest(n, nb, A.data_ptr(), b.data_ptr(), x.data_ptr(), torch.cuda.current_stream().cuda_stream), "mjlab_chol_selftest")
  torch.cuda.synchronize()
  return x.cpu().numpy().astype(np.float64), A.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)


def _check(A64, b64, what, backward_max=2e-7, forward_factor=1.5):
  x, A, b = _device_solve(A64, b64)  # (compared with the fp64 solution of the float32-rounded system: the kernel's input)
  assert np.isfinite(x).all(), what
  worst_b, worst_f = 0.0, 0.0
  for k in range(A.shape[0]):
    Ak = np.tril(A[k]) + np.tril(A[k], -1).T  # the kernel reads the lower triangle
    ref = np.linalg.solve(Ak, b[k])
    back = np.abs(Ak @ x[k] - b[k]).max() / (np.abs(Ak).sum(axis=1).max() * np.abs(x[k]).max() + np.abs(b[k]).max())
    fwd = np.abs(x[k] - ref).max() / np.abs(ref).max()
    cond = np.linalg.cond(Ak)
    worst_b, worst_f = max(worst_b, back), max(worst_f, fwd / (cond * EPS))
    assert back <= backward_max, (what, k, back)
    assert fwd <= forward_factor * cond * EPS + 1e-6, (what, k, fwd, cond)
  print(f"{what}: {A.shape[0]} systems of order {A.shape[1]}: worst backward error {worst_b:.2e}, worst forward error / (cond x eps) {worst_f:.3f}")
  return worst_b


@pytest.mark.parametrize("scene", ["g1_velocity_flat", "go1_velocity_flat", "g1_tracking_flat"])
def test_factor_and_substitution_on_the_gates_hessians(scene):
  from make_golden import models

  from oracle.oracle import OracleSim

  z = np.load(ROOT / "tests" / "golden" / f"rollout_states_{scene}.npz")
  model = models()[scene]
  n = min(256, z["qpos"].shape[0])
  o = OracleSim(model, n, njmax=300, ls_parallel=True)
  for f in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(o, f)[:] = z[f][:n]
  o.forward()
  nv = model.nv
  H, M, rhs = np.zeros((n, nv, nv)), np.zeros((n, nv, nv)), np.zeros((n, nv))
  rows = 0
  for w in range(n):
    ne = int(o.nefc[w, 0])
    Mw = o.qM[w].reshape(nv, nv)
    J = o.efc_J[w].reshape(-1, nv)[:ne]
    jar = J @ o.qacc[w] - o.efc_aref[w, :ne]
    act = jar < 0  # the rows in their quadratic zone at the solution: what the last Newton iteration factored
    rows += int(act.sum())
    M[w] = Mw
    H[w] = Mw + (J[act].T * o.efc_D[w, :ne][act]) @ J[act]
    rhs[w] = o.qfrc_smooth[w] + 0.3 * (w % 7) * np.cos(np.arange(nv) + w)
  assert rows >= 2 * n  # (states with contacts -- ~7 active rows per G1 world at the solution: the Hessians are not the mass matrices)
  _check(H, rhs, f"{scene}: Newton Hessians at the solution")
  _check(M, rhs, f"{scene}: mass matrices")


@pytest.mark.parametrize("n", [5, 8, 13, 16, 18, 20, 24, 29, 32, 35, 36, 40, 47, 48, 63, 64])
def test_factor_and_substitution_on_every_padded_size(n):
  rng = np.random.default_rng(n)
  nb = 64
  Q = np.linalg.qr(rng.standard_normal((nb, n, n)))[0]
  ev = np.exp(rng.uniform(np.log(1e-2), np.log(1e2), (nb, n)))  # condition numbers up to 1e4
  A = np.einsum("bij,bj,bkj->bik", Q, ev, Q)
  _check(A, rng.standard_normal((nb, n)), f"random SPD, n = {n}")
