"""The lane-level model of the register-resident blocked LDL^T (tools/chol_mfma_model.py; the HIP code in
mjlab_amd/csrc/common.h `chol_factor_tiles` / `chol_solve_tiles` follows it operation by operation): tile layout, elimination order,
masks and MFMA operand roles give the factor of the matrix, for every padded size the library is instantiated for and both panel
variants.  (The device code itself is exercised by every GPU parity test and by tools/chol_ubench.hip.)"""

import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
import chol_mfma_model as cm  # noqa: E402

SIZES = ((8, 6), (16, 13), (16, 16), (20, 18), (24, 22), (32, 29), (36, 35), (40, 37), (48, 45), (64, 61))


@pytest.mark.parametrize("panel", (0, 1))
@pytest.mark.parametrize("nvp,n", SIZES)
def test_model_solves_the_system(nvp, n, panel, monkeypatch):
  monkeypatch.setattr(cm, "PANEL", panel)
  assert cm.check(nvp, n, seed=3) < 2e-5


@pytest.mark.parametrize("nvp", (8, 20, 36, 48, 64))
def test_elimination_order_is_a_permutation_and_the_factor_is_triangular_in_it(nvp):
  order = cm.elimination_order(nvp)
  assert sorted(order) == list(range(nvp))
  rng = np.random.default_rng(0)
  n = nvp - 1
  B = rng.normal(size=(n, n + 3))
  A = (B @ B.T + 0.5 * np.eye(n)).astype(np.float32)
  C, invd = cm.factor(cm.tiles_from_matrix(A, nvp), nvp)
  pos = {c: s for s, c in enumerate(order)}
  for c in range(nvp):
    for i in range(nvp):
      if pos[i] <= pos[c]:
        assert C[c, i] == 0.0, (c, i)  # exactly zero unless column c is eliminated before row i: what the substitutions rely on
  # L D L^T in elimination order reproduces the matrix
  L = np.eye(nvp)
  for c in range(nvp):
    L[:, c] += C[c, :nvp]
  full = np.eye(nvp)
  full[:n, :n] = A
  rec = L @ np.diag(1.0 / invd[:nvp].astype(np.float64)) @ L.T
  assert np.abs(rec - full).max() <= 2e-5 * np.abs(full).max()
