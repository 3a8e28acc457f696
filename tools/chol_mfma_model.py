"""Lane-level model (numpy, CPU) of the register-resident blocked LDL^T of mjlab_amd/csrc/common.h (round 5, `CholTiles`).

The HIP code is written against this model operation by operation: the same tile layout, the same elimination order, the same
masks, the same MFMA operand roles.  `python tools/chol_mfma_model.py` checks the model against numpy for every padded size the
library is instantiated for (tests/test_chol_model.py runs the same check).

Layout.  The symmetric matrix A (NVP x NVP, identity beyond nv) lives in the accumulator registers of
`v_mfma_f32_16x16x4_f32` as UPPER block tiles U(Jb, I), Jb <= I:
    U(Jb, I)[reg r][lane l] = A[16 Jb + 4 kk + r][16 I + j],   kk = l >> 4, j = l & 15
(the C / D layout of the instruction: row 4 (l >> 4) + r, column l & 15).  One REGISTER r of the tile row Jb therefore holds four
matrix rows {16 Jb + 4 kk + r : kk = 0..3}, one per 16-lane group -- exactly the shape of an MFMA A / B operand (K index = lane
group).  The factorization eliminates the columns of a block in that order: panel r = columns c(kk) = 16 Jb + 4 kk + r, kk = 0..3,
panels r = 0..3 -- a fixed permutation inside every 16-column block (position of column 16 Jb + j: 16 Jb + 4 (j & 3) + (j >> 2)),
so that a panel's rows are MFMA operands AS THEY LIE, with no transposition through LDS:
    W  = panel x Linv(4x4 block)^T        one MFMA per tile (the block's inverse factor in the A operand, rows 0 4 8 12)
    U(J', I) -= W(J')^T-ish x L(I)        one MFMA per trailing tile  (A operand = -W of block J', B operand = L of block I)
"""
from __future__ import annotations

import numpy as np

MINVAL = 1e-15
PANEL = 0  # MJLAB_CHOL_PANEL of common.h
F = np.float32


def mfma_16x16x4(a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
  """D = A B + C of v_mfma_f32_16x16x4_f32: a[l] = A[l & 15][l >> 4], b[l] = B[l >> 4][l & 15], c[l][r] = C[4 (l >> 4) + r][l & 15]."""
  A = np.zeros((16, 4), F)
  B = np.zeros((4, 16), F)
  for l in range(64):
    A[l & 15, l >> 4] = a[l]
    B[l >> 4, l & 15] = b[l]
  D = (A.astype(np.float64) @ B.astype(np.float64)).astype(F)  # (the model does not imitate the instruction's internal rounding)
  out = c.copy()
  for l in range(64):
    for r in range(4):
      out[l, r] = c[l, r] + D[4 * (l >> 4) + r, l & 15]
  return out


def nblocks(nvp: int) -> int:
  return (nvp + 15) // 16


def tile_index(jb: int, i: int) -> int:
  return i * (i + 1) // 2 + jb


def pos16(j: int) -> int:
  """elimination position of column j of a 16-column block"""
  return 4 * (j & 3) + (j >> 2)


def elimination_order(nvp: int) -> list[int]:
  """columns < nvp in the order the factorization eliminates them (what the substitutions walk)"""
  nb = nblocks(nvp)
  order = []
  for jb in range(nb):
    if last_block_small(nvp) and jb == nb - 1:
      order += [16 * jb + r for r in range(4)]
      continue
    for r in range(4):
      for kk in range(4):
        c = 16 * jb + 4 * kk + r
        if c < nvp:
          order.append(c)
  return order


def last_block_small(nvp: int) -> bool:
  """the last block holds 4 real columns (NVP 20, 36): one 4x4 block factored on its own, no panels"""
  return nvp - 16 * (nblocks(nvp) - 1) == 4


def tiles_from_matrix(A: np.ndarray, nvp: int) -> np.ndarray:
  """U tiles [tile][lane][reg] of the symmetric matrix A (n x n, n <= nvp), identity beyond n"""
  nb = nblocks(nvp)
  n = A.shape[0]
  full = np.eye(16 * nb, dtype=F)
  full[:n, :n] = A
  T = np.zeros((nb * (nb + 1) // 2, 64, 4), F)
  for i in range(nb):
    for jb in range(i + 1):
      for l in range(64):
        kk, j = l >> 4, l & 15
        for r in range(4):
          T[tile_index(jb, i), l, r] = full[16 * jb + 4 * kk + r, 16 * i + j]
  return T


def ldl4(a):
  """uniform LDL^T of a 4 x 4 block (lower entries a[i][j], j <= i) with the pivot clamp; returns d, 1 / d, L, inverse of L"""
  d = [F(0)] * 4
  inv = [F(0)] * 4
  L = np.eye(4, dtype=F)
  clamp = lambda x: F(min(max(x, F(MINVAL)), F(3.0e38)))  # noqa: E731
  d[0] = clamp(a[0][0]); inv[0] = F(1) / d[0]
  L[1, 0] = a[1][0] * inv[0]; L[2, 0] = a[2][0] * inv[0]; L[3, 0] = a[3][0] * inv[0]
  d[1] = clamp(a[1][1] - L[1, 0] * a[1][0]); inv[1] = F(1) / d[1]
  t21 = a[2][1] - L[2, 0] * a[1][0]; L[2, 1] = t21 * inv[1]
  t31 = a[3][1] - L[3, 0] * a[1][0]; L[3, 1] = t31 * inv[1]
  d[2] = clamp(a[2][2] - L[2, 0] * a[2][0] - L[2, 1] * t21); inv[2] = F(1) / d[2]
  t32 = a[3][2] - L[3, 0] * a[2][0] - L[3, 1] * t21; L[3, 2] = t32 * inv[2]
  d[3] = clamp(a[3][3] - L[3, 0] * a[3][0] - L[3, 1] * t31 - L[3, 2] * t32); inv[3] = F(1) / d[3]
  M = np.eye(4, dtype=F)
  M[1, 0] = -L[1, 0]
  M[2, 1] = -L[2, 1]
  M[3, 2] = -L[3, 2]
  M[2, 0] = -(L[2, 0] + L[2, 1] * M[1, 0])
  M[3, 1] = -(L[3, 1] + L[3, 2] * M[2, 1])
  M[3, 0] = -(L[3, 0] + L[3, 1] * M[1, 0] + L[3, 2] * M[2, 0])
  return d, inv, L, M


def factor(T: np.ndarray, nvp: int):
  """In: tiles of A.  Out: C[c][i] = Lu[i][c] (column-major unit-lower factor in natural indices, zero unless column c is
  eliminated before row i) and invd[c] = 1 / D_c."""
  nb = nblocks(nvp)
  T = T.copy()
  C = np.zeros((16 * nb, 16 * nb), F)
  invd = np.ones(16 * nb, F)
  lanes = np.arange(64)
  kk_l, j_l = lanes >> 4, lanes & 15
  for jb in range(nb):
    tjj = tile_index(jb, jb)
    if last_block_small(nvp) and jb == nb - 1:
      # 4 real columns 16 jb + r (lane group 0, registers 0..3): a[r][r'] = U(jb, jb)[reg r][lane r']
      a = [[T[tjj, rp, r] for rp in range(4)] for r in range(4)]
      d, inv, L, _ = ldl4(a)
      for r in range(4):
        invd[16 * jb + r] = inv[r]
        for rp in range(r):
          C[16 * jb + rp, 16 * jb + r] = L[r, rp]
      continue
    for r in range(4):
      # 1. the panel's diagonal block: a[kk][kk'] = A[c_kk][c_kk'] = U(jb, jb)[reg r][lane 16 kk + 4 kk' + r]
      a = [[T[tjj, 16 * kk + 4 * kp + r, r] for kp in range(4)] for kk in range(4)]
      d, inv, L, M = ldl4(a)
      inv_l = np.array([inv[k] for k in kk_l], F)
      Wm, Lm = {}, {}
      # 2. W = panel x Linv^T.  PANEL == 0 (default): one MFMA per tile -- A operand: rows 0, 4, 8, 12 of a 16 x 4 matrix hold Linv
      #    (row 4 q = Linv[q][:]), B operand: the panel's register as it lies; register 0 of the result is W in operand layout.
      #    PANEL == 1 (experiment): substitution -- every lane gets its row's four panel entries (the same lane column j in the four
      #    16-lane groups: one v_permlane16_swap + two v_permlane32_swap), forms w_0..w_3 and keeps the one of its own group
      q_l = j_l >> 2
      G = np.where(((j_l & 3) == 0) & (q_l >= kk_l), M[q_l, kk_l], F(0)).astype(F)
      for i in range(jb, nb):
        x = T[tile_index(jb, i), :, r]
        if PANEL == 0:
          W = mfma_16x16x4(G, x, np.zeros((64, 4), F))[:, 0]  # lane (g, j): W[16 i + j][c_g]
        else:
          B = [x[16 * k + j_l] for k in range(4)]  # B[k][lane] = x[lane column j of group k] = A[16 i + j][c_k]
          w0 = B[0]
          w1 = (B[1] - w0 * L[1, 0]).astype(F)
          w2 = ((B[2] - w0 * L[2, 0]).astype(F) - w1 * L[2, 1]).astype(F)
          w3 = (((B[3] - w0 * L[3, 0]).astype(F) - w1 * L[3, 1]).astype(F) - w2 * L[3, 2]).astype(F)
          W = np.choose(kk_l, [w0, w1, w2, w3]).astype(F)
        valid = np.ones(64, bool) if i > jb else (np.array([pos16(j) for j in j_l]) > 4 * r + kk_l)
        Wm[i] = np.where(valid, W, F(0)).astype(F)
        Lm[i] = (Wm[i] * inv_l).astype(F)
      # 3. the factor's columns c_kk = 16 jb + 4 kk + r, all rows (zeros above and in earlier blocks)
      for l in range(64):
        c = 16 * jb + 4 * kk_l[l] + r
        invd[c] = inv[kk_l[l]]
        for i in range(nb):
          C[c, 16 * i + j_l[l]] = Lm[i][l] if i >= jb else F(0)
      # 4. trailing update of every tile that still holds uneliminated entries
      for jp in range(jb, nb):
        if jp == jb and r == 3:
          continue  # (the tile row of this block is finished)
        for i in range(jp, nb):
          T[tile_index(jp, i)] = mfma_16x16x4((-Wm[jp]).astype(F), Lm[i], T[tile_index(jp, i)])
  return C, invd


def solve(C: np.ndarray, invd: np.ndarray, b: np.ndarray, nvp: int) -> np.ndarray:
  """lane i owns b_i; forward over the columns in elimination order, scale, backward in reverse order"""
  x = b.astype(F).copy()
  order = elimination_order(nvp)
  for c in order:
    x = (x - C[c, : len(x)] * x[c]).astype(F)  # C[c][i] = 0 for rows not below c
  x = (x * invd[: len(x)]).astype(F)
  for c in reversed(order):
    x = (x - C[: len(x), c] * x[c]).astype(F)  # Lu[c][i]: zero unless c is eliminated after i
  return x


def check(nvp: int, n: int, seed: int = 0) -> float:
  rng = np.random.default_rng(seed)
  B = rng.normal(size=(n, n + 5))
  A = (B @ B.T + 0.1 * np.eye(n)).astype(F)
  C, invd = factor(tiles_from_matrix(A, nvp), nvp)
  b = rng.normal(size=n).astype(F)
  bp = np.zeros(nvp, F)
  bp[:n] = b
  x = solve(C[:nvp, :nvp], invd[:nvp], bp, nvp)[:n]
  ref = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
  return float(np.abs(x - ref).max() / np.abs(ref).max())


if __name__ == "__main__":
  import sys

  PANEL = int(sys.argv[1]) if len(sys.argv) > 1 else 0
  for nvp, n in ((8, 6), (16, 13), (20, 18), (24, 22), (32, 29), (36, 35), (40, 37), (48, 45), (64, 61)):
    print(nvp, n, f"{check(nvp, n):.2e}")
