"""Padded dof counts the solve / substep / control-step kernels are instantiated for (csrc/stage_solve.h solve_nvp,
csrc/nvp_launch.h MJLAB_NVP_SIZES, native.NVP_SIZES): the smallest listed size >= nv."""

from .native import NVP_SIZES


def solve_nvp(nv: int) -> int:
  for n in NVP_SIZES:
    if nv <= n:
      return n
  raise ValueError(f"nv = {nv} > {NVP_SIZES[-1]}")
