"""Host side of the environment terms (mjlab_amd/env_terms.py) without a GPU: the ctypes mirrors have the library's sizes, every
entry point is exported, and tensors the kernels could not address are refused before any launch."""

import ctypes
import types

import pytest
import torch

from mjlab_amd import env_terms, native


def test_struct_mirrors_and_exports():
  L = native.lib()
  assert ctypes.sizeof(env_terms.VelocityCommand) == L.mjlab_sizeof_velocity_command()
  assert ctypes.sizeof(env_terms.MotionTables) == L.mjlab_sizeof_motion_tables()
  for name in ("mjlab_event_reset_root_state_uniform", "mjlab_event_reset_joints_by_scale", "mjlab_event_push_by_setting_velocity", "mjlab_command_uniform_velocity",
               "mjlab_command_motion_write", "mjlab_command_motion_relative", "mjlab_command_motion_frame", "mjlab_copy_batch", "mjlab_reward_accumulate"):
    assert hasattr(L, name) and name in native.EXPORTED_SYMBOLS, name


def test_host_tensors_are_refused_not_launched():
  n = 8
  f = lambda *s: torch.zeros(s)  # noqa: E731  (CPU tensors)
  mask = torch.zeros(n, dtype=torch.bool)
  with pytest.raises(TypeError, match="device tensor"):
    env_terms.reset_root_state_uniform(f(n, 36), f(n, 35), 0, 0, mask, f(n, 13), f(n, 3), f(n, 12), f(2, 6), f(2, 6))
  with pytest.raises(TypeError, match="device tensor"):
    env_terms.push_by_setting_velocity(f(n, 35), 0, f(n), 0.02, f(2), f(n, 6), f(n, 4), f(n, 7), f(2, 6))
  term = types.SimpleNamespace(cfg=types.SimpleNamespace(init_velocity_prob=0.5))
  with pytest.raises(NotImplementedError, match="init-velocity"):
    env_terms.command_uniform_velocity(term, None, f(n, 8), f(4, 2), 0.02)


def test_null_and_size_errors_come_back_as_codes():
  """The C entry points validate before launching: callable without a GPU."""
  L = native.lib()
  assert L.mjlab_event_reset_root_state_uniform(None, 36, 0, None, 35, 0, 8, None, None, 0, None, None, 12, None, None, None) == -22
  assert b"null argument" in L.mjlab_last_error()
  assert L.mjlab_reward_accumulate(None, None, None, 1, 8, 0.02, None, None, None, 1, None) == -22
  assert L.mjlab_copy_batch(None, 3, None) == -24 and L.mjlab_copy_batch(None, 0, None) == 0
  assert L.mjlab_command_motion_frame(None, 8, None, None, None, None, 0, None, None, None, None, None, None, None, None, None, None, None, None) != 0
  assert L.mjlab_command_motion_write(None, None, 36, 0, None, 35, 0, None, None, 8, None, None, None, None, 0, None, 0, None, None, 0.0, 0.0, None) == -22
  assert L.mjlab_command_uniform_velocity(None, None) == -22
