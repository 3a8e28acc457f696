"""Read-only attribute bridges over device tensors: the drop-in seam of the reference's
``WarpBridge`` (reference: src/mjlab/sim/sim_data.py:174-229).

In the reference every ``sim.data.<field>`` / ``sim.model.<field>`` access yields a
tensor-like object sharing memory with the engine's array; assigning an attribute is
forbidden (it would change a pointer baked into the captured graph), in-place writes
``obj.field[idx] = value`` are the way to modify state.  Here the engine's arrays *are*
torch tensors (allocated by the host side, handed to the C ABI as raw device pointers), so
the bridge returns ``torch.Tensor`` objects directly; the contract is otherwise identical:
same field names and shapes, ``AttributeError`` on assignment, stable ``data_ptr``.
"""

from __future__ import annotations

from typing import Any

import torch


class Bridge:
  def __init__(self, kind: str, tensors: dict[str, torch.Tensor], scalars: dict[str, Any] | None = None,
               on_access: Any = None) -> None:
    object.__setattr__(self, "_kind", kind)
    object.__setattr__(self, "_tensors", tensors)
    object.__setattr__(self, "_scalars", scalars or {})
    object.__setattr__(self, "_on_access", on_access)

  def __getattr__(self, name: str) -> Any:
    t = self._tensors.get(name)
    if t is not None:
      if self._on_access is not None:
        self._on_access(name)  # the caller may be about to write through the returned tensor
      return t
    if name in self._scalars:
      return self._scalars[name]
    raise AttributeError(f"{self._kind} has no field '{name}'")

  def __setattr__(self, name: str, value: Any) -> None:
    raise AttributeError(
      f"Cannot set attribute '{name}' on {self._kind}. "
      f"This wrapper is read-only to preserve memory addresses for captured graphs. "
      f"Use in-place operations instead: obj.{name}[:] = value"
    )

  def __dir__(self):
    return sorted(set(self._tensors) | set(self._scalars))

  def __repr__(self) -> str:
    return f"Bridge({self._kind}, fields={len(self._tensors)})"
