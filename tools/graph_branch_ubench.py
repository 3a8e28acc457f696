"""Does a hipGraph run independent branches of tiny kernels concurrently?  100 elementwise launches on (4096, 29) floats captured as ONE
chain on one stream against B chains on B streams (forked from and joined into the capturing stream), replayed 200 times.
  python tools/graph_branch_ubench.py"""
import time

import torch

dev = "cuda:0"
N, W, K = 4096, 29, 96


def build(branches: int):
  xs = [torch.rand((N, W), device=dev) for _ in range(branches)]
  outs = [torch.empty_like(x) for x in xs]
  streams = [torch.cuda.Stream(device=dev) for _ in range(branches - 1)]

  def body():
    main = torch.cuda.current_stream(dev)
    per = K // branches

    def chain(b):
      y = xs[b]
      for _ in range(per):
        y = y * 1.0001 + 0.5
      outs[b].copy_(y)

    if branches == 1:
      chain(0)
      return
    for s in streams:
      s.wait_stream(main)
    for b, s in enumerate(streams):
      with torch.cuda.stream(s):
        chain(b + 1)
    chain(0)
    for s in streams:
      main.wait_stream(s)

  s0 = torch.cuda.Stream(device=dev)
  s0.wait_stream(torch.cuda.current_stream(dev))
  with torch.cuda.stream(s0):
    body()
  torch.cuda.current_stream(dev).wait_stream(s0)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    body()
  return g


for branches in (1, 2, 4, 8):
  g = build(branches)
  for _ in range(20):
    g.replay()
  torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(200):
    g.replay()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t) / 200
  print(f"BRANCH {branches} branches x {2 * (K // branches) + 1} launches each: {dt * 1e6:.1f} us per replay, {dt * 1e6 / (2 * K + branches):.2f} us per launch")


def build_mixed(kind: str):
  """One chain of 96 multiply-adds with every 4th step followed by `kind`: a contiguous copy_ (memcpy node), a zero_ (memset node), a
  strided copy (copy kernel), a gather, a different elementwise kernel."""
  x = torch.rand((N, W), device=dev)
  out = torch.empty_like(x)
  tmp = torch.empty_like(x)
  idx = torch.randperm(W, device=dev)

  def body():
    y = x
    for k in range(K):
      y = y * 1.0001 + 0.5
      if k % 4 == 3:
        if kind == "memcpy":
          tmp.copy_(y); y = tmp * 1.0
        elif kind == "memset":
          tmp.zero_(); y = y + tmp
        elif kind == "strided_copy":
          tmp.t().copy_(y.t()[:, :]); y = tmp * 1.0
        elif kind == "gather":
          y = y[:, idx]
        elif kind == "other_kernel":
          y = torch.sin(y); y = y * 1.0
    out.copy_(y)

  s0 = torch.cuda.Stream(device=dev)
  s0.wait_stream(torch.cuda.current_stream(dev))
  with torch.cuda.stream(s0):
    body()
  torch.cuda.current_stream(dev).wait_stream(s0)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    body()
  return g


for kind in ("none", "memcpy", "memset", "strided_copy", "gather", "other_kernel"):
  g = build_mixed(kind)
  for _ in range(20):
    g.replay()
  torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(200):
    g.replay()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t) / 200
  extra = 0 if kind == "none" else K // 4
  print(f"MIXED {kind:13s}: {dt * 1e6:.1f} us per replay ({2 * K + 1} multiply-add launches + {extra} x {kind})")
