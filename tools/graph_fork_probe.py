"""Does a captured hipGraph run independent branches concurrently, and what does a fork / join cost?  K small elementwise chains
(each `depth` kernels on 4096 x 32 floats, like a reward term of the reference) captured (a) on one stream, (b) dealt out over S
side streams between one fork and one join.  Prints the replay time per graph.  python tools/graph_fork_probe.py  (GPU box)"""

import torch

dev = "cuda:0"
n, w = 4096, 32


def chain(x, depth):
  y = x
  for i in range(depth):
    y = y * 1.0001 + 0.5 if i % 2 == 0 else torch.square(y)
  return y.sum(dim=1)


def build(K, depth, S):
  xs = [torch.rand(n, w, device=dev) for _ in range(K)]
  out = torch.zeros(K, n, device=dev)
  side = [torch.cuda.Stream(device=dev) for _ in range(S)]

  def body():
    main = torch.cuda.current_stream()
    if S == 0:
      for k in range(K):
        out[k] = chain(xs[k], depth)
      return
    ev0 = torch.cuda.Event()
    ev0.record(main)
    for s in side:
      s.wait_event(ev0)
    res = [None] * K
    for k in range(K):
      with torch.cuda.stream(side[k % S]):
        res[k] = chain(xs[k], depth)
    for s in side:
      ev = torch.cuda.Event()
      ev.record(s)
      main.wait_event(ev)
    for k in range(K):
      res[k].record_stream(main)
      out[k] = res[k]

  st = torch.cuda.Stream(device=dev)
  with torch.cuda.stream(st):
    for _ in range(3):
      body()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    body()
  ref = out.clone()
  g.replay()
  torch.cuda.synchronize()
  assert torch.equal(out, ref)
  t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  for _ in range(10):
    g.replay()
  t0.record()
  for _ in range(100):
    g.replay()
  t1.record()
  torch.cuda.synchronize()
  return t0.elapsed_time(t1) / 100 * 1e3


for K, depth in ((12, 5), (30, 8), (60, 2)):
  base = build(K, depth, 0)
  line = f"{K} terms x {depth} kernels (+ sum, + copy = {K * (depth + 2)} nodes): one stream {base:7.1f} us"
  for S in (2, 4, 8):
    line += f" | {S} streams {build(K, depth, S):7.1f} us"
  print(line)
