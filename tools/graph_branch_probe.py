"""Does a captured HIP graph run independent branches concurrently?  Two chains of
`n` dependent small kernels (each chain a latency chain that leaves the chip
idle), on one stream and on two streams joined at the end -- eagerly and inside a
captured graph.  Prints microseconds per pair of chains."""
import sys, time, torch
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
a = torch.zeros((1 << 16,), device=dev); b = torch.zeros((1 << 16,), device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def chains(two):
  cur = torch.cuda.current_stream()
  if not two:
    for _ in range(n): a.add_(1.0)
    for _ in range(n): b.add_(1.0)
    return
  s2.wait_stream(cur)
  for _ in range(n): a.add_(1.0)
  with torch.cuda.stream(s2):
    for _ in range(n): b.add_(1.0)
  cur.wait_stream(s2)


def timeit(fn, reps=20):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  fn(); torch.cuda.synchronize()
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


for two in (False, True):
  with torch.cuda.stream(s1):
    t_eager = timeit(lambda: chains(two))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1):
      chains(two)
    t_graph = timeit(g.replay)
  print('%s: eager %.1f us, graph %.1f us (2 x %d kernels)' % ('two streams' if two else 'one stream ', t_eager, t_graph, n))
