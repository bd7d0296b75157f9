"""Do the parallel branches of a replayed hipGraph run CONCURRENTLY on this box?  Two chains of N single-workgroup kernels of ~50 us each
(mr_f32_sum_rows_strided over one 256-column block and many row groups: one workgroup, a serial loop) -- on one stream, on two streams eagerly, and the
same two captured as one graph with a fork and a join.  Concurrent branches halve the time."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import _lib

lib = _lib.load()
dev = torch.device('cuda:0')
G = 12000
xs = [torch.randn(G, 256, device=dev) for _ in range(2)]
outs = [torch.zeros(256, device=dev) for _ in range(2)]
side = torch.cuda.Stream()
N = 40


def k(i):
    _lib.check(lib.mr_f32_sum_rows_strided(xs[i].data_ptr(), 256, G, 1, 0, 256, outs[i].data_ptr(), torch.cuda.current_stream().cuda_stream), 'k')


def one():
    for _ in range(N):
        k(0)
    for _ in range(N):
        k(1)


def two():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(N):
            k(1)
    for _ in range(N):
        k(0)
    main.wait_stream(side)


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    print(f'eager one stream : {timeit(one):.3f} ms for {2 * N} kernels', flush=True)
    print(f'eager two streams: {timeit(two):.3f} ms', flush=True)
    for name, fn in (('one chain', one), ('fork/join', two)):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        print(f'graph {name}: {timeit(g.replay):.3f} ms', flush=True)
