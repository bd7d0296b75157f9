"""What a kernel NODE costs in a replayed hipGraph on this box: a chain of N dependent tiny kernels (one stream) and the same N split over two
forked streams, captured and replayed; us per node.  The step's graph has ~900 kernel nodes on three streams (bench.py)."""
import time
import torch

dev = torch.device('cuda:0')
x = torch.zeros(64, device=dev)
y = torch.zeros(64, device=dev)
side = torch.cuda.Stream()


def chain(n, two):
    if not two:
        for _ in range(n):
            x.add_(1.0)
        return
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(n // 2):
            y.add_(1.0)
    for _ in range(n // 2):
        x.add_(1.0)
    main.wait_stream(side)


for two in (False, True):
    for n in (200, 1000):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            chain(n, two)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                chain(n, two)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            # eager for comparison
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                chain(n, two)
            torch.cuda.synchronize()
            de = (time.perf_counter() - t0) / reps
        print(f'streams={2 if two else 1} nodes={n}: replay {dt * 1e6 / n:.2f} us/node ({dt * 1e3:.3f} ms), eager {de * 1e6 / n:.2f} us/node', flush=True)
