"""Cost of a kernel boundary inside a replayed hipGraph on this GPU: N dependent launches of a trivial kernel on one stream, captured
once and replayed -- (replay time) / N is the floor a step pays per launch on its critical stream, whatever the kernel does."""
import torch
dev = torch.device('cuda:0')
x = torch.zeros(64, device=dev)
big = torch.zeros(15424 * 768, device=dev, dtype=torch.bfloat16)
for name, fn, n in (('1 wave', lambda: x.add_(1.0), 2000), ('24 MB elementwise', lambda: big.add_(1.0), 500)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / 5 / n * 1e3:.2f} us per launch in a replayed graph of {n}')
    # the same launches eagerly
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / n * 1e3:.2f} us per launch, eager')
