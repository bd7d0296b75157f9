"""hipGraph capture crash hunt: nested fork main -> side -> wg.  python cap_repro2.py <flags: j=inner joins side, m=inner joins main, w=work on side after the join, p=pre-fork inner from main>"""
import torch, sys, faulthandler
faulthandler.enable()
dev = torch.device('cuda:0')
x = torch.zeros(1 << 20, device=dev); y = torch.zeros_like(x)
side, wg2 = torch.cuda.Stream(), torch.cuda.Stream()
flags = sys.argv[1]
def program():
    main = torch.cuda.current_stream()
    if 'p' in flags:
        wg2.wait_stream(main)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        x.add_(1)
        wg2.wait_stream(side)
        with torch.cuda.stream(wg2):
            y.add_(x)
        if 'j' in flags:
            side.wait_stream(wg2)
        if 'w' in flags:
            x.add_(1)
    main.wait_stream(side)
    if 'm' in flags:
        main.wait_stream(wg2)
program(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode=('global' if 'g' in flags else 'thread_local')):
    program()
g.replay(); torch.cuda.synchronize()
print(flags, 'ok', float(y[0]))
