"""How a persistent GEMM behaves when another kernel (an emulated RCCL all-reduce) already holds some CUs, with the grid of
all 256 CUs and with mr_set_option("gemm_cus", 240) (30 workgroups per XCD: what the data-parallel trainer uses while gradient buckets are
in flight).  python scripts/bench_contention.py [nblocks ...]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merlot_reserve_amd import ops, _lib
dev = torch.device('cuda:0')
_lib.load()
occ = C.CDLL(os.path.join(ROOT, 'scripts', 'micro', 'liboccupier.so'))
occ.occupy.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
WS = torch.zeros(32 * 1024 * 1024, device=dev)
side = torch.cuda.Stream()
shapes = [(15424, 3072, 768), (15424, 768, 3072), (15424, 2304, 768)]
lib = _lib.load()
for nb in [int(x) for x in sys.argv[1:]] or [0, 8, 16, 32]:
  for cus in (256, 240):
    lib.mr_set_option(b'gemm_cus', cus)
    for (m, n, k) in shapes:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm(a, b, c, transB=True, ws=WS)
        torch.cuda.synchronize()
        if nb:
            occ.occupy(nb, 3000.0, sink.data_ptr(), side.cuda_stream)      # holds nb CUs for 3 ms
            torch.cuda._sleep(200000)                                      # let it get resident
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(a, b, c, transB=True, ws=WS)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 100
        print(f'occupied CUs {nb:3d}  gemm_cus {cus}  gemm {m}x{n}x{k}: {t:7.1f} us  {2.0 * m * n * k / t / 1e6:6.0f} TF/s', flush=True)
lib.mr_set_option(b'gemm_cus', 0)
