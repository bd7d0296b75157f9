"""Instruction-level probe (scripts/micro/pk_probe.hip) for the packed-fp32 hazard: each packed-instruction form looped on one stream, alone and beside the
small 128 x 128 MFMA GEMM on a second stream; mismatches against the scalar result, per lane."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'libpk_probe.so'))
lib.pk_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
g = torch.Generator().manual_seed(0)
inp = torch.randn(1 << 20, generator=g).to(dev)
a = torch.randn(5760, 136, generator=g).to(BF16).to(dev)
w = (torch.randn(136, 768, generator=g) * 0.1).to(BF16).to(dev)
o = torch.zeros(5760, 768, dtype=BF16, device=dev)
a2, w2, o2 = torch.randn(192, 768, generator=g).to(BF16).to(dev), (torch.randn(768, 768, generator=g) * 0.05).to(BF16).to(dev), torch.zeros(192, 768, dtype=BF16, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
names = ['pk_add op_sel:[0,1] neg', 'pk_add neg', 'pk_add op_sel:[0,1]', 'pk_mul op_sel_hi:[0,1]', 'pk_add plain', 'pk_mov_b32 op_sel:[1,0]', 'pk_fma op_sel:[0,1,0]']
for variant, name in enumerate(names):
    for beside in (False, True):
        bad = torch.zeros(64, dtype=torch.int32, device=dev)
        for rep in range(20):
            if beside:
                with torch.cuda.stream(sb):
                    for _ in range(30):
                        ops.gemm(a, w, o)
                        ops.gemm(a2, w2, o2, transB=True)
            with torch.cuda.stream(sa):
                for _ in range(8):
                    lib.pk_probe(variant, 4096, 2000, inp.data_ptr(), bad.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        b = bad.cpu().tolist()
        print(f'{name:28s} {"beside the small MFMA GEMM" if beside else "alone":28s}: {sum(b)} mismatches; lanes with mismatches: {[i for i, v in enumerate(b) if v][:20]}', flush=True)
