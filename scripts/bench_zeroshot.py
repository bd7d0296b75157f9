"""Zero-shot forward (BASELINE config 1) timing: embed_video of 8 segments at grid 18x32, random weights.
python scripts/bench_zeroshot.py [base|large] [fp32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merlot_reserve_amd import modeling as M, preprocess as P
name = sys.argv[1] if len(sys.argv) > 1 else 'base'
bf16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
pm = M.PretrainedMerlotReserve.from_random(name, image_grid_size=(18, 32), device='cuda:0', use_bfloat16=bf16)
rng = np.random.default_rng(0)
segs = [{'patches': rng.random((576, 768)).astype(np.float32), 'text': rng.integers(10, 32768, size=12).tolist() + [3]}]
for i in range(1, 8):
    segs.append({'patches': rng.random((576, 768)).astype(np.float32), 'spectrogram': (rng.random((3, 60, 65)) * 5).astype(np.float32), 'use_text_as_input': False})
v = P.preprocess_video(segs, (18, 32))
v = {k: torch.from_numpy(x).cuda() for k, x in v.items()}
for _ in range(2):
    out = pm.embed_video(**v)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = pm.embed_video(**v)
torch.cuda.synchronize()
print(f'embed_video {name} {"bf16" if bf16 else "fp32"} 8 segments 18x32: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per call, out {tuple(out.shape)} {out.dtype}')
