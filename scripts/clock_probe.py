"""Sample the shader clock (rocm-smi) while a GEMM loop / the training step runs: is the MFMA peak we price against
(2.5 PF at 2.4 GHz) reachable under sustained load?"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
WS = torch.zeros(32 * 1024 * 1024, device=dev)
samples = []
stop = False
def watch():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out))
        except Exception as e:
            samples.append((time.time(), str(e)))
        time.sleep(0.2)
th = threading.Thread(target=watch); th.start()
time.sleep(1.0)
m, n, k = 8192, 8192, 8192
a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(k, n, device=dev).to(torch.bfloat16); c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 4.0:
    for _ in range(50):
        ops.gemm(a, b, c, ws=WS)
    torch.cuda.synchronize()
t1 = time.time()
time.sleep(1.0)
stop = True; th.join()
import json, re
for t, out in samples:
    try:
        j = json.loads(out)
        card = next(iter(j.values()))
        keys = {k: v for k, v in card.items() if 'sclk' in k.lower() or 'power' in k.lower() or 'mclk' in k.lower()}
        print(f'{t - t0:6.2f}s {"LOAD" if t0 <= t <= t1 else "idle"} {keys}')
    except Exception:
        print(f'{t - t0:6.2f}s raw {out[:200]!r}')
