"""Copies the evidence scripts/r0N_profiles.sh left under gpurun_out/ into profiles/ under the round's names (the judged, committed
copies): rocprofv3 kernel summaries of BASELINE configs 2-5, the PMC passes, the vendor-library comparison.
python scripts/install_profiles.py [round tag, default r03]"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
G, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
copies = [(f'prof_{tag}_base/summary.txt', f'{tag}_base_b4_step_summary.txt'), (f'prof_{tag}_base/kernel_stats.csv', f'{tag}_base_b4_kernel_stats.csv'),
          (f'prof_{tag}_large/summary.txt', f'{tag}_large_b4_step_summary.txt'), (f'prof_{tag}_large_resadapt/summary.txt', f'{tag}_large_resadapt_b2_step_summary.txt'),
          (f'prof_{tag}_vcr_large_b4_summary.txt', f'{tag}_vcr_large_b4_step_summary.txt'), (f'pmc_mfma_{tag}_base_b4.json', f'{tag}_pmc_mfma_base_b4.json'),
          (f'pmc_mfma_{tag}_large_b4.json', f'{tag}_pmc_mfma_large_b4.json')]
for src, dst in copies:
    shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
    print('installed', dst)
with open(os.path.join(G, f'{tag}_gemm_vs_hipblaslt.txt')) as f:
    lines = [l for l in f if 'amdgpu.ids' not in l]
open(os.path.join(P, f'{tag}_gemm_vs_hipblaslt.txt'), 'w').writelines(lines)
k = json.load(open(os.path.join(G, 'pmc_step.json')))
commit = os.environ.get('MR_COMMIT') or (open(os.path.join(G, 'profile_commit.txt')).read().strip() if os.path.exists(os.path.join(G, 'profile_commit.txt')) else 'not recorded')
nsteps = 4        # bench.py --no-graph --steps 2 --warmup 1: 1 eager first step + 1 warm-up + 2 timed
tot = lambda sel: sum(v['launches'] * (v['fetch_bytes_per_launch'] + v['write_bytes_per_launch']) for n, v in k.items() if sel(n)) / nsteps
out = {'_about': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/pmc_step.sh) over "bench.py --no-graph --steps 2 --warmup 1 '
                 '--no-roofline --no-h2d --no-secondary" (base, 4 records/GPU; incl. the eager first step: 4 steps of launches); bytes per launch, '
                 'FETCH_SIZE x2 x1024 per the gfx950 correction (MI355X_MICROARCH.md, HBM section), WRITE_SIZE x1024',
       'bytes_per_step_all_kernels': tot(lambda n: 'cast_params' not in n), 'bytes_per_step_gemm': tot(lambda n: 'gemm' in n or 'splitk' in n),
       'kernels': k, 'workload': {'model': 'base', 'records_per_gpu': 4}, 'commit': commit}
json.dump(out, open(os.path.join(P, f'{tag}_pmc_hbm_traffic.json'), 'w'), indent=1, sort_keys=True)
print(f"installed {tag}_pmc_hbm_traffic.json: {out['bytes_per_step_all_kernels'] / 1e9:.1f} GB / step, {out['bytes_per_step_gemm'] / 1e9:.1f} GB in GEMMs")
