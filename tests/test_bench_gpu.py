"""bench.py end to end on the GPU (a child process, few steps): the ONE JSON line the driver parses carries every field of the
contract -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
config.workload, the roofline object (bound, achieved, peak, unit, frac, traffic) and the per-family breakdown."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract(dev):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-h2d']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['metric'].startswith('video-segments/sec') and d['unit'] == 'video-segments/sec'
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'bf16' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config'] and d['config']['hipgraph'] is True
    assert abs(d['value'] - 8 * 1000.0 / d['ms_per_step']) < 1e-6 * d['value']          # 4 records = 8 video-segment groups per step
    r = d['roofline']
    assert r['unit'] == 'TFLOP/s' and r['peak'] == 2500.0
    # round 6: what bounds the STEP -- the matrix-core time of its FLOPs against the HBM time of the bytes its kernels move (the committed PMC passes);
    # `bound` names the larger floor, achieved / peak / frac stay the GEMM family's, step_hbm prices the step against HBM
    fl = r['step_floors_ms']
    assert set(fl) >= {'mfma', 'hbm_at_8TBs', 'hbm_at_6.3TBs'} and 5.0 < fl['mfma'] < 12.0 and fl['hbm_at_8TBs'] < fl['hbm_at_6.3TBs'] < d['ms_per_step']
    assert r['bound'] == ('hbm' if fl['hbm_at_8TBs'] > fl['mfma'] else 'mfma') and r['kernel_bound'] == 'mfma'
    assert r['step_hbm']['unit'] == 'GB/s' and r['step_hbm']['peak'] == 8000.0 and 0.1 < r['step_hbm']['frac'] < 1.0
    assert 0.05 < r['frac'] < 1.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert r['traffic'] is None or r['traffic'] > 0
    # round 5: the GEMM launches' algorithmic bytes beside the PMC total, a calibration figure around the timed region, graph == eager
    assert 20e9 < r['algorithmic_bytes_per_step'] < 80e9 and (r['traffic_ratio'] is None or 0.8 < r['traffic_ratio'] < 3.0)
    cal = d['config']['calibration_tflops']
    assert len(cal) == 2 and all(500.0 < c_ < 2500.0 for c_ in cal)
    hbm = d['config']['calibration_hbm_tbs']                # round 6: the box's memory speed beside its matrix-core speed
    assert len(hbm) == 2 and all(1.5 < c_ < 8.0 for c_ in hbm)
    assert d['config']['graph_equals_eager'] is True
    b = d['breakdown']
    assert set(b['ms_per_step']) >= {'gemm', 'attention', 'layernorm+reductions', 'optimizer'} and b['sum_ms'] > 0
    assert 'degraded' not in d
    # BASELINE configs 3-5 ride on the same line (one GPU of their eight): ms / step, throughput, whole-step MFMA fraction
    sec = d['config']['secondary']
    assert set(sec) == {'large_b4', 'large_resadapt_b2', 'vcr_large_b4'}
    for k, v in sec.items():
        assert v['ms_per_step'] > 0 and 0.05 < v['step_mfma_frac'] < 1.0 and v['replays'] >= 3, (k, v)
    assert abs(sec['large_b4']['video_segments_per_sec'] - 8 * 1000.0 / sec['large_b4']['ms_per_step']) < 1e-6 * sec['large_b4']['video_segments_per_sec']
    assert abs(sec['vcr_large_b4']['examples_per_sec'] - 4 * 1000.0 / sec['vcr_large_b4']['ms_per_step']) < 1e-6 * sec['vcr_large_b4']['examples_per_sec']


def test_bench_multi_rank_code_path_with_one_rank(dev):
    """--force-comm: the N > 1 program (gloo control plane, the library's RCCL communicator, all-gather / reduce-scatter / five bucket
    all-reduces captured inside the step's graph) with a single rank -- what can be rehearsed on a 1-GPU box."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-h2d',
           '--no-roofline', '--force-comm']
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    c = d['config']
    assert c['comm'] == 'rccl-native' and c['rccl_ranks'] == 1 and c['hipgraph'] is True and 'degraded' not in d
    assert len(c['rccl']['allreduce_bucket_mbytes']) == 5 and c['rccl']['version'] and 'NCCL_ALGO' in c['rccl']
    assert len(c['gradient_buckets']) == 5 and 0.0 < c['exposed_gradient_fraction'] < 0.2
    assert d['value'] > 0 and d['n_gpus'] == 1
    assert c['graph_equals_eager'] is True, 'the one-shot check of the captured multi-rank program against the eager step'

