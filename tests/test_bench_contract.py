"""bench.py's host-side contract (CPU): the N > 1 self-launch command and the algorithmic work per unit."""
import importlib
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module('bench')


def test_self_launch_starts_one_rank_per_gpu_as_a_child(monkeypatch):
    """`python bench.py --gpus N` without a launcher: a CHILD `torch.distributed.run` with N ranks on 127.0.0.1 running this very
    file with the same arguments; the parent only relays the return code (it never touches the GPU and never execs)."""
    import subprocess
    bench = _bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env
        return types.SimpleNamespace(returncode=7)
    monkeypatch.setattr(subprocess, 'run', fake_run)
    argv = ['--gpus', '4', '--steps', '5', '--warmup', '2', '--model', 'large']
    rc = bench._self_launch(types.SimpleNamespace(gpus=4), argv)
    cmd = seen['cmd']
    assert rc == 7
    assert cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == argv
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'         # dmabuf IPC: RCCL between processes needs it


def test_algorithmic_work_per_video_segment():
    """the base model's training step is 2.633 TFLOP per video-segment group (a record holds two)."""
    from merlot_reserve_amd.config import load_config
    bench = _bench()
    base = bench.algorithmic_flops_per_record(load_config('base')) / 2
    assert abs(base / 1e12 - 2.633) < 0.01, base
    fwd = bench.algorithmic_flops_per_record(load_config('base'), train=False) / 2
    assert abs(base / fwd - 3.0) < 1e-9
    large = bench.algorithmic_flops_per_record(load_config('large')) / 2
    assert 3.0 < large / base < 3.5           # 24 layers x 1024 wide against 12 x 768


def test_gemm_algorithmic_bytes():
    """ops.gemm_bytes: every operand, output and epilogue operand once (the roofline's algorithmic_bytes_per_step)."""
    import types
    from merlot_reserve_amd import ops
    g = types.SimpleNamespace(M=256, N=512, K=128, c_dtype=ops.MR_DT_BF16, c2=None, residual=None, aux=None, bias=None)
    assert ops.gemm_bytes(g) == 2 * (256 * 128 + 512 * 128) + 2 * 256 * 512
    g = types.SimpleNamespace(M=256, N=512, K=128, c_dtype=ops.MR_DT_F32, c2=1, residual=1, aux=None, bias=1)
    assert ops.gemm_bytes(g) == 2 * (256 * 128 + 512 * 128) + 4 * 256 * 512 + 2 * 2 * 256 * 512 + 2 * 512
