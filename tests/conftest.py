import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The oracle runs on the host cores.  A GPU box shows 256 logical CPUs but a job owns a share of ~16 (gpurun): torch's default of one thread per
    # physical core oversubscribes that share four to eight times (bench.py's thread sweep: one oracle record 5.6 s at 16 threads, 21.8 s at 128).
    nthr = max(1, min(16, os.cpu_count() or 1))
    os.environ.setdefault('OMP_NUM_THREADS', str(nthr))      # (inherited by the spawned ranks of the multi-process tests)
    try:
        import torch
        torch.set_num_threads(nthr)
    except Exception:                                    # noqa: BLE001 -- torch missing / already threaded: leave the default
        pass


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
