import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The spawned ranks of the multi-process tests each run the oracle on the host cores: with torch's default of one thread per physical core, two
    # ranks oversubscribe the job's CPU share many times over (a two-rank oracle test: 40-60 s; with 16 threads per rank: 3 s).  The variable is set
    # AFTER this process has initialised torch, so only the children are capped -- the single-process oracle tests measured slower with the cap
    # (large B = 4: 127 s against 108 s).
    try:
        import torch
        torch.get_num_threads()
    except Exception:                                    # noqa: BLE001
        pass
    os.environ.setdefault('OMP_NUM_THREADS', str(max(1, min(16, os.cpu_count() or 1))))


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
