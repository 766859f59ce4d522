import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The float64 / float32 CPU restatements (oracle/) are what the slowest tests spend their time on, and torch's default of one thread per
    # physical core is the wrong setting for their small convolutions on a 128-core host: measured on the GPU box (debug-size joint step, 2 images)
    # float64 31.7 s at 128 threads, 18.3 s at 64, 16.4 s at 32; float32 7.9 / 3.9 / 2.8 s.
    try:
        import torch
        if torch.get_num_threads() > 32:
            torch.set_num_threads(32)
    except ImportError:
        pass


@pytest.fixture(scope='session')
def repo_root():
    return ROOT
