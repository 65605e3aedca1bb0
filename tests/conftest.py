import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (authoring container only)')
    # The CPU oracle runs small networks in most tests: with torch's default of one OpenMP thread per core a 256-core GPU host
    # spends 20 s per tiny iteration in thread barriers (1.5 s on 8 threads).  The full-size test raises the count itself.
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 8))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
