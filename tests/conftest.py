import os
import sys

# see bench.py: memset nodes of replayed linear hipGraphs (torch's reduction semaphores) lose their ordering with the
# ROCm 7.2 packet-capture fast path; must be set before the HIP runtime is loaded
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
