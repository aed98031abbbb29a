import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # a plain `pytest tests` on a box without a GPU runs the CPU suite and reports the GPU tests as skipped
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='needs a real MI355X: run with -m gpu on the GPU box')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def oracle():
    # BL_POWF_LIBM=1 switches BOTH sides to the reference's own JIT build (no -O flag: libm powf under the Newton derivative term,
    # boardlaw/cuda.py:29-45, mcts/cpp/cpu.cpp:60): the product through bl_tune_t.powf_libm (_native.tune reads the variable), the
    # checker through oracle/liboracle_powf.so, which calls the host libm itself
    import oracle_lib
    return oracle_lib.load('_powf' if os.environ.get('BL_POWF_LIBM') == '1' else '')
