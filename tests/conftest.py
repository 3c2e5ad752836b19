import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

graft.load_package()


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def hp():
    from danet_amd.hparams import hparams
    hparams.reset()
    yield hparams
    hparams.reset()


@pytest.fixture(autouse=True)
def _restore_library_options():
    '''tests flip library options with _lib.set_option(); every test starts from and leaves
    behind the defaults (+ the DANET_* overrides of the process environment)'''
    yield
    from danet_amd import _lib
    if _lib._lib is not None:
        _lib.apply_env_options()
