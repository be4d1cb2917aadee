import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')
    # make sure liblfgc.so matches the sources before any test binds it (seconds when up to date; hipcc cross-compiles
    # without a GPU).  A missing hipcc is not fatal here: tests that need the library then fail loudly on their own.
    try:
        from latent_feature_grid_compression_amd.build import build
        build(verbose=False)
    except Exception as exc:      # noqa: BLE001
        print('[conftest] could not (re)build liblfgc.so: %s' % exc)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
