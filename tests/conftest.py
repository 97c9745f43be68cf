import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def lut_arrays():
    from digital_earth_amd import luts
    names, crf = luts.load_crfs()
    return luts.load_cie(), luts.load_srgb2spec(), luts.load_o3(), crf, names


@pytest.fixture(autouse=True)
def _free_device_memory_between_tests():
    """A Renderer of a full-size frame holds tens of GB of device memory (pipeline queues): make sure the previous test's are gone."""
    yield
    import gc
    gc.collect()
