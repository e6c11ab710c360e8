import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    if p not in sys.path:
        sys.path.insert(0, p)

# OMNI_EMU=1 python -m pytest tests/test_gpu_*.py -m gpu : REHEARSAL of the `-m gpu` tests without a GPU — the product's device path
# runs on the host emulation of its own kernels (tests/emu).  Sizes are the GPU's, so this takes minutes to hours; it is a
# developer tool, never what the driver runs (OMNI_EMU is unset there and on the GPU box).
REHEARSE = os.environ.get("OMNI_EMU", "0") == "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu or REHEARSE:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _rehearse_on_emulation(request):
    if not (REHEARSE and "gpu" in request.keywords):
        yield
        return
    import torch
    import gpu_checks  # noqa: F401  (emulated_device redirects its DEV / _sync)
    from emu_runtime import emulated_device
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with emulated_device():
            yield
    finally:
        torch.Tensor.cuda = cuda


@pytest.fixture
def emu():
    """host emulation of the device library bound behind omniparser_amd._lib for the duration of one (CPU) test."""
    import gpu_checks  # noqa: F401
    from emu_runtime import emulated_device
    with emulated_device() as L:
        yield L
