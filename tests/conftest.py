import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# The convolution kernel is chosen per call (nlc_conv_desc.policy).  Every GPU parity module that runs convolutions is
# collected TWICE: under the production dispatch ("auto": what bench.py measures - the LDS-halo kernel only for launches
# with >= 128 tiles, otherwise conv_fast / split-K / the generic kernel) and with the LDS-halo kernel forced for every
# eligible shape (the small parity shapes would otherwise never reach it).
@pytest.fixture(params=["auto", "halo"], ids=["production-dispatch", "forced-halo"])
def conv_policy(request):
    from diffusion_nlc_amd import ops
    old = ops.CONV_POLICY
    ops.CONV_POLICY = request.param
    yield request.param
    ops.CONV_POLICY = old
