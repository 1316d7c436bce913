"""pytest configuration.

Markers: ``gpu`` = needs a real MI355X (run with ``-m gpu`` on the GPU box); everything else runs on CPU.
CPU tests that execute kernels use the host functional simulator (tests/hostsim): the SAME kernel and
launcher sources compiled for the host, loaded through ``SFAMD_LIBRARY``.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU")
    config.addinivalue_line("markers", "slow: minute-long host-simulator twin of a test that also runs on the GPU (-m gpu); "
                                       "skipped unless SF_RUN_SLOW=1 or selected with -m slow")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("SF_RUN_SLOW") or "slow" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="slow host-simulator twin of a -m gpu test (SF_RUN_SLOW=1 runs it)")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def _use_library(path):
    from slowfast_amd import lib
    if path is None:
        os.environ.pop("SFAMD_LIBRARY", None)
    else:
        os.environ["SFAMD_LIBRARY"] = path
    lib.reset_lib()
    return lib.get_lib()


@pytest.fixture(scope="session")
def hostsim_path():
    from slowfast_amd import build_ext, lib
    return build_ext.build_hostsim(act=lib.ACT_MODE)      # a bf16 process (SF_ACT_DTYPE=bf16) gets the bf16 simulator build


@pytest.fixture()
def sim(hostsim_path):
    """Route slowfast_amd through the host functional simulator for this test; yields the torch device."""
    import torch
    lib = _use_library(hostsim_path)
    assert lib.backend == "hostsim"
    yield torch.device("cpu")
    _use_library(None) if os.path.exists(os.path.join(ROOT, "slowfast_amd", "libsfamd.so")) else None


@pytest.fixture()
def gpu():
    """The product library on cuda:0."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    os.environ.pop("SFAMD_LIBRARY", None)
    from slowfast_amd import lib
    lib.reset_lib()
    assert lib.get_lib().backend == "gfx950"
    return torch.device("cuda:0")
