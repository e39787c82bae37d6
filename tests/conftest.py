import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def native_library_built():
    """A fresh checkout has no libskp_hip.so (built artefacts are git-ignored): build it once per session
    (hipcc cross-compiles gfx950 without a GPU)."""
    lib = os.path.join(ROOT, "stablekeypoints_amd", "csrc", "libskp_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-C", os.path.dirname(lib), "-j8"], check=True)
    return lib


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))
    return load


@pytest.fixture
def tune():
    """Developer overrides of the library's launch plans (include/skp.h: skp_tune_set), restored after the test."""
    from stablekeypoints_amd import _native as N
    touched = set()

    def set_(key, value):
        touched.add(key)
        N.tune(key, value)
    yield set_
    for key in touched:
        N.tune(key, 0)


@pytest.fixture(scope="session")
def sd15_cpu():
    """The full-width SD-1.5 tree on the HOST (seeded synthetic weights: the same tensors `load_ldm("cuda", "sd15")` draws), built
    once per session for the oracle's reference-order CPU steps (each test registers its own reference hook on it)."""
    from stablekeypoints_amd.optimize_token import load_ldm
    cpu, _, _ = load_ldm("cpu", "sd15", feature_upsample_res=128)
    return cpu
