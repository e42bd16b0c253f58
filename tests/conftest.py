import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the in-tree native artefacts (gfx950 C-ABI library, cf_c module, oracle) are git-ignored build products: build them
    # once if this checkout has none yet (hipcc cross-compiles without a GPU; ~1 minute)
    lib = os.path.join(ROOT, "heat_amd", "lib", "libheat_cf.so")
    oracle_lib = os.path.join(ROOT, "oracle", "libcf_oracle.so")
    have_cf_c = any(f.startswith("cf_c.") and f.endswith(".so") for f in os.listdir(os.path.join(ROOT, "heat_amd")))
    if not (os.path.exists(lib) and os.path.exists(oracle_lib) and have_cf_c):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
