import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "marl-hideandseek_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/), built on demand.  Test infrastructure only."""
    import hs_ref
    hs_ref.build()
    return hs_ref


@pytest.fixture(scope="session")
def hideseek_lib():
    """Path of the in-tree HIP library; built with hipcc if missing (cross-compiles without a GPU)."""
    import build as hs_build
    return hs_build.build_lib()


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """The HIP library and the headless driver exist before any test imports the package (no-op when up to date)."""
    import build as hs_build
    hs_build.build_lib()
    try:
        hs_build.build_headless()
        hs_build.build_smallcap()
    except Exception:
        pass
