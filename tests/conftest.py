import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    """CPU-side build of everything the tests load: oracle C walker, generator, engine + extension.
    (hipcc cross-compiles gfx950 without a GPU; prebuilt .so files travel to the GPU box.)"""
    from oracle.build import build as build_oracle
    from avrogen.fastgen import build as build_gen
    from pyruhvro_amd._build import build_all
    build_oracle()
    build_gen()
    build_all()
    yield


def has_gpu() -> bool:
    try:
        import pyruhvro_amd
        return pyruhvro_amd.device_count() > 0
    except Exception:
        return False
