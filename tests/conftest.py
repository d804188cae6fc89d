import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch brings its own HIP runtime; whichever runtime touches the GPU first in a process keeps it.  The engine is built
    # to share torch's (bench.py and smoke() import torch first), so the test session does the same whatever order the test
    # files run in: initialise torch's runtime before libruhvro_hip.so makes its first HIP call.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:      # no torch / no GPU: the CPU suite does not need it
        pass


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


@pytest.fixture(scope="session", autouse=True)
def _warm_kernel_cache(request, _build_native):
    """GPU runs force the schema-specialised kernels for every test schema.  build() normally left their code
    objects in pyruhvro_amd/_kcache (they travel with the tree); if the cache is cold on this box, compile them in
    parallel once (hiprtc, ~30 s, `python scripts/known_schemas.py`) instead of one by one inside the tests.  A no-op when everything is cached."""
    selected_gpu = any(item.get_closest_marker("gpu") for item in request.session.items)
    if selected_gpu and has_gpu() and not os.environ.get("RUHVRO_HIP_SKIP_WARM"):      # (targeted runs compile what they meet)
        import subprocess
        from pyruhvro_amd.prebuild import cache_looks_warm
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from known_schemas import known_schemas
        if not cache_looks_warm(known_schemas()):
            # a fresh interpreter: this process already talks to the GPU and must not be forked
            subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "known_schemas.py")], cwd=ROOT, check=True, timeout=900)
    yield


def has_gpu() -> bool:
    try:
        import pyruhvro_amd
        return pyruhvro_amd.device_count() > 0
    except Exception:
        return False


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Tests that need a SECOND GPU skip on a one-GPU box: say so in one loud line at the end of the run, so that a green
    summary on such a box is not read as 'the multi-GPU paths ran' (VERDICT round 5, item 8)."""
    skipped = terminalreporter.stats.get("skipped", [])
    need2 = [r for r in skipped if "ONE GPU VISIBLE" in str(getattr(r, "longrepr", ""))]
    if need2:
        terminalreporter.write_sep("!", f"{len(need2)} TEST(S) NOT RUN: THEY NEED A SECOND GPU AND THIS BOX SHOWS ONE", red=True, bold=True)
        for r in need2:
            terminalreporter.write_line(f"    not run (1 GPU visible): {r.nodeid}")
