"""Populate the kernel cache (pyruhvro_amd/_kcache/*.hsaco) with schema-specialised kernels.

hiprtc cross-compiles gfx950 without a GPU, so this runs in the build container; the cache travels
with the tree to the GPU box, where the engine then only loads code objects.  Cache entries are keyed
by a content hash of the generated source + the device headers, so stale entries are never used.
"""
from __future__ import annotations

import os
import sys
from concurrent.futures import ProcessPoolExecutor
from typing import Iterable, List, Tuple


def _one(schema_json: str) -> Tuple[bool, str]:
    from pyruhvro_amd import cabi
    try:
        return cabi.prebuild(schema_json), ""
    except Exception as e:  # noqa: BLE001 - reported to the caller
        return False, str(e)


def _cache_dir() -> str:
    return os.environ.get("RUHVRO_HIP_KERNEL_CACHE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_kcache")


def _marker(schemas: List[str]) -> str:
    import hashlib
    variant = os.environ.get("RUHVRO_HIP_VARIANT", "")      # staged kernel variants have their own code objects
    h = hashlib.sha1(("\0".join(sorted(schemas)) + "\1" + variant).encode()).hexdigest()[:16]
    return os.path.join(_cache_dir(), f"warm_{h}.ok")


def cache_looks_warm(schemas: Iterable[str]) -> bool:
    """True when a prebuild_many() of exactly these schemas completed against the current library build (it leaves a
    marker next to the code objects; rebuilding the library prunes both).  Needs no compile and no child process."""
    return os.path.exists(_marker(list(dict.fromkeys(schemas))))


def prebuild_many(schemas: Iterable[str], jobs: int = 0, verbose: bool = False) -> List[str]:
    """Compile every schema's kernels (parallel processes).  Returns the list of error messages.
    Forks workers: call it from a process that has not initialised the HIP runtime (build(), `python -m
    pyruhvro_amd.prebuild`)."""
    uniq = list(dict.fromkeys(schemas))
    # (half the CPUs as worker processes, each compiling two of its schema's kernels at a time in rh_kcompile helpers: measured
    #  best on 8 vCPUs -- 16 schemas cold: 18 s, against 23-24 s with 8 workers and 32 s with 2)
    jobs = jobs or max(1, min(len(uniq), (os.cpu_count() or 2) // 2, 16))
    os.environ.setdefault("RUHVRO_HIP_COMPILE_JOBS", "2")
    errors: List[str] = []
    if not uniq:
        return errors
    if jobs <= 1:
        results = [_one(s) for s in uniq]
    else:
        with ProcessPoolExecutor(max_workers=jobs) as ex:
            results = list(ex.map(_one, uniq))
    hits = sum(1 for hit, err in results if hit and not err)
    for (hit, err), s in zip(results, uniq):
        if err:
            errors.append(err)
    if not errors:
        try:
            os.makedirs(_cache_dir(), exist_ok=True)
            open(_marker(uniq), "w").close()
        except OSError:
            pass
    if verbose:
        print(f"kernel cache: {len(uniq)} schemas, {hits} already cached, {len(uniq) - hits - len(errors)} compiled, "
              f"{len(errors)} failed")
    return errors


def mark_warm(schemas: Iterable[str]) -> None:
    """Leave the marker cache_looks_warm() looks for: these schemas were prebuilt (possibly by several prebuild_many() calls)."""
    try:
        os.makedirs(_cache_dir(), exist_ok=True)
        open(_marker(list(dict.fromkeys(schemas))), "w").close()
    except OSError:
        pass


def benchmark_schemas() -> List[str]:
    """The schemas bench.py decodes (avrogen is the synthetic-input generator, not the oracle)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from avrogen.schemas import SCHEMAS
    return list(SCHEMAS.values())


if __name__ == "__main__":
    errs = prebuild_many(benchmark_schemas(), verbose=True)
    for e in errs:
        print(e[:2000])
    sys.exit(1 if errs else 0)
