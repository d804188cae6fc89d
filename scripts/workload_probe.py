"""Device-resident decode of one synthetic workload: kernel times (HIP timestamps of the launches, rh_stats), the tile
statistics of rh_engine_counters, and buffer identity with the oracle -- bench.py's `workload_line` as a stand-alone command
for A/B runs (env knobs: RUHVRO_HIP_VARIANT, RUHVRO_HIP_WIN_BYTES, RUHVRO_HIP_RANGED, RUHVRO_HIP_NO_TRUST ...).

    python scripts/workload_probe.py full_realistic 1000000 [--kernel generic] [--chunks 8] [--reps 20] [--no-parity]

Prints ONE JSON line.  Needs an MI355X."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("records", type=int)
    ap.add_argument("--chunks", type=int, default=8)
    ap.add_argument("--kernel", default="auto", choices=("auto", "generic", "specialized"))
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--parity-max", type=int, default=0)
    a = ap.parse_args()
    import torch  # noqa: F401  (first: the library binds to torch's HIP runtime)
    import bench
    print(json.dumps(bench.workload_line(a.workload, a.records, a.chunks, a.kernel, a.reps, not a.no_parity, a.parity_max)))


if __name__ == "__main__":
    main()
