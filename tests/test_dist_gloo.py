"""N>1 path on CPU: two gloo ranks run bench.run() with a stub step (the decode itself needs a GPU);
covers shard assignment, barrier + MAX-over-ranks timing and the stats all-gather / aggregation."""
import json
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT
from pyruhvro_amd.dist import aggregate, partition_chunks, shard_rows

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import bench
    def make_step(gen_cfg, n, row_lo, num_chunks, dev, local_rank):
        rank = int(os.environ["RANK"])
        def step():
            time.sleep(0.01 * (1 + rank))          # rank 1 is the slow one -> MAX must pick it
            return {"size_kernel_ms": 1.0 + rank, "scan_kernel_ms": 0.5, "emit_kernel_ms": 2.0 * (1 + rank)}
        return step, {"input_bytes": 1000 * (rank + 1), "output_bytes": 2000 * (rank + 1), "row_lo": row_lo}
    args = bench.parse_args(["--gpus", "2", "--steps", "5", "--warmup", "1", "--workload", "full1m", "--records", "1000"])
    rank, world, wall, per_rank, agg, cfg = bench.run(args, make_step, backend="gloo")
    print("RESULT " + json.dumps({"rank": rank, "world": world, "wall": wall, "per_rank": per_rank, "agg": agg}))
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_bench_plumbing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out
        outs.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:]))
    for o in outs:
        assert o["world"] == 2
        assert o["wall"] >= 5 * 0.02 * 0.9                                   # the slow rank's time, on both ranks
        assert abs(o["wall"] - outs[0]["wall"]) < 1e-9                       # MAX all-reduce gave everyone the same number
        assert [r["records"] for r in o["per_rank"]] == [1000, 1000]
        assert [r["input_bytes"] for r in o["per_rank"]] == [1000, 2000]     # all-gather kept rank order
        assert o["agg"]["records_total"] == 2 * 1000 * 5
        assert abs(o["agg"]["records_per_s"] - 10000 / o["wall"]) < 1e-6
        assert o["agg"]["emit_kernel_ms_max"] == 4.0


def test_sharding_helpers():
    assert shard_rows(10, 0) == (0, 10) and shard_rows(10, 3) == (30, 40)
    parts = partition_chunks(103, 8, 2)
    flat = [b for p in parts for b in p]
    assert flat == [(i * 12, 103 if i == 7 else (i + 1) * 12) for i in range(8)]      # reference chunk bounds kept
    assert [len(p) for p in parts] == [4, 4]
    for world in (1, 2, 4, 8):
        parts = partition_chunks(10_000_000, 8, world)
        assert sum(len(p) for p in parts) == 8 and all(len(p) == 8 // world for p in parts)
    assert partition_chunks(0, 4, 2) == [[], [(0, 0)]]
    agg = aggregate([{"records": 5, "input_bytes": 1, "output_bytes": 2, "emit_kernel_ms": 3, "size_kernel_ms": 1}] * 2, 4, 2.0)
    assert agg["records_per_s"] == 20.0
