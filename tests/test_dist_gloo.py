"""N>1 path, world size 2, gloo rendezvous on 127.0.0.1.

On the CPU box: two ranks run bench.run() with a stub step (the decode itself needs a GPU); covers the strong-scaling
shard assignment (whole reference chunks of ONE list per rank), barrier + MAX-over-ranks timing and the stats
all-gather / aggregation.  On the GPU box (-m gpu): the same two ranks run a REAL decode of their chunks on device 0
(one process per rank, as torch.distributed.run launches bench.py) and check their batches against the oracle."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT
from pyruhvro_amd.dist import aggregate, partition_chunks, shard_rows

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import bench
    EXTRA = %r
    def make_step(gen_cfg, shard, dev, local_rank):
        rank = int(os.environ["RANK"])
        def step():
            time.sleep(0.01 * (1 + rank))          # rank 1 is the slow one -> MAX must pick it
            return {"size_kernel_ms": 1.0 + rank, "scan_kernel_ms": 0.5, "emit_kernel_ms": 2.0 * (1 + rank)}
        return step, {"input_bytes": 1000 * (rank + 1), "output_bytes": 2000 * (rank + 1), "shard": shard}
    args = bench.parse_args(["--gpus", "2", "--steps", "5", "--warmup", "1", "--workload", "full1m", "--records", "1003"] + EXTRA)
    rank, world, wall, per_rank, agg, cfg = bench.run(args, make_step, backend="gloo")
    print("RESULT " + json.dumps({"rank": rank, "world": world, "wall": wall, "per_rank": per_rank, "agg": agg,
                                  "shard": bench.run.info["shard"]}))
    import torch.distributed as dist
    dist.destroy_process_group()
""")

# the same two ranks with a REAL decode: each rank generates only ITS rows of the seeded list, decodes them on
# device 0 through rh_decode_device with the explicit chunk geometry, and compares every batch with the oracle's
# decode of the same rows under the same geometry
REAL_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import torch
    import bench
    from arrow_compare import assert_batches_identical
    from avrogen import fastgen
    from avrogen.schemas import SCHEMAS
    from oracle import c_walker
    from pyruhvro_amd import cabi
    N, K, WORLD = %d, 8, %d
    checked = {}
    def make_step(gen_cfg, shard, dev, local_rank):
        step, info = bench.gpu_step_factory(gen_cfg, shard, torch.device("cuda", 0), 0)
        # the oracle over the WHOLE list, chunked like the reference; this rank must have produced chunks [c0, c1)
        data, offsets = fastgen.generate(gen_cfg, N)
        exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS[gen_cfg]), data, offsets, K, threaded=True)
        mine = exp[shard["chunk_lo"]: shard["chunk_hi"]]
        d_data, d_off = step.keepalive[:2]
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), info["input_bytes"], shard["rows"], SCHEMAS[gen_cfg],
                               shard["chunks"], device=0, chunk_rows=shard["chunk_rows"])
        got = r.to_host()
        assert len(got) == len(mine) == shard["chunks"]
        for g, e in zip(got, mine):
            assert_batches_identical(g, e)
        checked["batches"] = len(got); checked["rows"] = sum(b.num_rows for b in got)
        return step, info
    args = bench.parse_args(["--gpus", str(WORLD), "--steps", "3", "--warmup", "1", "--workload", "full1m", "--records", str(N)])
    rank, world, wall, per_rank, agg, cfg = bench.run(args, make_step, backend="gloo")
    print("RESULT " + json.dumps({"rank": rank, "world": world, "per_rank": per_rank, "agg": agg, "checked": checked}))
    import torch.distributed as dist
    dist.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_two_ranks(tmp_path, text, world=2):
    script = tmp_path / "worker.py"
    script.write_text(text)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=400)
        assert p.returncode == 0, out
        outs.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:]))
    return outs


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_two_rank_gloo_bench_plumbing(tmp_path, scaling):
    outs = _run_two_ranks(tmp_path, WORKER % (ROOT, ["--scaling", scaling]))
    for o in outs:
        assert o["world"] == 2
        assert o["wall"] >= 5 * 0.02 * 0.9                                   # the slow rank's time, on both ranks
        assert abs(o["wall"] - outs[0]["wall"]) < 1e-9                       # MAX all-reduce gave everyone the same number
        assert [r["input_bytes"] for r in o["per_rank"]] == [1000, 2000]     # all-gather kept rank order
        assert o["agg"]["emit_kernel_ms_max"] == 4.0
        if scaling == "strong":      # ONE 1003-record list, 8 chunks of 125 (the last one 128): 4 chunks per rank
            assert [r["records"] for r in o["per_rank"]] == [500, 503]
            assert o["agg"]["records_total"] == 1003 * 5
            assert abs(o["agg"]["records_per_s"] - 1003 * 5 / o["wall"]) < 1e-6
        else:
            assert [r["records"] for r in o["per_rank"]] == [1003, 1003]
            assert o["agg"]["records_total"] == 2 * 1003 * 5
    if scaling == "strong":
        assert [(o["shard"]["row_lo"], o["shard"]["rows"], o["shard"]["chunks"], o["shard"]["chunk_rows"]) for o in outs] == \
            [(0, 500, 4, 125), (500, 503, 4, 125)]
    else:
        assert [o["shard"]["row_lo"] for o in outs] == [0, 1003]


@pytest.mark.gpu
def test_two_ranks_real_decode_of_one_list(tmp_path):
    """Two processes (gloo rendezvous, both on device 0), each decoding its whole chunks of ONE 20003-record list and
    checking them against the oracle's chunks of the whole list: the config-5 path with a real decode under N > 1."""
    outs = _run_two_ranks(tmp_path, REAL_WORKER % (ROOT, ROOT, 20003, 2))
    assert [o["checked"]["batches"] for o in outs] == [4, 4]
    assert sum(o["checked"]["rows"] for o in outs) == 20003
    for o in outs:
        assert [r["records"] for r in o["per_rank"]] == [10000, 10003]
        assert o["agg"]["records_total"] == 20003 * 3 and o["agg"]["emit_kernel_ms_max"] > 0


@pytest.mark.gpu
def test_eight_ranks_one_chunk_each_real_decode(tmp_path):
    """BASELINE config 5's shape as far as one GPU can show it: EIGHT processes (gloo rendezvous, all on device 0), each holding
    exactly one reference chunk of ONE list (rh_decode_device + rh_opts.chunk_rows), each checking its batch against the oracle's
    chunk of the whole list; the stats all-gather and the MAX all-reduce run over the eight ranks (deserialize.rs:57-68, 92-120:
    one task per chunk, ordered join -- here one process per chunk, stats only across them)."""
    n = 400_003
    outs = _run_two_ranks(tmp_path, REAL_WORKER % (ROOT, ROOT, n, 8), world=8)
    assert [o["checked"]["batches"] for o in outs] == [1] * 8
    assert [o["checked"]["rows"] for o in sorted(outs, key=lambda o: o["rank"])] == [n // 8] * 7 + [n - 7 * (n // 8)]
    for o in outs:
        assert o["world"] == 8 and [r["records"] for r in o["per_rank"]] == [n // 8] * 7 + [n - 7 * (n // 8)]
        assert o["agg"]["records_total"] == n * 3 and o["agg"]["emit_kernel_ms_max"] > 0


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
