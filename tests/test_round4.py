"""Round-4 evidence tests (all need an MI355X):

* the configuration bench.py TIMES -- rh_decode_device + RH_ASYNC + explicit chunk geometry (rh_opts.chunk_rows), with and
  without the call dealing its chunk groups to internal streams -- at the benchmark's full size, buffer for buffer
  against the oracle (the reference's own check is assert_round_trip, ruhvro/src/fast_decode.rs:945-953);
* bench.py through a one-rank RCCL group (backend "nccl"): init_process_group(device_id=...), barrier(device_ids=...), the
  MAX all-reduce and the all-gather of pyruhvro_amd/dist.py run for real on one GPU (the reference's only cross-shard
  step is the ordered join of ruhvro/src/deserialize.rs:115-119; ours is stats only);
* the bench line carries `parity_check` for its own timed configuration;
* RUHVRO_HIP_NO_TRUST=1 (the emit pass keeps its own bounds / anomaly checks in every tile) produces the same buffers;
* RUHVRO_HIP_POISON=1 (pooled memory handed out as 0xA5): both directions still produce the oracle's bytes;
* the CPython boundary resolves "current device" on the calling thread.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from arrow_compare import assert_batches_identical
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from oracle import c_walker

import pyruhvro_amd as P
from pyruhvro_amd import cabi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra_args, extra_env, timeout=600):
    env = dict(os.environ)
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def ten_million():
    n = 10_000_000
    data, offsets = fastgen.generate("full", n)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["full"]), data, offsets, 8, threaded=True)
    d_data = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda:0")
    d_data[: len(data)].copy_(torch.from_numpy(data))
    d_off = torch.from_numpy(offsets.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    return n, offsets, exp, d_data, d_off


@pytest.mark.parametrize("internal_streams", ["1", "2", "3"])
def test_timed_configuration_at_10m_is_identical_to_the_oracle(ten_million, internal_streams, monkeypatch):
    """What bench.py times: the device-resident call made with RH_ASYNC on torch's stream, settled later, with the list's
    chunk geometry given explicitly -- the whole list (k = 8, chunk_rows = n / 8) and one rank's share of it (the first
    four chunks, as rank 0 of 2 decodes them), three calls in flight."""
    monkeypatch.setenv("RUHVRO_HIP_INTERNAL_STREAMS", internal_streams)
    n, offsets, exp, d_data, d_off = ten_million
    stream = torch.cuda.current_stream().cuda_stream
    cr = n // 8
    for rows, k in ((n, 8), (4 * cr, 4)):
        dl = int(offsets[rows])
        call = cabi.PreparedDeviceDecode(d_data.data_ptr(), d_off.data_ptr(), dl, rows, SCHEMAS["full"], k, device=0, stream=stream,
                                         kernel=cabi.KERNEL_SPECIALIZED, chunk_rows=cr, asynchronous=True)
        call.free(call.run(False))                            # (size history)
        hs = [call.run(False) for _ in range(3)]              # three calls on the stream, none settled
        for h in hs[:2]:
            call.wait(h)
            call.free(h)
        got = call.to_host(hs[2])                             # settles the third inside its first accessor
        call.free(hs[2])
        assert len(got) == k
        for g, e in zip(got, exp[:k]):
            assert_batches_identical(g, e)


def test_bench_line_through_a_one_rank_rccl_group():
    d = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "full1m", "--no-cpu-baseline", "--no-end-to-end",
                    "--no-other-configs"],
                   {"BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29631", "RANK": "0", "LOCAL_RANK": "0",
                    "WORLD_SIZE": "1"})
    assert d["dist"]["backend"] == "nccl" and d["dist"]["world_size"] == 1 and len(d["dist"]["collectives"]) == 3
    assert d["n_gpus"] == 1 and d["config"]["records_total"] == 1_000_000 and d["value"] > 0
    assert d["config"]["kernel_ms"]["k_emit"] > 0          # the stats vector went through the all-gather


def test_bench_line_carries_parity_evidence_for_its_timed_configuration():
    d = _run_bench(["--steps", "2", "--warmup", "1", "--workload", "full1m", "--no-end-to-end", "--no-other-configs"], {})
    pc = d["parity_check"]
    assert pc["result"] == "identical" and pc["records"] == 1_000_000 and pc["chunks"] == 8 and pc["buffers"] > 8 * 30
    assert d["cpu_baseline"]["kind"] == "port" and len(d["cpu_baseline"]["sample"]) < 160


def test_no_trust_knob_same_buffers():
    """RUHVRO_HIP_NO_TRUST=1: every tile of the emit pass is walked with its own checks (read once per process)."""
    env = dict(os.environ, RUHVRO_HIP_NO_TRUST="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_quick.py"), "200000"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "parity ok" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]


def test_poisoned_pools_both_directions():
    """RUHVRO_HIP_POISON=1: every pooled block (device arena, workspaces, pinned staging) is handed out filled with 0xA5, so
    a kernel that reads padding or a slot nobody wrote gets garbage instead of the zeros of a fresh allocation -- how the
    encode kernels' look-ahead read was found (profiles/r04zg_*).  The encode suite's nested / generated cases and the decode
    parity core run under it in a process of their own (the switch is read once)."""
    env = dict(os.environ, RUHVRO_HIP_POISON="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_encode.py", "-m", "gpu", "-q", "-x", "-k",
                        "nested or generated or random or round_trip"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2500:] + p.stderr[-1500:]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_quick.py"), "200000"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "parity ok" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]


def test_current_device_is_resolved_on_the_calling_thread():
    """rh_current_device() is the caller's device; a list above the streaming threshold (the engine then runs on a thread
    of its own, where HIP's current device would be 0 again) decodes on it."""
    assert cabi.lib().rh_current_device() == torch.cuda.current_device()
    if torch.cuda.device_count() < 2:
        pytest.skip("NOT RUN: ONE GPU VISIBLE.  The non-zero-device half of this test (a streaming call made with device 1 current must decode "
                    "on device 1 and leave device 0 untouched) needs two GPUs -- on a multi-GPU box this skip must not appear")
    data, offsets = fastgen.generate("full", 70_000)
    recs = fastgen.split(data, offsets)
    torch.cuda.set_device(1)
    try:
        assert cabi.lib().rh_current_device() == 1
        before = torch.cuda.memory_stats(0).get("num_alloc_retries", 0)       # (device 0 must stay untouched by this call)
        got, st = P.deserialize_array_threaded_with_stats(recs, SCHEMAS["full"], 4)
        exp = c_walker.decode_threaded(recs, SCHEMAS["full"], 4)
        for g, e in zip(got, exp):
            assert_batches_identical(g, e)
        assert torch.cuda.memory_stats(0).get("num_alloc_retries", 0) == before
    finally:
        torch.cuda.set_device(0)
