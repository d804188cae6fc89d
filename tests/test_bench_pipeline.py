"""bench.py's Pipeline (the RH_ASYNC step loop): every submitted call is settled exactly once, in order, at most
PIPELINE_DEPTH + 1 are in flight, stats come back for the steps that asked for them, an error raised by a settle frees
the call, and several prepared calls are used round-robin.  A fake call object stands in for the C ABI (no GPU)."""
import pytest

import bench


class FakeStats:
    def __init__(self, d):
        self.d = d

    def as_dict(self):
        return dict(self.d)


class FakeCall:
    def __init__(self, name, fail_at=None):
        self.name, self.fail_at = name, fail_at
        self.next, self.in_flight, self.max_in_flight = 0, set(), 0
        self.waited, self.freed = [], []
        self.stats = FakeStats({})

    def run(self, want_stats):
        h = (self.name, self.next)
        self.next += 1
        self.in_flight.add(h)
        self.max_in_flight = max(self.max_in_flight, len(self.in_flight))
        return h

    def wait(self, h, want):
        assert h in self.in_flight
        self.waited.append(h)
        if self.fail_at is not None and h[1] == self.fail_at:
            raise ValueError("malformed record")
        if want:
            self.stats = FakeStats({"records": 7, "emit_kernel_ms": 1.0 + h[1]})

    def output_bytes(self, h):
        return 1234

    def free(self, h):
        self.in_flight.remove(h)
        self.freed.append(h)


def test_calls_are_settled_in_order_with_bounded_depth():
    c = FakeCall("a")
    info = {}
    p = bench.Pipeline(c, info)
    got = []
    for i in range(10):
        got += p.submit(i % 3 == 0)
    got += p.drain()
    assert c.waited == c.freed == [("a", i) for i in range(10)]
    assert c.max_in_flight == bench.PIPELINE_DEPTH + 1 and not c.in_flight
    assert [g["emit_kernel_ms"] for g in got] == [1.0, 4.0, 7.0, 10.0] and info["output_bytes"] == 1234


def test_round_robin_over_streams_and_error_frees_the_call():
    a, b = FakeCall("a"), FakeCall("b", fail_at=1)
    p = bench.Pipeline([a, b], {})
    for _ in range(3):
        p.submit(False)
    assert a.next == 2 and b.next == 1
    p.submit(False)                       # b's second call (index 1) is submitted ...
    with pytest.raises(ValueError):
        p.drain()                         # ... and fails when it is settled
    assert ("b", 1) in b.freed            # freed although its wait raised
