"""Seeded random (schema, records) pairs inside the reference's direct-decode subset
(fast_decode::is_supported, ruhvro/src/fast_decode.rs:38-61): nested records, enums, arrays, maps,
2-variant null unions in both orders and N-variant unions, with values that exercise every wire form the
encoder can produce (multi-block / negative-count blocks, empty containers, long strings, extreme ints)."""
from __future__ import annotations

import json
import random
import struct
from typing import List, Tuple

from avrogen.encoder import Blocks, Branch, to_datum
from oracle.avro_schema import AvroSchema, SchemaError, build_tree, parse_schema

PRIMS = ["int", "long", "float", "double", "boolean", "string",
         {"type": "int", "logicalType": "date"}, {"type": "long", "logicalType": "timestamp-millis"},
         {"type": "long", "logicalType": "timestamp-micros"}]


PREBUILT_SEEDS = 40      # seeds whose specialised kernels build() compiles ahead of the GPU run (scripts/known_schemas.py)


def _rand_type(r: random.Random, depth: int, counter: List[int]):
    def named(kind):
        counter[0] += 1
        return f"{kind}{counter[0]}"
    if depth <= 0 or r.random() < 0.35:
        return r.choice(PRIMS)
    k = r.randrange(4)
    if k == 0:
        return {"type": "record", "name": named("R"),
                "fields": [{"name": f"f{i}", "type": _wrap(r, _rand_type(r, depth - 1, counter))} for i in range(r.randint(1, 3))]}
    if k == 1:
        return {"type": "enum", "name": named("E"), "symbols": r.sample(["A", "B", "CC", "DDD", "eeeee"], r.randint(1, 4))}
    if k == 2:
        return {"type": "array", "items": _wrap(r, _rand_type(r, depth - 1, counter))}
    return {"type": "map", "values": _wrap(r, _rand_type(r, depth - 1, counter), allow_nvariant=False)}


def _wrap(r: random.Random, t, allow_nvariant: bool = True):
    is_map = isinstance(t, dict) and t.get("type") == "map"
    m = r.randrange(5)
    if is_map or m == 0 or m == 4:
        return t
    if m == 1:
        return ["null", t]
    if m == 2:
        return [t, "null"]
    if not allow_nvariant:
        return t
    other = "boolean" if t != "boolean" else "int"
    return ["null", t, other]


def random_schema(seed: int) -> str:
    """A schema the oracle accepts (schemas the reference cannot translate are re-drawn)."""
    r = random.Random(seed)
    for _ in range(200):
        counter = [0]
        fields = [{"name": f"c{i}", "type": _wrap(r, _rand_type(r, 2, counter))} for i in range(r.randint(1, 5))]
        js = json.dumps({"type": "record", "name": "Top", "fields": fields})
        try:
            build_tree(parse_schema(js))
            return js
        except (SchemaError, ValueError):
            continue
    raise RuntimeError("no supported schema drawn")


def _rand_value(r: random.Random, s: AvroSchema):
    k = s.kind
    if k == "null":
        return None
    if k == "boolean":
        return r.random() < 0.5
    if k in ("int", "date"):
        return r.choice([0, 1, -1, 63, 64, -65, 2**31 - 1, -2**31, r.randint(-10**6, 10**6)])
    if k in ("long", "timestamp-millis", "timestamp-micros"):
        return r.choice([0, -1, 2**62, -2**63, 2**63 - 1, r.randint(-10**12, 10**12), 1_700_000_000_000])
    if k == "float":
        return struct.unpack("<f", struct.pack("<f", r.uniform(-1e6, 1e6)))[0]
    if k == "double":
        return r.choice([0.0, -0.0, float("inf"), r.uniform(-1e12, 1e12)])
    if k == "string":
        n = r.choice([0, 1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 31, 40, 200]) if r.random() < 0.8 else r.randint(0, 1500)
        return "".join(r.choice("abcxyz0189 _é") for _ in range(n))
    if k == "enum":
        return r.choice(s.symbols)
    if k == "record":
        return {f.name: _rand_value(r, f.schema) for f in s.fields}
    if k == "union":
        i = r.randrange(len(s.variants))
        return Branch(i, _rand_value(r, s.variants[i]))
    if k in ("array", "map"):
        n = r.choice([0, 0, 1, 2, 3, 5, 70]) if r.random() < 0.9 else r.randint(0, 200)
        if k == "array":
            items = [_rand_value(r, s.items) for _ in range(n)]
        else:
            items = [("k%d" % j * r.randint(1, 3), _rand_value(r, s.items)) for j in range(n)]
        if n and r.random() < 0.4:       # split into blocks, some with the negative-count + byte-size form
            cut = r.randint(0, n)
            return Blocks([(items[:cut], r.random() < 0.5), (items[cut:], r.random() < 0.5)] if 0 < cut < n
                          else [(items, True)])
        return items
    raise ValueError(k)


def random_case(seed: int, nrec: int) -> Tuple[str, List[bytes]]:
    js = random_schema(seed)
    s = parse_schema(js)
    r = random.Random(seed * 7919 + 1)
    recs = [to_datum(s, _rand_value(r, s)) for _ in range(nrec)]
    if recs and r.random() < 0.5:
        recs[r.randrange(len(recs))] += b"\x01\x02\x03"      # trailing bytes are ignored (fast_decode.rs:825-828)
    return js, recs
