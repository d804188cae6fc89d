"""Seeded synthetic record generators (Python spec; ``fastgen.c`` emits the same bytes).

Distributions follow the reference's own generators:
  * FULL   -- scripts/generate_avro.py:44-62 (Faker replaced by seeded letters
              of the same length ranges; SURVEY.md section 8d lists them);
  * FLAT4  -- ruhvro/benches/common/mod.rs:52-63 minus ``f`` and ``s``;
  * CFG3   -- BASELINE.json config 3 (string + nullable-union + enum);
and the reference's bench builders (common/mod.rs:52-165) for the other names.

Every record is a pure function of (seed, row) so any shard of any size can be
generated independently (multi-GPU weak scaling uses row offsets per rank).
"""
from __future__ import annotations

from typing import Callable, Dict, List

M64 = 0xFFFFFFFFFFFFFFFF


def _mix(z: int) -> int:
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


class Rng:
    """splitmix64 stream keyed by (seed, row)."""

    def __init__(self, seed: int, row: int):
        self.s = _mix((seed ^ ((row * 0xD1342543DE82EF95) & M64)) & M64)

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        return _mix(self.s)

    def below(self, n: int) -> int:
        return self.next() % n

    def between(self, lo: int, hi: int) -> int:  # inclusive
        return lo + self.next() % (hi - lo + 1)

    def letters(self, n: int, base: str = "a", span: int = 26) -> str:
        out = []
        r = 0
        for j in range(n):
            if j % 8 == 0:
                r = self.next()
            out.append(chr(ord(base) + ((r >> (8 * (j % 8))) & 0xFF) % span))
        return "".join(out)


def gen_full(seed: int, row: int) -> dict:
    g = Rng(seed, row)
    name = g.letters(g.between(9, 18)) if g.below(2) else None
    age = g.between(18, 80) if g.below(2) else None
    emails = [g.letters(g.between(17, 28)) for _ in range(g.below(4))]
    address = None
    if g.below(2):
        address = {"street": g.letters(g.between(14, 30)), "city": g.letters(g.between(8, 18)),
                   "zipcode": g.letters(5, "0", 10)}
    phones = [(g.letters(g.between(3, 9)), g.letters(g.between(10, 22), "0", 10)) for _ in range(g.below(4))]
    prefs = None
    if g.below(2):
        prefs = {"contact_method": [None, "email", "phone"][g.below(3)], "newsletter": bool(g.below(2))}
    sk = g.below(4)
    if sk == 0:
        status = None
    elif sk == 1:
        status = g.letters(g.between(3, 9))
    elif sk == 2:
        status = g.between(0, 100)
    else:
        status = bool(g.below(2))
    created = 1_726_000_000 + g.below(31_536_000)
    cls = "ABC"[g.below(3)]
    return {"name": name, "age": age, "emails": emails, "address": address, "phone_numbers": phones,
            "preferences": prefs, "status": status, "created_at": created, "class": cls}


def gen_flat4(seed: int, row: int) -> dict:
    return {"i": row, "l": row * 7, "d": row * 2.25, "b": row % 2 == 0}


def gen_cfg3(seed: int, row: int) -> dict:
    g = Rng(seed, row)
    name = g.letters(g.between(8, 19)) if g.below(2) else None
    age = g.between(18, 80) if g.below(2) else None
    return {"id": row * 7, "name": name, "age": age, "s": f"row-{row}", "class": "ABC"[row % 3]}


# ---- the reference's bench builders (ruhvro/benches/common/mod.rs) ----------
def gen_flat_primitives(seed: int, i: int) -> dict:          # mod.rs:52-63
    import struct
    f32 = struct.unpack("<f", struct.pack("<f", i * 1.5))[0]
    return {"i": i, "l": i * 7, "f": f32, "d": i * 2.25, "b": i % 2 == 0, "s": f"row-{i}"}


def gen_nullable_primitives(seed: int, i: int) -> dict:      # mod.rs:83-98
    if i % 2 == 1:
        return {"i": None, "l": None, "d": None, "b": None, "s": None}
    return {"i": i, "l": i * 7, "d": i * 2.25, "b": i % 4 == 0, "s": f"row-{i}"}


def gen_nested_struct(seed: int, i: int) -> dict:            # mod.rs:119-131
    return {"outer_id": i, "inner": {"x": i, "y": i * 3, "label": f"lbl-{i}"}}


def gen_array_and_map(seed: int, i: int) -> dict:            # mod.rs:148-165
    return {"id": i, "tags": [f"t-{i}-a", f"t-{i}-b", f"t-{i}-c"],
            "props": [(f"k{i}-1", f"v{i}-1"), (f"k{i}-2", f"v{i}-2")]}


GENERATORS: Dict[str, Callable[[int, int], dict]] = {
    "full": gen_full, "flat4": gen_flat4, "cfg3": gen_cfg3,
    "flat_primitives": gen_flat_primitives, "nullable_primitives": gen_nullable_primitives,
    "nested_struct": gen_nested_struct, "array_and_map": gen_array_and_map,
}


def records(name: str, n: int, seed: int = 20260921, start: int = 0) -> List[bytes]:
    """Encode rows [start, start+n) of the named configuration."""
    from oracle.avro_schema import parse_schema
    from .encoder import to_datum
    from .schemas import SCHEMAS
    s = parse_schema(SCHEMAS[name])
    gen = GENERATORS[name]
    return [to_datum(s, gen(seed, start + i)) for i in range(n)]
