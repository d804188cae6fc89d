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


# ---- round 6: the full schema off its friendly value distribution --------------------
BIG_ARRAY_PER_MILLION = {"full_realistic": 100, "full_realistic_heavy": 10_000, "full_realistic_nogiant": 0}


def _gen_realistic(seed: int, row: int, big_per_million: int) -> dict:
    """gen_full's columns + what production adds: ~1 % of the records carry a note of 8 KiB and more, `big_per_million` of
    them an `emails` array of more than 8,191 items (both outside the two-byte length / count form of the fast walk), every
    record a nullable timestamp-micros of today (8-byte varint), an int of epoch seconds (5 bytes), a snowflake id (9 bytes)."""
    g = Rng(seed, row)
    name = g.letters(g.between(9, 18)) if g.below(2) else None
    age = g.between(18, 80) if g.below(2) else None
    ne = g.between(8192, 9000) if g.below(1_000_000) < big_per_million else g.below(4)
    emails = [g.letters(g.between(17, 28)) for _ in range(ne)]
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
    created = None if g.below(20) == 0 else 1_750_000_000_000_000 + g.below(31_536_000_000_000)
    cls = "ABC"[g.below(3)]
    login = 1_750_000_000 + g.below(31_536_000)
    event_id = ((461_165_025_343 + g.below(31_536_000_000)) << 22) | (g.below(1024) << 12) | g.below(4096)
    r = g.below(100)
    note = g.letters(g.between(8192, 9215)) if r == 0 else g.letters(g.between(20, 60)) if r < 10 else None
    return {"name": name, "age": age, "emails": emails, "address": address, "phone_numbers": phones,
            "preferences": prefs, "status": status, "created_at": created, "class": cls,
            "login_ts": login, "event_id": event_id, "note": note}


def gen_full_realistic(seed: int, row: int) -> dict:
    return _gen_realistic(seed, row, BIG_ARRAY_PER_MILLION["full_realistic"])


def gen_full_realistic_heavy(seed: int, row: int) -> dict:
    return _gen_realistic(seed, row, BIG_ARRAY_PER_MILLION["full_realistic_heavy"])


def gen_full_realistic_nogiant(seed: int, row: int) -> dict:
    return _gen_realistic(seed, row, 0)


_SKEW_REC = ((512, 1), (768, 2), (896, 3), (960, 4), (992, 6), (1008, 8), (1016, 12), (1020, 16), (1022, 24), (1023, 32), (1024, 48))
_SKEW_RUN = ((600, 1), (800, 2), (900, 3), (960, 5), (1000, 8))
SKEW_RUN_ROWS = 97


def _pick(table, r: int) -> int:
    for lim, v in table:
        if r < lim:
            return v
    return table[-1][1]


def skew_scale(seed: int, row: int) -> int:
    """String-length multiplier of a record of `full_skewed`: a heavy-tailed per-record factor times a factor shared by runs of
    SKEW_RUN_ROWS consecutive records (a tenant with large payloads), so that whole tiles outgrow a window sized for the mean."""
    run = Rng(seed ^ 0x5EED5CA1E, row // SKEW_RUN_ROWS)
    rec = Rng(seed ^ 0x0DDBA11, row)
    return _pick(_SKEW_RUN, run.below(1000)) * _pick(_SKEW_REC, rec.below(1024))


def gen_full_skewed(seed: int, row: int) -> dict:
    """gen_full with every generated string `skew_scale` times as long (record sizes roughly log-normal, correlated in runs)."""
    m = skew_scale(seed, row)
    g = Rng(seed, row)
    name = g.letters(m * g.between(9, 18)) if g.below(2) else None
    age = g.between(18, 80) if g.below(2) else None
    emails = [g.letters(m * g.between(17, 28)) for _ in range(g.below(4))]
    address = None
    if g.below(2):
        address = {"street": g.letters(m * g.between(14, 30)), "city": g.letters(m * g.between(8, 18)),
                   "zipcode": g.letters(5, "0", 10)}
    phones = [(g.letters(g.between(3, 9)), g.letters(m * g.between(10, 22), "0", 10)) for _ in range(g.below(4))]
    prefs = None
    if g.below(2):
        prefs = {"contact_method": [None, "email", "phone"][g.below(3)], "newsletter": bool(g.below(2))}
    sk = g.below(4)
    if sk == 0:
        status = None
    elif sk == 1:
        status = g.letters(m * g.between(3, 9))
    elif sk == 2:
        status = g.between(0, 100)
    else:
        status = bool(g.below(2))
    created = 1_726_000_000 + g.below(31_536_000)
    cls = "ABC"[g.below(3)]
    return {"name": name, "age": age, "emails": emails, "address": address, "phone_numbers": phones,
            "preferences": prefs, "status": status, "created_at": created, "class": cls}


def gen_wide(ncols: int):
    """Rows of schemas.wide_schema(ncols): every string column null with p = 1/2, else 4-16 letters; arrays of 0-3 strings."""
    from .schemas import WIDE_ARRAYS
    every = max(ncols // WIDE_ARRAYS, 1)

    def gen(seed: int, row: int) -> dict:
        g = Rng(seed, row)
        v = {}
        na = 0
        for i in range(ncols):
            v[f"c{i}"] = g.letters(g.between(4, 16)) if g.below(2) else None
            if (i + 1) % every == 0 and na < WIDE_ARRAYS:
                v[f"a{na}"] = [g.letters(g.between(3, 12)) for _ in range(g.below(4))]
                na += 1
        while na < WIDE_ARRAYS:
            v[f"a{na}"] = [g.letters(g.between(3, 12)) for _ in range(g.below(4))]
            na += 1
        return v
    return gen


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
    "full_realistic": gen_full_realistic, "full_realistic_heavy": gen_full_realistic_heavy, "full_realistic_nogiant": gen_full_realistic_nogiant, "full_skewed": gen_full_skewed,
    "wide97": gen_wide(97), "wide200": gen_wide(200), "wide400": gen_wide(400),
}


def records(name: str, n: int, seed: int = 20260921, start: int = 0) -> List[bytes]:
    """Encode rows [start, start+n) of the named configuration."""
    from oracle.avro_schema import parse_schema
    from .encoder import to_datum
    from .schemas import SCHEMAS
    s = parse_schema(SCHEMAS[name])
    gen = GENERATORS[name]
    return [to_datum(s, gen(seed, start + i)) for i in range(n)]
