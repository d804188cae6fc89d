"""Every schema whose specialised kernels should be in the kernel cache before a GPU run: the benchmark schemas plus
every schema the parity tests decode.  Lives outside the product package because it imports the test suite's case
tables (which use the oracle's schema model): build() and tests/conftest.py call it, pyruhvro_amd never does.

    python scripts/known_schemas.py        # prebuild all of them (hiprtc, gfx950; needs no GPU)
"""
import os
import sys
from typing import List

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def WIDE_SCHEMAS() -> List[str]:
    """The wide schemas (more than 64 scanned counters) once the engine takes them (round 6); [] while it refuses them."""
    from avrogen.schemas import SCHEMAS
    from pyruhvro_amd import cabi
    out = []
    for k, v in SCHEMAS.items():
        if k.startswith("wide"):
            try:
                cabi.Schema.get(v)
                out.append(v)
            except ValueError:
                pass
    return out


def known_schemas() -> List[str]:
    """Benchmark schemas + every schema the parity tests decode."""
    root = ROOT
    from avrogen.schemas import SCHEMAS
    out = [v for k, v in SCHEMAS.items() if not k.startswith("wide")] + WIDE_SCHEMAS()
    try:
        import json
        import cases
        for c in cases.wire_cases() + cases.nesting_cases() + cases.dense_list_cases() + cases.enum_form_cases() + cases.wide_counter_cases():
            out.append(c[1])
        for c in cases.wide_form_cases() + cases.giant_record_cases() + cases.deep_nesting_cases():
            out.append(c[1])
        for c in cases.error_cases():
            out.append(c[1])
        for c in cases.differential_cases():
            out.append(c[1])
        out.append(cases.logical_case()[0])
        out += cases.encode_extra_schemas()
        out.append(cases.long_string_case(1)[0])
        import random_cases
        out += [random_cases.random_schema(seed) for seed in range(random_cases.PREBUILT_SEEDS)]
        import test_n4_types                      # SURVEY 8f N4 schemas
        out.append(test_n4_types.SCHEMA)
        out.append(test_n4_types.DUR_SCHEMA)
        import test_named_refs                    # named-type references (resolved by the front-end)
        out.append(test_named_refs.WITH_REFS)
        g = json.load(open(os.path.join(root, "tests", "golden", "reference_vectors.json")))
        out += [json.dumps(s) for s in g["schemas"].values()]
    except ImportError:
        raise
    return out




def single_pass_schemas() -> List[str]:
    """The schemas whose single-pass kernel (rh_spec_fused) a bench line or a test runs: the only ones build() compiles it for --
    it is the most expensive of a schema's five kernels and opt-in; every other schema gets it on first request."""
    from avrogen.schemas import SCHEMAS
    return [SCHEMAS[k] for k in ("full", "cfg3", "flat4", "array_and_map", "nullable_primitives", "t_enum", "t_union")]


if __name__ == "__main__":
    from pyruhvro_amd.prebuild import prebuild_many
    with_fused = list(dict.fromkeys(single_pass_schemas()))
    rest = [s for s in dict.fromkeys(known_schemas()) if s not in set(with_fused)]
    rest.sort(key=len, reverse=True)            # (the widest schemas first: one of their kernels alone takes hiprtc minutes)
    errs = prebuild_many(with_fused, verbose=True)
    os.environ["RUHVRO_HIP_PREBUILD_FUSED"] = "0"
    errs += prebuild_many(rest, verbose=True)
    if not errs:
        from pyruhvro_amd.prebuild import mark_warm
        mark_warm(known_schemas())
    for e in errs:
        print(e[:2000])
    sys.exit(1 if errs else 0)
