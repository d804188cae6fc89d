"""Quick GPU-vs-oracle buffer identity on the full schema (300k records, 8 chunks, specialised kernels) -- used by the
A/B scripts to reject a variant that changes results before its timing is looked at."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import pyruhvro_amd as P
from arrow_compare import assert_batches_identical
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from oracle import c_walker
from pyruhvro_amd import cabi
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
for cfg in ("full",):
    data, offsets = fastgen.generate(cfg, n)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS[cfg]), data, offsets, 8, threaded=True)
    got = cabi.decode_packed(data, offsets, SCHEMAS[cfg], 8, kernel=2)
    for g, e in zip(got, exp):
        assert_batches_identical(g, e)
print("parity ok")
