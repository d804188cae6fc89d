"""Which columns / rows of the giant_arrays case differ from the oracle (debugging aid)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa
import numpy as np
import pyarrow as pa
import cases
from oracle import c_walker
import pyruhvro_amd as P
from pyruhvro_amd import cabi
name, schema, recs = cases.giant_record_cases()[0]
if os.environ.get("GD_TAIL"): recs = recs[256:]
P.set_kernel_mode("specialized")
K = int(os.environ.get("GD_K", "1")); CH = int(os.environ.get("GD_CH", "0"))
got = P.deserialize_array_threaded(recs, schema, K)[CH]
exp = c_walker.decode_threaded(recs, schema, K)[CH]
N = len(exp)
for i, f in enumerate(exp.schema):
    g, e = got.column(i), exp.column(i)
    try:
        same = g.equals(e)
    except Exception as ex:
        same = f"ERR {str(ex)[:80]}"
    msg = ""
    if same is not True and pa.types.is_list(f.type) or pa.types.is_map(f.type):
        go = np.frombuffer(g.buffers()[1], np.int32, len(e) + 1); eo = np.frombuffer(e.buffers()[1], np.int32, len(e) + 1)
        d = np.flatnonzero(go != eo)
        msg = f"list offsets differ at rows {d[:5]} (n={len(d)})" if len(d) else "list offsets equal"
        gv, ev = (g.values, e.values) if pa.types.is_list(f.type) else (g.items, e.items)
        if pa.types.is_list(ev.type):
            g2 = np.frombuffer(gv.buffers()[1], np.int32, len(ev) + 1); e2 = np.frombuffer(ev.buffers()[1], np.int32, len(ev) + 1)
            d2 = np.flatnonzero(g2 != e2)
            msg += f"; inner offsets differ first at {d2[:3]} (n={len(d2)}), got {g2[d2[:3]]} exp {e2[d2[:3]]}"
            if len(d2):
                a = int(d2[0]); msg += f"\n   got  {g2[a-4:a+12].tolist()}\n   exp  {e2[a-4:a+12].tolist()}\n   tail got {g2[-6:].tolist()} exp {e2[-6:].tolist()}"
                gi = np.frombuffer(gv.values.buffers()[1], np.int32, len(ev.values)); ei = np.frombuffer(ev.values.buffers()[1], np.int32, len(ev.values))
                dv = np.flatnonzero(gi != ei); msg += f"\n   inner values differ n={len(dv)} first {dv[:4].tolist()} got {gi[dv[:8]].tolist()} exp {ei[dv[:8]].tolist()}"
        else:
            try:
                neq = 0 if gv.equals(ev) else 1
                if neq:
                    gl, el = gv.to_pylist()[:200000], ev.to_pylist()[:200000]
                    first = next((j for j in range(min(len(gl), len(el))) if gl[j] != el[j]), -1)
                    msg += f"; values differ first at item {first}: got {gl[first] if first >= 0 else None} exp {el[first] if first >= 0 else None} (len {len(gv)} vs {len(ev)})"
            except Exception as ex:
                msg += f"; values ERR {str(ex)[:80]}"
    print(f.name, same, msg)
# which records are giants, where do their items start
po = np.frombuffer(got.column(7).buffers()[1], np.int32, N + 1); pe = np.frombuffer(exp.column(7).buffers()[1], np.int32, N + 1)
gd = np.frombuffer(got.column(7).buffers()[2], np.uint8); ed = np.frombuffer(exp.column(7).buffers()[2], np.uint8)
do = np.flatnonzero(po != pe); print("post offsets differ rows", do[:8].tolist(), "n", len(do), "data len", len(gd), len(ed))
m_ = min(len(gd), len(ed)); dd = np.flatnonzero(gd[:m_] != ed[:m_]); print("post data differ first", dd[:6].tolist(), "last", dd[-3:].tolist(), "n", len(dd), "got", bytes(gd[dd[:12]]) if len(dd) else b"", "row of first", int(np.searchsorted(pe, dd[0], side="right") - 1) if len(dd) else -1, "row start", int(pe[np.searchsorted(pe, dd[0], side="right") - 1]) if len(dd) else -1)
print("post offsets tail got", po[-4:].tolist(), "exp", pe[-4:].tolist())
print("counters", {k: v for k, v in cabi.engine_counters().items() if v})
print("giants at rows", [i for i, r in enumerate(recs) if len(r) > 30000], "lens", [len(r) for r in recs if len(r) > 30000])
