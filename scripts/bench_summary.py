"""Short human summary of a bench.py JSON line (what the GPU runner prints into the gpurun tail)."""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
c = d.get("config", {})
print("ms/step", round(d["ms_per_step"], 4), "value", f"{d['value']:.4g}", c.get("kernel_ms"), c.get("kernel_form"))
print("roofline", {k: (round(r[k], 4) if isinstance(r.get(k), float) else r.get(k)) for k in ("kernel", "frac", "path_frac", "traffic", "read_frac", "hbm_frac", "path_traffic")})
if "single_pass" in d:
    print("single_pass", {k: v for k, v in d["single_pass"].items() if k != "what"})
if "overlapped" in d:
    print("overlapped", round(d["overlapped"]["ms_per_step"], 4), "sync_call_ms", c.get("sync_call_ms"))
if "config5_projection" in d:
    print("proj", {g: (round(v["ms_per_step"], 4), round(v["implied_efficiency"], 3)) for g, v in d["config5_projection"]["g"].items()})
if "parity_check" in d:
    print("parity", {k: v for k, v in d["parity_check"].items() if k in ("result", "records", "chunks", "buffers")})
if "cpu_baseline" in d:
    cb = d["cpu_baseline"]
    print("cpu", f"{cb['value']:.4g}", cb["cores"], "reference:", cb.get("reference"), {k: (round(v["value"] / 1e6, 2), round(v["wall_ms"], 3)) for k, v in cb.get("at_metric_sizes", {}).items()})
if "other_configs" in d:
    # (the 10M lines of round 6 carry kernel_ms["path"] instead of a pipelined ms_per_step)
    print({k: (round(v.get("ms_per_step", v.get("kernel_ms", {}).get("path", 0)), 4), round(v.get("sync_call_ms", 0), 4), round(v.get("emit_frac", 0), 3),
               v.get("traffic"), (v.get("parity_check") or {}).get("result")) for k, v in d["other_configs"].items()})
if "end_to_end" in d:
    e = d["end_to_end"]
    print({k: (round(e[k]["value"] / 1e6, 1), round(e[k]["wall_ms"], 2)) for k in ("packed_pageable", "record_slices", "packed_8_logical_shards") if k in e})
    print({m: (round(v["value"] / 1e6, 1), round(v["wall_ms"], 3), v["gil_held_ms"], round(v["vs_record_slices"], 3), v["phase_ms"]) for m, v in e["python_list_bytes"].items()})
    print("config1 10k wall_ms", round(e["config1_python_10k"]["wall_ms"], 4))
if "cold_start" in d:
    print("cold_start", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["cold_start"].items() if k != "what"})
