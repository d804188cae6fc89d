#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (ROCm 7 default output) as text:
per-kernel stats (count / total / mean / min / max duration, VGPR/SGPR/LDS) and, when a PMC pass
was collected, per-kernel mean counter values.   python scripts/rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def summarise(path: str) -> str:
    c = sqlite3.connect(path)
    out = [f"# {path}"]
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out.append(f"{'kernel':40s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s} "
               f"{'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scr':>5s} {'grid':>10s} {'wg':>4s}")
    for r in rows:
        out.append(f"{r[0][:40]:40s} {r[1]:6d} {r[2]/1e3:12.1f} {r[3]/1e3:10.2f} {r[4]/1e3:10.2f} {r[5]/1e3:10.2f} {100*r[2]/tot:6.2f} "
                   f"{r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:5d} {r[10]:10d} {r[11]:4d}")
    try:
        pmc = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                        "group by kernel_name, counter_name order by 1, 2").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        out.append("")
        out.append(f"{'kernel':40s} {'counter':24s} {'n':>5s} {'mean':>16s} {'min':>16s} {'max':>16s}")
        for r in pmc:
            out.append(f"{r[0][:40]:40s} {r[1][:24]:24s} {r[2]:5d} {r[3]:16.1f} {r[4]:16.1f} {r[5]:16.1f}")
    return "\n".join(out)


def traffic(fetch_db: str, write_db: str) -> dict:
    """Per-kernel mean FETCH_SIZE / WRITE_SIZE (KiB per dispatch) -> bytes, with the gfx950 correction of
    MI355X_MICROARCH.md (FETCH_SIZE under-reports wide coalesced reads by 2x)."""
    out = {}
    for key, db in (("FETCH_SIZE", fetch_db), ("WRITE_SIZE", write_db)):
        c = sqlite3.connect(db)
        for name, val, dur in c.execute("select kernel_name, avg(value), avg(duration) from counters_collection "
                                        "where counter_name = ? group by kernel_name", (key,)):
            d = out.setdefault(name, {})
            d[key + "_KiB"] = val
            d.setdefault("avg_duration_us", dur / 1e3)
    for name, d in out.items():
        rd = d.get("FETCH_SIZE_KiB", 0.0) * 1024 * 2      # x2: gfx950 correction
        wr = d.get("WRITE_SIZE_KiB", 0.0) * 1024
        d["hbm_read_bytes"] = rd
        d["hbm_write_bytes"] = wr
        d["hbm_bytes"] = rd + wr
    return out


if __name__ == "__main__":
    if len(sys.argv) in (4, 5) and sys.argv[1] == "--traffic-json":
        # --traffic-json <fetch.db> <write.db> [kernel_key]: the key (pyruhvro_amd.cabi.kernel_key of the schema the
        # passes decoded) stamps the file; bench.py only reports the traffic when the kernels it runs carry that key
        import json
        d = traffic(sys.argv[2], sys.argv[3])
        if len(sys.argv) == 5:
            d["_kernel_key"] = sys.argv[4]
        print(json.dumps(d, indent=1))
    else:
        for p in sys.argv[1:]:
            print(summarise(p))
            print()
