#!/usr/bin/env python3
"""Timeline of the LAST call(s) in a rocprofv3 rocpd SQLite result: kernels and memory copies in start order, with the
idle gap before each.   python scripts/rocpd_timeline.py <results.db> [n_last=16]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rows = []
    try:
        rows += [(r[1], r[2], "K " + r[0]) for r in c.execute("select name, start, end from kernels")]
    except sqlite3.Error as e:
        print("no kernels view:", e)
    for tbl in ("memory_copies", "memory_copy"):
        try:
            rows += [(r[1], r[2], "M " + str(r[0])) for r in c.execute(f"select name, start, end from {tbl}")]
            break
        except sqlite3.Error:
            continue
    rows.sort()
    rows = rows[-n:]
    prev = None
    t0 = rows[0][0] if rows else 0
    for s, e, name in rows:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        print(f"{(s - t0) / 1e3:10.2f} us  +{(e - s) / 1e3:8.2f} us  gap {gap:8.2f} us  {name[:60]}")
        prev = e


if __name__ == "__main__":
    main()
