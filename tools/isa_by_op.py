#!/usr/bin/env python3
"""Static VALU / SALU / DS / VMEM instructions of a specialised kernel per GENERATED-SOURCE line (= per schema op instance),
from the inlined-at chains of the .loc comments in an annotated .s (tools/isa_hist.py --keep-asm).

    python tools/isa_by_op.py scratch/emit.s rh_spec_emit
"""
import collections
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    src = open(path + ".hip").read().split("\n")
    agg = collections.defaultdict(lambda: [0, 0, 0, 0])
    inside, cur = False, "?"
    for l in open(path):
        if l.startswith(kern + ":"):
            inside = True
            continue
        if inside and re.match(r"\.Lfunc_end|\s*\.size\s+" + kern, l):
            break
        if not inside:
            continue
        if re.match(r"\s*\.loc\s", l):
            m = re.findall(r"k\.hip:(\d+):", l)
            head = re.search(r";\s*\S*/([\w.]+):(\d+)", l)
            # innermost k.hip line that is inside Spec::walk (the op), else the outermost frame's file
            ops = [int(x) for x in m]
            cur = ("op", ops[0]) if ops and len(ops) > 1 or (ops and "walk" in src[ops[0] - 1] is False) else None
            if ops:
                cur = ("k", ops[0])
            else:
                cur = ("f", head.group(1) + ":" + head.group(2) if head else "?")
            continue
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        a = agg[cur]
        a[0 if t.startswith("v_") else 1 if t.startswith("s_") else 2 if t.startswith("ds_") else 3] += 1
    tot = [0, 0, 0, 0]
    rows = []
    for k, v in agg.items():
        for i in range(4):
            tot[i] += v[i]
        if k and k[0] == "k":
            rows.append((k[1], v, src[k[1] - 1].strip()[:110]))
    other = [0, 0, 0, 0]
    for k, v in agg.items():
        if not (k and k[0] == "k"):
            for i in range(4):
                other[i] += v[i]
    print("%5s %5s %5s %4s %4s  %s" % ("line", "VALU", "SALU", "DS", "VMEM", "generated source"))
    for ln, v, text in sorted(rows):
        print("%5d %5d %5d %4d %4d  %s" % (ln, *v, text))
    print("other %5d %5d %4d %4d  (no generated-source frame)" % tuple(other))
    print("TOTAL %5d %5d %4d %4d" % tuple(tot))


if __name__ == "__main__":
    main()
