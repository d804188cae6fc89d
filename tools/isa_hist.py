#!/usr/bin/env python3
"""Static instruction budget of a schema-specialised kernel, per source function (needs only hipcc, no GPU).

    python tools/isa_hist.py [--schema full] [--kernel emit|size|eemit|esize] [--fast-only] [--lines N]

Generates the kernel source of the schema (rh_schema_kernel_source / rh_schema_encode_kernel_source), compiles it
for gfx950 with -gline-tables-only -S, and attributes every instruction of the kernel to the innermost inlined
source line.  One lane = one record and the walk has no per-record loop, so apart from list loops and skipped
branches the static count of the fast walk IS the per-wave dynamic count -- which is what the emit kernel is bound
by (VALU/SALU issue, not HBM).  --fast-only compiles the decode emit kernel with the careful
walk, the re-size walk and the global-memory walk removed, i.e. just the path 98 % of the tiles take.
"""
import argparse
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "pyruhvro_amd", "csrc")
HEADERS = ["program.h", "walk.h", "kernel_common.h", "spec_body.h", "encode.h", "encode_walk.h", "encode_spec.h"]
KERNELS = {"emit": "rh_spec_emit", "fused": "rh_spec_fused", "size": "rh_spec_size", "eemit": "rh_espec_emit", "esize": "rh_espec_size"}


def function_ranges(path):
    """[(first_line, last_line, name)] of the top-level functions / structs of a header (next definition ends one)."""
    starts = []
    for i, l in enumerate(open(path).read().split("\n"), 1):
        m = re.match(r"(?:__device__|__host__|template|struct|inline)", l)
        if m and not l.startswith("template"):
            name = re.search(r"(\w+)\s*(?:\(|\{|$)", l.replace("__forceinline__", "").replace("__device__", "")
                             .replace("__host__", "").replace("inline", "").replace("constexpr", "").strip())
            m2 = re.search(r"(?:struct\s+(\w+))|(\w+)\s*\(", l)
            starts.append((i, (m2.group(1) or m2.group(2)) if m2 else (name.group(1) if name else "?")))
    out = []
    for j, (ln, nm) in enumerate(starts):
        end = starts[j + 1][0] - 1 if j + 1 < len(starts) else 10 ** 9
        out.append((ln, end, nm))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--schema", default="full")
    ap.add_argument("--kernel", default="emit", choices=sorted(KERNELS))
    ap.add_argument("--fast-only", action="store_true")
    ap.add_argument("--lines", type=int, default=0, help="also print the N hottest source lines")
    ap.add_argument("--opcode", default="", help="also print the source lines that emit this opcode (e.g. v_lshl_add_u64)")
    ap.add_argument("--keep-asm", default="", help="copy the kernel's annotated assembly (.s) to this path")
    ap.add_argument("--variant", default="", help="RUHVRO_HIP_VARIANT names (staged experimental code paths), comma separated")
    args = ap.parse_args()
    if args.variant:
        os.environ["RUHVRO_HIP_VARIANT"] = args.variant
    from avrogen.schemas import SCHEMAS
    from pyruhvro_amd import cabi
    schema = SCHEMAS.get(args.schema) or open(args.schema).read()
    src = cabi.encode_kernel_source(schema) if args.kernel in ("eemit", "esize") else cabi.kernel_source(schema)
    tmp = tempfile.mkdtemp(prefix="isa_hist_")
    try:
        for h in HEADERS:
            shutil.copy(os.path.join(CSRC, h), tmp)
        if args.fast_only:
            p = os.path.join(tmp, "spec_body.h")
            s = open(p).read()
            s = s.replace("if (rewalk) {", "if (false) {")
            s = s.replace("  if (__any(L.redo)) {   //", "  if (false) {   //")      # k_size: no careful re-walk of the wave
            s = s.replace("  if (careful) {\n    spec_run_walk<S, true, true>", "  if (false) {\n    spec_run_walk<S, true, true>")
            s, n = re.subn(r"  \} else \{\n    GlobalSrc src\{[^\n]*\n    S::template walk<EMIT, true>\(c, src, L\);\n  \}",
                           "  }", s)
            assert n == 1, "spec_run_walk changed: update tools/isa_hist.py"
            s = s.replace("  if (fits) {\n    LdsSrc src{win};", "  {\n    LdsSrc src{win};")
            open(p, "w").write(s)
        open(os.path.join(tmp, "k.hip"), "w").write("#include <hip/hip_runtime.h>\n" + src)
        asm = os.path.join(tmp, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-gline-tables-only", "-I", tmp,
                               "--cuda-device-only", "-S", os.path.join(tmp, "k.hip"), "-o", asm])
        lines = open(asm).read().split("\n")
        if args.keep_asm:
            shutil.copy(asm, args.keep_asm)
            shutil.copy(os.path.join(tmp, "k.hip"), args.keep_asm + ".hip")
        ranges = {h: function_ranges(os.path.join(tmp, h)) for h in HEADERS}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    kern = KERNELS[args.kernel]
    inside, cur = False, ("?", 0)
    agg = collections.defaultdict(lambda: [0, 0, 0, 0, 0])
    by_line = collections.Counter()
    op_lines = collections.Counter()
    ops = collections.Counter()
    for l in lines:
        if l.startswith(kern + ":"):
            inside = True
            continue
        if inside and re.match(r"\.Lfunc_end|\s*\.size\s+" + kern, l):      # (a kernel may hold several s_endpgm)
            break
        if not inside:
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        f, ln = cur
        key = f + ": (no line: merged by the compiler)" if ln == 0 else f
        for a, b, name in ranges.get(f, []):
            if a <= ln <= b:
                key = f"{f}: {name}"
        a = agg[key]
        a[0] += 1
        a[1 if t.startswith("v_") else 2 if t.startswith("s_") else 3 if t.startswith("ds_") else 4] += 1
        by_line[cur] += 1
        ops[t.split()[0]] += 1
        if args.opcode and t.split()[0] == args.opcode:
            op_lines[cur] += 1
    print(f"# {kern}, schema {args.schema}{', fast walk only' if args.fast_only else ''}"
          f"{', variant ' + args.variant if args.variant else ''}: static instructions per source function")
    print("%-58s %6s %6s %6s %5s %5s" % ("where", "total", "VALU", "SALU", "DS", "VMEM"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print("%-58s %6d %6d %6d %5d %5d" % (k[:58], *v))
    print("%-58s %6d %6d %6d %5d %5d" % ("TOTAL", *[sum(v[i] for v in agg.values()) for i in range(5)]))
    print("opcodes:", ", ".join(f"{o} {n}" for o, n in ops.most_common(24)))
    for k, v in by_line.most_common(args.lines):
        print(f"  {k[0]}:{k[1]}  {v}")
    if args.opcode:
        print(f"{args.opcode} by source line:")
        for k, v in op_lines.most_common(30):
            print(f"  {k[0]}:{k[1]}  {v}")


if __name__ == "__main__":
    main()
