#!/usr/bin/env python3
"""Re-flow a Markdown file to at most WIDTH columns: paragraphs and list items are re-wrapped (continuation lines of a list
item keep its indent), fenced code blocks and headings are left alone, and a table that has a line longer than WIDTH is turned
into a list (one item per row, one sub-item per further column, labelled with the column's heading) -- Markdown cannot wrap a
table cell.  Usage: python tools/wrap_md.py FILE [WIDTH=120]   (rewrites FILE in place)"""
import re
import sys
import textwrap


def cells(line):
    line = line.strip()
    if line.startswith("|"):
        line = line[1:]
    if line.endswith("|"):
        line = line[:-1]
    return [c.strip() for c in re.split(r"(?<!\\)\|", line)]


def wrap_block(text, width, first, rest):
    return textwrap.fill(" ".join(text.split()), width=width, initial_indent=first, subsequent_indent=rest,
                         break_long_words=False, break_on_hyphens=False)


def main(path, width=120):
    src = open(path).read().split("\n")
    out, i, n = [], 0, len(src)
    while i < n:
        line = src[i]
        if line.startswith("```"):                       # fenced code: as is
            out.append(line); i += 1
            while i < n and not src[i].startswith("```"):
                out.append(src[i]); i += 1
            if i < n:
                out.append(src[i]); i += 1
            continue
        if line.startswith("#") or not line.strip():
            out.append(line); i += 1
            continue
        if line.lstrip().startswith("|") and i + 1 < n and re.match(r"^\s*\|?[\s:|-]+\|?\s*$", src[i + 1]) and "-" in src[i + 1]:
            j = i
            while j < n and src[j].lstrip().startswith("|"):
                j += 1
            rows = src[i:j]
            if max(len(r) for r in rows) <= width:
                out += rows
            else:
                head = cells(rows[0])
                for r in rows[2:]:
                    c = cells(r)
                    out.append(wrap_block("* **" + c[0].strip("* ") + "**", width, "", "  "))
                    for h, v in zip(head[1:], c[1:]):
                        if v:
                            out.append(wrap_block(("*" + h + "*: " if h else "") + v, width, "  * ", "    "))
            i = j
            continue
        m = re.match(r"^(\s*)([*+-]|\d+\.)\s+", line)
        if m:                                            # a list item with its continuation lines
            indent = " " * len(m.group(0))
            buf = [line[len(m.group(0)):]]
            i += 1
            while i < n and src[i].strip() and not re.match(r"^\s*([*+-]|\d+\.)\s+", src[i]) and not src[i].startswith("#") \
                    and not src[i].lstrip().startswith("|") and not src[i].startswith("```"):
                buf.append(src[i].strip()); i += 1
            out.append(wrap_block(" ".join(buf), width, m.group(0), indent))
            continue
        buf = [line]                                     # a paragraph
        i += 1
        while i < n and src[i].strip() and not re.match(r"^\s*([*+-]|\d+\.)\s+", src[i]) and not src[i].startswith("#") \
                and not src[i].lstrip().startswith("|") and not src[i].startswith("```"):
            buf.append(src[i]); i += 1
        out.append(wrap_block(" ".join(buf), width, "", ""))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 120)
