"""Paragraph-aware re-wrap of a markdown file to <= WIDTH columns (DESIGN.md's rule: <= 140). Only paragraphs / bullets holding a line longer
than WIDTH are touched; tables, headings, fenced blocks stay as they are. usage: python tools/reflow_md.py DESIGN.md [width]"""
import sys
import textwrap

path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 140
lines = open(path).read().split("\n")
out, i, fenced = [], 0, False


def starts_block(l):
    return l.startswith(("* ", "- ", "#", "|", "```")) or not l.strip()


while i < len(lines):
    l = lines[i]
    if l.startswith("```"):
        fenced = not fenced
    if fenced or l.startswith(("#", "|", "```")) or not l.strip():
        out.append(l)
        i += 1
        continue
    j = i + 1
    while j < len(lines) and not starts_block(lines[j]):
        j += 1
    para = lines[i:j]
    if max(len(x) for x in para) > W:
        bullet = para[0].startswith(("* ", "- "))
        text = " ".join(x.strip() for x in para)
        para = textwrap.wrap(text, W, subsequent_indent="  " if bullet else "", break_long_words=False, break_on_hyphens=False)
    out.extend(para)
    i = j
open(path, "w").write("\n".join(out))
print(path, len(out), "lines, longest", max(len(x) for x in out))
