"""Re-flow the prose of a markdown file to <= WIDTH columns (tables, headings, code blocks and HTML comments stay as they are;
list items keep a hanging indent).  usage: python tools/wrap_md.py FILE [WIDTH=120]"""
import re
import sys
import textwrap

path, width = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 120
out, block, in_code = [], [], False


def flush():
    if not block:
        return
    if any(l.startswith(("|", "#", "<!--")) for l in block):
        out.extend(block)
    else:
        items, cur = [], []
        for l in block:                                  # split the block into list items / paragraphs
            if re.match(r"^\s*([*-]|\d+\.)\s", l) and cur:
                items.append(cur)
                cur = []
            cur.append(l)
        items.append(cur)
        for it in items:
            m = re.match(r"^(\s*(?:[*-]|\d+\.)\s+)", it[0])
            first = m.group(1) if m else re.match(r"^\s*", it[0]).group(0)
            rest = " " * len(first) if m else first
            text = " ".join(x.strip() for x in it)
            if m:
                text = text[len(m.group(1).strip()) + 1:].strip() if text.startswith(m.group(1).strip()) else text
            out.extend(textwrap.wrap(text, width=width, initial_indent=first, subsequent_indent=rest,
                                     break_long_words=False, break_on_hyphens=False))
    block.clear()


for line in open(path).read().split("\n"):
    if line.startswith("```"):
        flush()
        in_code = not in_code
        out.append(line)
    elif in_code:
        out.append(line)
    elif line.strip() == "":
        flush()
        out.append("")
    else:
        block.append(line)
flush()
open(path, "w").write("\n".join(out))
