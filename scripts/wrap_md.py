#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file at a column (default 120): paragraphs and list items are re-flowed with their
indentation kept; tables, fenced code, headings and lines that end in two spaces are left alone.

    python scripts/wrap_md.py DESIGN.md [width]
"""
import re
import sys
import textwrap


def wrap(text, width=120):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([*\-+] |\d+\. )?", first)
        lead = m.group(1)
        bullet = m.group(2) or ""
        body = " ".join([first[len(lead) + len(bullet):].strip()] + [p.strip() for p in para[1:]])
        out.extend(textwrap.wrap(body, width=width, initial_indent=lead + bullet,
                                 subsequent_indent=lead + " " * len(bullet), break_long_words=False,
                                 break_on_hyphens=False) or [""])
        para.clear()

    for line in text.split("\n"):
        stripped = line.strip()
        if stripped.startswith("```"):
            flush()
            fence = not fence
            out.append(line)
            continue
        if fence or stripped.startswith("|") or stripped.startswith("#") or stripped == "" or line.endswith("  "):
            flush()
            out.append(line)
            continue
        is_item = re.match(r"^\s*([*\-+] |\d+\. )", line) is not None
        if is_item:
            flush()
            para.append(line)
        elif para:
            # a continuation line: same paragraph unless its indentation says it is a new block
            para.append(line)
        else:
            para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    with open(path) as fh:
        src = fh.read()
    with open(path, "w") as fh:
        fh.write(wrap(src, width))
