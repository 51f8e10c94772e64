#!/usr/bin/env python3
"""Re-wraps the long lines of a markdown file at WIDTH columns (tables, code fences and headings stay as they are; list items keep their hanging indent)."""
import re
import sys
import textwrap

WIDTH = 150


def wrap(path):
    out, fence = [], False
    for line in open(path).read().split("\n"):
        if line.lstrip().startswith("```"):
            fence = not fence
        if fence or len(line) <= WIDTH or line.lstrip().startswith("|") or line.startswith("#"):
            out.append(line)
            continue
        m = re.match(r"^(\s*)((?:[*+-]|\d+\.)\s+)?", line)
        lead, bullet = m.group(1), m.group(2) or ""
        body = line[len(lead) + len(bullet):]
        out.extend(textwrap.wrap(body, WIDTH, initial_indent=lead + bullet, subsequent_indent=lead + " " * len(bullet), break_long_words=False, break_on_hyphens=False))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        wrap(p)
