#!/usr/bin/env python3
"""Wrap a markdown file's prose at N columns (default 120) and give it a table of contents of its `##` headings.
Tables, code fences, headings and lines that are already short are left alone; list items wrap with a hanging indent.
  python3 tools/wrap_md.py FILE [--width 120] [--toc]"""
import re, sys, textwrap


def wrap(lines, width):
    out, fence, toc = [], False, False
    for ln in lines:
        if ln.strip() in ("<!-- toc -->", "<!-- /toc -->"):
            toc = ln.strip() == "<!-- toc -->"; out.append(ln); continue
        if ln.lstrip().startswith("```"):
            fence = not fence; out.append(ln); continue
        if fence or toc or len(ln) <= width or ln.lstrip().startswith(("|", "#")) or not ln.strip():
            out.append(ln); continue
        m = re.match(r"^(\s*(?:[*+-]|\d+\.)\s+)", ln)
        lead = m.group(1) if m else re.match(r"^(\s*)", ln).group(1)
        hang = " " * len(lead)
        body = ln[len(lead):]
        out += textwrap.wrap(body, width=width, initial_indent=lead, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False)
    return out


def slug(h):
    s = re.sub(r"[`*]", "", h.strip().lower())
    s = re.sub(r"[^\w\- ]", "", s, flags=re.UNICODE)
    return s.replace(" ", "-")


def main():
    path = sys.argv[1]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 120
    lines = open(path).read().split("\n")
    if "--toc" in sys.argv:
        # drop an old table of contents (between the markers), then write a new one after the title block
        if "<!-- toc -->" in lines:
            a, b = lines.index("<!-- toc -->"), lines.index("<!-- /toc -->")
            lines = lines[:a] + lines[b + 1:]
        heads = [l[3:] for l in lines if l.startswith("## ")]
        heads = [h for h in heads if h.strip()]
        first = next(i for i, l in enumerate(lines) if l.startswith("## "))
        toc = ["<!-- toc -->", "**Contents**", ""] + ["%d. [%s](#%s)" % (i + 1, h if len(h) <= 110 else h[:107] + "...", slug(h)) for i, h in enumerate(heads)] + ["", "<!-- /toc -->", ""]
        lines = lines[:first] + toc + lines[first:]
    open(path, "w").write("\n".join(wrap(lines, width)))


if __name__ == "__main__":
    main()
