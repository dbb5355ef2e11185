"""Wrap the paragraphs of a markdown file at a readable width (tables, code blocks, headings and list markers are kept):
python tools/wrap_md.py FILE [width]"""
import sys
import textwrap


def wrap(text, width=132):
    out, in_code = [], False
    for line in text.split("\n"):
        s = line.lstrip()
        if s.startswith("```"):
            in_code = not in_code
            out.append(line)
            continue
        if in_code or len(line) <= width or s.startswith("|") or s.startswith("#"):
            out.append(line)
            continue
        indent = line[:len(line) - len(s)]
        sub = indent
        for m in ("* ", "- ", "+ "):
            if s.startswith(m):
                sub = indent + "  "
                break
        else:
            head = s.split(" ", 1)[0]
            if head.rstrip(".)").isdigit() and head[-1:] in ".)":
                sub = indent + " " * (len(head) + 1)
        out.extend(textwrap.wrap(s, width=width, initial_indent=indent, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False))
    return "\n".join(out)


if __name__ == "__main__":
    p = sys.argv[1]
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 132
    t = open(p).read()
    open(p, "w").write(wrap(t, w))
