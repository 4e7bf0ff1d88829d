"""Re-flow over-long COMMENT-ONLY lines (housekeeping): a line that holds nothing but a `#` (Python) or `//` (HIP / C++)
comment and is wider than WIDTH is wrapped at word boundaries into lines of the same indentation and comment marker.  Code lines, trailing
comments, docstrings and block comments are left alone, so the program text is unchanged (Python files are re-parsed and compared by AST;
for HIP files the token stream outside comments is compared).

    python tools/wrap_comments.py [--width 150] file ...
"""
import ast
import io
import re
import sys
import textwrap
import tokenize

WIDTH = 150


def wrap_line(indent, marker, text, width):
    body = textwrap.wrap(text, width=width - len(indent) - len(marker) - 1, break_long_words=False, break_on_hyphens=False)
    return [f"{indent}{marker} {b}" for b in body] or [f"{indent}{marker}"]


def py_comment_only_lines(src):
    rows = set()
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type == tokenize.COMMENT and tok.line[:tok.start[1]].strip() == "":
            rows.add(tok.start[0])
    return rows


def process(path, width):
    src = open(path).read()
    lines = src.split("\n")
    is_py = path.endswith(".py")
    rows = py_comment_only_lines(src) if is_py else None
    out, changed, in_block = [], 0, False
    for i, ln in enumerate(lines, 1):
        if not is_py:
            if in_block:
                in_block = "*/" not in ln
                out.append(ln)
                continue
            if "/*" in ln and "*/" not in ln.split("/*", 1)[1]:
                in_block = True
                out.append(ln)
                continue
        m = re.match(r"^(\s*)(#|//)( ?)(.*)$", ln)
        ok = m and len(ln) > width and (i in rows if is_py else True) and not ln.rstrip().endswith("\\")
        # keep hand-aligned tables / continuation-indented comment text as they are: only plain prose lines are re-flowed
        if ok and not m.group(4).startswith(("  ", "\t", "|", "!")) and "   " not in m.group(4).strip():
            first = wrap_line(m.group(1), m.group(2), m.group(4).strip(), width)
            # continuation lines of a "(...)" or "  *" style comment keep one extra space when the original text began with one
            out.extend(first)
            changed += 1
        else:
            out.append(ln)
    new = "\n".join(out)
    if is_py:
        assert ast.dump(ast.parse(src)) == ast.dump(ast.parse(new)), path
    else:
        strip = lambda s: re.sub(r"//[^\n]*", "", s)                      # noqa: E731
        assert re.sub(r"\s+", " ", strip(src)) == re.sub(r"\s+", " ", strip(new)), path
    if changed:
        open(path, "w").write(new)
    return changed


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--width":
        WIDTH = int(args[1])
        args = args[2:]
    for f in args:
        print(f, process(f, WIDTH), "comment lines re-flowed")
