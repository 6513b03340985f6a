#!/usr/bin/env python3
"""Whitespace-only repair of the reference's 1d-burgers/ide_cont_burgers.py.

TEST INFRASTRUCTURE (build container only: needs /root/reference).

The reference file does not parse: its class body mixes 0-, 2-, 4- and 6-space indentation
(IndentationError at line 31).  Every statement is there, only the leading blanks are wrong.  This
module holds NO reference text -- just the table `INDENT` {line number: leading blanks the line needs
for the evident block structure} -- and `repaired_source()` applies it to the file read from
/root/reference at run time.  `check()` proves the repair is whitespace-only: with all leading
blanks stripped the two files are equal line by line (the same statement `diff -w` makes), and the
result compiles.  make_golden.py runs the repaired module over the test shims to pin SURVEY 8(a)
row 12 (identification: ide_cont_burgers.py:52-118,176-210) to the reference's own code.
"""
import os

REF_FILE = "/root/reference/1d-burgers/ide_cont_burgers.py"


def _span(a, b, n):
    return {i: n for i in range(a, b + 1)}


# 1-based line number -> number of leading blanks.  Lines not listed are kept as they are.
INDENT = {}
INDENT.update(_span(30, 43, 4))      # default hp block: body of `else:` (:29)
INDENT.update(_span(51, 53, 8))      # lambda_1 / lambda_2: body of __init__ (:48)
INDENT.update({55: 4, 56: 4})        # def f_model
INDENT.update(_span(57, 64, 8))
INDENT.update(_span(65, 75, 12))     # body of `with tf.GradientTape(...)` (:64)
INDENT.update(_span(77, 85, 8))
INDENT.update({87: 4, 88: 4})        # def loss
INDENT.update({89: 8, 90: 8, 91: 12})
INDENT.update({93: 4, 94: 8, 95: 8, 96: 8})            # wrap_training_variables
INDENT.update({98: 4, 99: 8, 100: 8, 101: 8, 102: 8})  # get_weights
INDENT.update({104: 4, 105: 8, 106: 8, 107: 8})        # set_weights
INDENT.update({109: 4, 110: 8, 111: 8, 112: 8, 113: 12, 114: 8})   # get_params
INDENT.update({116: 4, 117: 8, 118: 8})                # fit
INDENT.update(_span(120, 167, 4))    # commented-out older fit(): comment lines, any indent parses
INDENT.update({169: 4, 170: 8, 171: 8, 172: 8})        # predict
INDENT.update(_span(188, 192, 4))    # body of error() (:187)


def repaired_source(path=REF_FILE):
    with open(path, encoding="utf-8") as fh:
        lines = fh.read().split("\n")
    out = []
    for no, line in enumerate(lines, 1):
        body = line.lstrip(" \t")
        if no in INDENT and body:
            out.append(" " * INDENT[no] + body)
        else:
            out.append(line)
    return "\n".join(out)


def check(path=REF_FILE):
    """-> repaired source; raises if the repair touches anything but leading blanks or does not compile."""
    with open(path, encoding="utf-8") as fh:
        orig = fh.read()
    fixed = repaired_source(path)
    a = [l.lstrip(" \t") for l in orig.split("\n")]
    b = [l.lstrip(" \t") for l in fixed.split("\n")]
    if a != b:
        raise AssertionError("repair changed more than leading whitespace")
    compile(fixed, path, "exec")
    return fixed


def write(dst):
    src = check()
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w", encoding="utf-8") as fh:
        fh.write(src)
    return dst


if __name__ == "__main__":
    import sys
    s = check()
    print("whitespace-only repair ok: %d lines, %d re-indented" % (s.count("\n") + 1, len(INDENT)))
    if len(sys.argv) > 1:
        print("written to", write(sys.argv[1]))
