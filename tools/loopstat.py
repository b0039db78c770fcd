"""Static loop statistics of a gfx950 .s file: for each kernel named on the command line,
list every backward-branch loop with its instruction count and mnemonic histogram.
Usage: python tools/loopstat.py file.s kernel_substring [...]"""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read()
for kname in sys.argv[2:]:
    m = re.search(r"^(_Z\w*" + kname + r"\w*):[^\n]*\n(.*?)s_endpgm", src, re.S | re.M)
    if not m:
        print("kernel not found:", kname)
        continue
    body = m.group(2).split("\n")
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    tot = sum(1 for l in body if re.match(r"^\s+[a-z]", l))
    print("==", m.group(1), "total instrs", tot)
    for a, b in loops:
        c = Counter()
        for l in body[a:b + 1]:
            mm = re.match(r"^\s+([a-z_0-9]+)", l)
            if mm:
                c[mm.group(1)] += 1
        n = sum(c.values())
        if n < 40:
            continue
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print("  loop lines %d-%d: %d instrs, valu %d | " % (a, b, n, valu)
              + ", ".join("%s:%d" % (k, v) for k, v in c.most_common(16)))
