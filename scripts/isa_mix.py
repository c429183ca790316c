#!/usr/bin/env python3
"""Instruction mix of the largest loop of one kernel in towr_amd/csrc/kernels.s (`make -C towr_amd/csrc asm`).
usage: isa_mix.py <mangled-name-substring> [top]"""
import collections
import re
import sys

path = "towr_amd/csrc/kernels.s"
lines = open(path).read().split("\n")
key = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
start = [i for i, l in enumerate(lines) if key in l and re.match(r"^_Z\w+:", l)][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end")][0]
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i))
print("function lines", len(body), "loops", [(a, b, b - a) for a, b in loops])
a, b = max(loops, key=lambda p: p[1] - p[0]) if loops and "--whole" not in sys.argv else (0, len(body) - 1)
c = collections.Counter()
for l in body[a:b + 1]:
    l = l.strip()
    if not l or l.startswith((".", ";", "//")) or l.endswith(":"):
        continue
    c[l.split()[0]] += 1
cat = collections.Counter()
for op, n in c.items():
    if op.startswith("v_") and "f64" in op:
        cat["valu f64"] += n
    elif op.startswith("v_"):
        cat["valu other"] += n
    elif op.startswith("s_"):
        cat["salu"] += n
    elif op.startswith("ds_"):
        cat["lds"] += n
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        cat["vmem"] += n
    else:
        cat[op] += n
print("loop instructions", sum(c.values()), dict(cat))
for op, n in c.most_common(top):
    print("%-28s %d" % (op, n))
