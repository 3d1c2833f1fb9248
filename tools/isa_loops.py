"""Lists every innermost loop of a kernel in a gfx950 .s file with its instruction counts (VALU / LDS /
global), e.g. for the reference-stream kernels (smmc_ref_kernels.hip), whose period loops are not at a
fixed nesting depth.

usage: isa_loops.py <file.s | source.hip> <kernel symbol substring>
"""
import os
import re
import sys
from collections import Counter


def kernel_body(lines, sym):
    beg = [i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0] and l.rstrip().endswith(l.split(":")[0].strip()) or
           (l.startswith("_Z") and sym in l and ":" in l)][0]
    fin = [i for i, l in enumerate(lines) if i > beg and "s_endpgm" in l][0]
    return lines[beg:fin]


def loops(body):
    out = []
    for i, l in enumerate(body):
        if "Inner Loop Header" not in l:
            continue
        j = i
        while not re.match(r"^\.LBB\d+_\d+:", body[j]):
            j -= 1
        label = body[j].split(":")[0]
        k = i
        while k < len(body) and not ("s_cbranch" in body[k] and body[k].split()[-1] == label):
            k += 1
        if k == len(body):
            continue
        ins = [x.split()[0] for x in body[j + 1:k + 1] if x.strip() and x.strip()[0] not in ";."]
        out.append((label, Counter(ins)))
    return out


def summary(c):
    v = sum(n for k, n in c.items() if k.startswith("v_"))
    d = sum(n for k, n in c.items() if k.startswith("ds_"))
    g = sum(n for k, n in c.items() if k.startswith("global_") or k.startswith("buffer_"))
    return v, d, g


if __name__ == "__main__":
    path, sym = sys.argv[1], sys.argv[2]
    if not path.endswith(".s"):
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import isa_loop_count as I
        path = I.emit_asm("/tmp/" + os.path.basename(path) + ".s", os.path.basename(path))
    lines = open(path).read().splitlines()
    for label, c in loops(kernel_body(lines, sym)):
        v, d, g = summary(c)
        print(f"{label}: VALU {v}  LDS {d}  global {g}   {dict(c.most_common(8))}")
