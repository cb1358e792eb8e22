#!/usr/bin/env python3
"""Instruction mix of a kernel's basic blocks from hipcc -S output (tuning aid).
usage: tools/isa_mix.py file.s <substring of the mangled kernel name> [--dump BLOCK]
Classes: mfma, valu (v_* other than mfma), salu, lds (ds_*), vmem (global_/buffer_/scratch_), wait (s_waitcnt/s_nop/s_barrier), other."""
import re
import sys
from collections import Counter, OrderedDict


def cls(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        blocks[cur].append(t.split(";")[0].strip())
    tot = Counter()
    for b, ins in blocks.items():
        c = Counter(cls(i.split()[0]) for i in ins)
        tot.update(c)
        br = [i for i in ins if i.startswith(("s_cbranch", "s_branch"))]
        print("%-14s n=%4d  mfma %3d valu %4d lds %3d vmem %3d salu %3d wait %3d  %s" % (
            b, len(ins), c["mfma"], c["valu"], c["lds"], c["vmem"], c["salu"], c["wait"], " ".join(x.split()[-1] for x in br)))
        if dump == b:
            ops = Counter(i.split()[0] for i in ins)
            for k, v in ops.most_common():
                print("      %4d %s" % (v, k))
    print("total", dict(tot))


main()
