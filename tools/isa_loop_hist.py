#!/usr/bin/env python3
"""Instruction histogram of a kernel's barrier-to-barrier intervals, from the compiler's assembly -- no GPU needed.

    hipcc -O3 --offload-arch=gfx950 --cuda-device-only -mllvm -amdgpu-mfma-vgpr-form=1 [per-file flags of asr_hip/build.py] \\
          -S -o /tmp/k.s end2end-asr-pytorch_amd/csrc/attention_pp.hip -Iend2end-asr-pytorch_amd/csrc -Iinclude
    python tools/isa_loop_hist.py /tmp/k.s 'attn_fwd_pp.*ILb1ELi4' [--blocks]

Prints, for every interval between two s_barrier of the first kernel whose mangled name matches the pattern, the number of MFMA,
vector, LDS and scalar instructions and the most frequent opcodes; --blocks splits an interval at its labels / branches (a block behind a
wave-uniform branch -- mask tile, moved running maximum -- is in the listing but not in the steady state).  This is where DESIGN.md's
"vector instructions per tile" figures come from; they agree with SQ_INSTS_VALU / SQ_INSTS_MFMA of the PMC passes."""
import collections
import re
import sys


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    blocks = "--blocks" in sys.argv[3:]
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    mine = [i for i in starts if pat.search(lines[i])]
    if not mine:
        raise SystemExit("no kernel matches %r" % sys.argv[2])
    a = mine[0]
    b = min([i for i in starts if i > a] + [len(lines)])
    seg = lines[a:b]
    print(lines[a].split(":")[0])

    def hist(x, y):
        c = collections.Counter()
        for l in seg[x:y]:
            t = l.strip().split()
            if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
                continue
            c[t[0]] += 1
        return c

    def line(x, y, tag):
        h = hist(x, y)
        mfma = sum(v for k, v in h.items() if k.startswith("v_mfma"))
        valu = sum(v for k, v in h.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        lds = sum(v for k, v in h.items() if k.startswith("ds_"))
        vmem = sum(v for k, v in h.items() if k.startswith(("global_", "buffer_", "flat_")))
        salu = sum(v for k, v in h.items() if k.startswith("s_"))
        top = ", ".join("%s %d" % kv for kv in h.most_common(12))
        print("%s lines %5d-%5d  mfma %3d  valu %4d  lds %3d  vmem %3d  salu %4d | %s" % (tag, x, y, mfma, valu, lds, vmem, salu, top))

    bars = [i for i, l in enumerate(seg) if l.strip().startswith("s_barrier")]
    pts = [0] + bars + [len(seg)]
    for x, y in zip(pts, pts[1:]):
        line(x, y, "interval")
        if blocks:
            cut = [x] + [i for i in range(x + 1, y) if seg[i].strip().startswith((".LBB", "s_cbranch", "s_branch"))] + [y]
            for u, v in zip(cut, cut[1:]):
                if v - u > 3:
                    line(u, v, "   block  ")


if __name__ == "__main__":
    main()
