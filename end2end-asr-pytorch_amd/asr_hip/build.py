"""Builds libasr_hip.so (gfx950 only) in-tree with hipcc.  No torch involvement: the library is a plain C ABI."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
LIB = os.path.join(HERE, "libasr_hip.so")
OBJ = os.path.join(HERE, "_obj")
# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in architectural VGPRs (gfx950 has a unified file); without it the
# softmax / epilogue code pays a v_accvgpr_read/write pair for every accumulator element it touches.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + os.environ.get("ASR_HIPCC_EXTRA", "").split()    # e.g. -DASR_TUNE_ABLATE (tuning builds)
# Per-file additions.  attention_pp.hip: with NaNs honoured every fmaxf() on an MFMA result is preceded by a canonicalising
# v_max_f32 v, x, x (the row maximum of a score tile cost 55 vector instructions instead of 16 v_max3_f32); the kernel never
# produces or consumes a NaN by construction (masked scores are -inf, the reference starts finite).
PER_FILE_FLAGS = {"attention_pp.hip": ["-fno-honor-nans"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256()
    for root in (CSRC, INCLUDE):
        for f in sorted(os.listdir(root)):
            if f.endswith((".hip", ".h")):
                h.update(f.encode())
                h.update(open(os.path.join(root, f), "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hipcc = _hipcc()

    def one(f):
        o = os.path.join(OBJ, f[:-4] + ".o")
        cmd = [hipcc] + FLAGS + PER_FILE_FLAGS.get(f, []) + ["-c", os.path.join(CSRC, f), "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (f, r.stderr[-4000:]))
        if verbose:
            print("compiled", f)
        return o

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(one, srcs))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    open(stamp, "w").write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
