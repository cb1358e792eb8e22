"""Static checks on the gfx950 code the compiler emits for three software-pipelined kernels (no GPU: hipcc cross-compiles to assembly
with the flags of asr_hip/build.py).  They guard properties the measurements of DESIGN.md section 4 depend on and that a compiler
upgrade or an innocent edit can silently undo:
  * the private three-stage rings of the four-wave GEMM kernels keep their COUNTED waits -- a compiler that sees the LDS-DMA in flight
    drains it (s_waitcnt vmcnt(0)) before every LDS read and the ring degenerates to "load, wait, compute";
  * the long-sequence attention forward applies its dropout mask with three packed 16-bit instructions per key pair and one 32-bit
    multiply (the builtin form was compiled to two compares, two selects and a byte permute: 13 instead of 9 instructions per pair)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end2end-asr-pytorch_amd"))


def _asm(tmp, name):
    from asr_hip import build
    hipcc = build._hipcc()
    if os.path.isabs(hipcc) and not os.path.exists(hipcc) or not os.path.isabs(hipcc) and shutil.which(hipcc) is None:
        pytest.skip("hipcc not available")
    out = os.path.join(tmp, name + ".s")
    flags = [f for f in build.FLAGS if f != "-fPIC"] + build.PER_FILE_FLAGS.get(name, [])
    r = subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", os.path.join(build.CSRC, name), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().split("\n")


def _kernel(lines, pattern):
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    mine = [i for i in starts if re.search(pattern, lines[i])]
    assert mine, pattern
    a = mine[0]
    return lines[a:min([i for i in starts if i > a] + [len(lines)])]


def _ops(seg):
    out = []
    for l in seg:
        t = l.strip().split()
        if t and not t[0].startswith((".", ";")) and not t[0].endswith(":"):
            out.append((t[0], l.strip()))
    return out


@pytest.fixture(scope="module")
def gemm_asm(tmp_path_factory):
    return _asm(str(tmp_path_factory.mktemp("isa")), "gemm.hip")


@pytest.mark.parametrize("kernel,pieces", [(r"gemm_nn_kernelIttLi64ELi3E", 4), (r"gemm_glds_kernelIttLi64ELi64ELi3E", 4)])
def test_ring_kernels_keep_their_counted_waits(gemm_asm, kernel, pieces):
    ops = _ops(_kernel(gemm_asm, kernel))
    mf = [i for i, (o, _) in enumerate(ops) if o.startswith("v_mfma")]
    assert len(mf) == 8                                         # one 64-deep K step of a 32 x 32 quadrant: the loop is not unrolled
    # the main loop = everything up to the barrier that follows the last MFMA (the epilogue starts there)
    end = next(i for i in range(mf[-1], len(ops)) if ops[i][0] == "s_barrier")
    waits = [t for o, t in ops[:end] if o == "s_waitcnt" and "vmcnt" in t]
    counted = [t for t in waits if "vmcnt(%d)" % pieces in t]
    drained = [t for t in waits if "vmcnt(0)" in t]
    # exactly the kernel's own two waits: `pieces` DMA instructions of the next step may stay in flight; the last step drains
    assert len(counted) == 1 and len(drained) == 1 and len(waits) == 2, waits
    assert sum(1 for o, _ in ops[:end] if o == "global_load_lds_dwordx4") >= 2 * pieces        # prologue stage(s) + the refill inside the loop


@pytest.mark.parametrize("kernel", [r"gemm_tn256g_kernelILi3ELb1ELb1EE", r"gemm_tn256g_kernelILi3ELb1ELb0EE", r"gemm_tn256s_kernelILi3EE"])
def test_grouped_weight_gradient_loop_reads_under_its_mfmas(gemm_asm, kernel):
    """csrc/gemm.hip tn256r_body (round 5): the operand reads of stage s + 1 are hand-issued under the MFMAs of stage s with counted
    s_waitcnt lgkmcnt.  That only holds while (a) nothing else uses lgkmcnt in the loop -- round 3's loop reloaded lda / ldb from the
    kernel arguments behind every barrier (s_load + s_waitcnt lgkmcnt(0)); (b) a fragment's registers are written by the reads and
    consumed by the MFMAs directly -- a compiler-inserted copy of a register whose read is still in flight would copy stale bits;
    (c) the ring position costs one add per fragment, not the three instructions per read of the old loop."""
    ops = _ops(_kernel(gemm_asm, kernel))
    mf = [i for i, (o, _) in enumerate(ops) if o.startswith("v_mfma")]
    assert len(mf) == 32                                            # one 32-row stage of a 64 x 128 quadrant, not unrolled
    # the stage loop (its basic blocks are not in source order: row 3 precedes the header): everything between the first and the last of
    # its 32 MFMAs and 24 operand reads -- the reads are the LAST 24 of the kernel, the prologue's 24 come before the loop
    tr = [i for i, (o, _) in enumerate(ops) if o == "ds_read_b64_tr_b16"]
    assert len(tr) == 48
    body = mf + tr[24:]
    loop = ops[min(body):max(body) + 1]
    names = [o for o, _ in loop]
    assert names.count("ds_read_b64_tr_b16") == 24 and names.count("s_barrier") == 1
    assert not [t for o, t in loop if o.startswith("s_load")], "a kernel argument is read inside the stage loop"
    lg = [t for o, t in loop if o == "s_waitcnt" and "lgkmcnt" in t]
    assert [int(re.search(r"lgkmcnt\((\d+)\)", t).group(1)) for t in lg] == [15, 14, 12, 10, 8, 6, 4, 2, 0], lg
    frag_regs = set()
    for o, t in loop:
        if o == "ds_read_b64_tr_b16":
            lo, hi = map(int, re.search(r"v\[(\d+):(\d+)\]", t).groups())
            frag_regs.update(range(lo, hi + 1))
    assert len(frag_regs) == 48                                     # 12 fragments x 4 registers, each read in place
    copies = [t for o, t in loop if o.startswith("v_mov") and int(re.search(r"v(\d+)$", t.split(",")[-1].strip()).group(1) if re.search(r"v(\d+)$", t.split(",")[-1].strip()) else -1) in frag_regs]
    assert not copies, copies
    valu = sum(1 for o in names if o.startswith("v_") and not o.startswith("v_mfma"))
    assert valu <= 80 + (64 if "s_kernel" in kernel or "ELb0EE" in kernel else 64), valu      # 12 adds + DMA addresses (+ the bias column sums of the waves that own them)
    seg_txt = "\n".join(t for _, t in ops)
    assert "scratch_" not in seg_txt and "v_accvgpr" not in seg_txt


def test_attention_dropout_mask_is_packed_sixteen_bit_arithmetic(tmp_path):
    lines = _asm(str(tmp_path), "attention_pp.hip")
    seg = _kernel(lines, r"attn_fwd_pp_bf16_d64_kernelILb1ELi4E")
    ops = _ops(seg)
    bars = [i for i, (o, _) in enumerate(ops) if o == "s_barrier"]
    tiles = []
    for a, b in zip(bars, bars[1:]):
        names = [o for o, _ in ops[a:b]]
        if sum(n.startswith("v_mfma") for n in names) == 16:      # one 64-key tile: 8 MFMAs of P V + 8 of the next K Q^T
            tiles.append(names)
    assert len(tiles) >= 2
    for names in tiles[-2:]:                                       # the steady-state iterations of the three-stage loop
        for op in ("v_pk_sub_u16", "v_pk_min_u16", "v_pk_mul_lo_u16"):
            assert names.count(op) == 16, (op, names.count(op))    # 16 key pairs per lane and tile
        assert names.count("v_mul_lo_u32") <= 17                   # one multiply per pair (+ the tile's base)
        assert names.count("v_perm_b32") == 0
        valu = sum(n.startswith("v_") and not n.startswith("v_mfma") for n in names)
        assert valu <= 430, valu                                   # 415 measured (482 before); includes the wave-uniform rare branches
    # and without dropout the tile stays where profiles/r03_attention_d64_pmc.txt measured it
    ops0 = _ops(_kernel(lines, r"attn_fwd_pp_bf16_d64_kernelILb0ELi4E"))
    bars0 = [i for i, (o, _) in enumerate(ops0) if o == "s_barrier"]
    steady = [[o for o, _ in ops0[a:b]] for a, b in zip(bars0, bars0[1:])]
    steady = [n for n in steady if sum(x.startswith("v_mfma") for x in n) == 16][-2:]
    for names in steady:
        assert sum(n.startswith("v_") and not n.startswith("v_mfma") for n in names) <= 285
        assert not any(n.startswith("v_pk_sub_u16") for n in names)


@pytest.fixture(scope="module")
def conv_ws_asm(tmp_path_factory):
    return _asm(str(tmp_path_factory.mktemp("isa_ws")), "conv_ws.hip")


@pytest.mark.parametrize("kernel,mfmas,max_regs", [
    (r"conv3x3_ws128_kernelILi128ELi8ELi128ELi0ELi2ELb0E", 576, 512),      # conv.7 forward, pooled epilogue on tile pairs
    (r"conv3x3_ws128_kernelILi128ELi8ELi128ELi1ELi0ELb0E", 576, 512),      # conv.7 data gradient + mask
    (r"conv3x3_ws128_kernelILi128ELi8ELi128ELi2ELi0ELb0E", 576, 512),      # ... with the mask as one bit per element
    (r"conv3x3_ws128_kernelILi128ELi8ELi64ELi0ELi0ELb0E", 288, 512),       # conv.5 data gradient
    (r"conv3x3_ws128_kernelILi64ELi4ELi128ELi0ELi0ELb0E", 144, 256),       # conv.5 forward in one pass: TWO workgroups per CU
    (r"conv3x3_ws128_kernelILi64ELi4ELi128ELi3ELi0ELb0E", 144, 256),       # ... writing its ReLU mask as bits
])
def test_weight_stationary_conv_keeps_its_weights_where_the_mfma_reads_them(conv_ws_asm, kernel, mfmas, max_regs):
    """csrc/conv_ws.hip (round 5) rests on three compiler-dependent facts: the weights loaded with an "=a" constraint STAY in the
    accumulation half of the register file and the MFMAs take them from there (no v_accvgpr_read: left to itself the allocator treats that
    half as spill space, 253 reads per tile in the first build), nothing spills to scratch, and the 64-input-channel form fits 256
    registers (two workgroups per CU).  One tile's contraction is fully unrolled: 9 taps x k steps x fragments MFMAs."""
    seg = _kernel(conv_ws_asm, kernel)
    ops = _ops(seg)
    assert sum(1 for o, _ in ops if o.startswith("v_mfma")) == mfmas
    assert not [l for o, l in ops if o.startswith("v_accvgpr")], "accumulation-half registers are being copied instead of read by the MFMA"
    assert not [l for o, l in ops if o.startswith("scratch_") or o.startswith("buffer_store") or o.startswith("buffer_load_dword ")]
    assert sum(1 for o, l in ops if o.startswith("v_mfma") and re.search(r"\ba\[\d+:\d+\]", l)) >= mfmas * 3 // 4     # A operands in a[..]
    meta = "\n".join(seg)
    nv = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta)
    sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta)
    assert nv and int(nv.group(1)) <= max_regs and sc and int(sc.group(1)) == 0, (nv and nv.group(1), sc and sc.group(1))
