"""ctypes binding of libasr_hip.so (include/asr_hip.h).  Plumbing only: tensors in, error codes checked.

The library is REQUIRED: there is no CPU or eager-PyTorch fallback for any op on the hot path.  `load()` raises if
the shared object is missing, and every wrapper raises if a tensor is not on a HIP device.
"""
import ctypes
import os
import re

# Kernel arguments of every launch (and of every node of a replayed hipGraph) in DEVICE memory: the runtime's default on this
# stack, pinned here because the step is ~216 dependent launches -- with host-resident arguments every launch starts with a
# PCIe read (measured with HIP_FORCE_DEV_KERNARG=0: 7.17 instead of 6.83 ms/step, profiles/r02_ab_kernarg.txt).  Read by the
# HIP runtime when it initialises, i.e. at the first device call, which comes after this import.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libasr_hip.so")
HEADER = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "asr_hip.h")

F32, BF16 = 0, 1
GEMM_RELU, GEMM_ACCUMULATE = 1, 2
EUNSUPPORTED = -3
ATTN_DELTA, ATTN_DQ, ATTN_DKV, ATTN_ALL = 1, 2, 4, 7
OP_GEMM, OP_CONV_IGEMM, OP_CONV_WGRAD, OP_ATTN_FWD, OP_ATTN_BWD, OP_ADD_LN, OP_CE, OP_ADAM, OP_CONV1, OP_POOL, \
    OP_LAYOUT = range(11)

_lib = None

_P, _I, _L, _F, _U64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64

_SIGS = {
    "asr_strerror": (ctypes.c_char_p, [_I]),
    "asr_abi_version": (_I, []),
    "asr_set_tuning": (_I, [ctypes.c_char_p, _L]),
    "asr_clear_tuning": (_I, [ctypes.c_char_p]),
    "asr_prof_enable": (_I, [_I, _I]),
    "asr_prof_collect": (_I, [_I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "asr_gemm_nt": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _I, _I, _I, _F, _I, _I, _I, _I, _P]),
    "asr_gemm_tn_workspace": (_L, [_I, _I, _I, _I, _I]),
    "asr_gemm_tn": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "asr_gemm_tn_grouped": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "asr_gemm_nn": (_I, [_P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _F, _I, _I, _I, _P]),
    "asr_gemm_nn_rowdot": (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "asr_gemm_tn_grouped_plan": (_I, [_I, _P, _P, _P, _P]),
    "asr_gemm_nn_tn_splits": (_I, [_I, _I]),
    "asr_gemm_nn_tn_workspace": (_L, [_I, _I, _I, _I]),
    "asr_gemm_nn_tn": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _L, _I, _I, _I, _P]),
    "asr_tn_reduce_multi": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "asr_cast_flat": (_I, [_P, _P, _L, _I, _P]),
    "asr_widen_flat": (_I, [_P, _P, _L, _P]),
    "asr_transpose": (_I, [_P, _L, _P, _L, _I, _I, _P, _I, _P]),
    "asr_cast_weight": (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _P]),
    "asr_colsum_acc": (_I, [_P, _L, _I, _I, _P, _I, _P]),
    "asr_add_ln_fwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _F, _F, _U64, _P, _I, _P]),
    "asr_add_ln_bwd_workspace": (_L, [_I, _I]),
    "asr_add_ln_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _U64, _P, _I, _P]),
    "asr_add_ln_bwd_partials": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _U64, _P, _I, _P]),
    "asr_ln_reduce_multi": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "asr_attn_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _L, _P, _P, _L, _L,
                          _I, _F, _F, _U64, _P, _I, _P]),
    "asr_attn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _L,
                          _P, _P, _L, _L, _I, _F, _F, _U64, _P, _I, _I, _P]),
    "asr_decoder_preprocess": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "asr_embed_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _U64, _P, _I, _P]),
    "asr_embed_bwd": (_I, [_P, _P, _P, _I, _I, _I, _F, _F, _U64, _P, _I, _I, _P]),
    "asr_quant_fp8": (_I, [_P, _L, _I, _I, _I, _P, _L, _P, _P]),
    "asr_gemm_nt_fp8": (_I, [_P, _L, _P, _P, _L, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P]),
    "asr_ctc_workspace": (_L, [_I, _I, _I]),
    "asr_ctc_fwd": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P, _P]),
    "asr_ctc_bwd": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _L, _P]),
    "asr_decode_prepare": (_I, [_P, _I, _P, _P, _I, _P, _I, _P]),
    "asr_kv_append": (_I, [_P, _P, _L, _P, _P, _I, _I, _I, _P, _I, _P]),
    "asr_ce_fwd": (_I, [_P, _L, _P, _I, _I, _F, _I, _P, _P, _P, _P]),
    "asr_ce_partial_blocks": (_I, [_I]),
    "asr_ce_fwd_partials": (_I, [_P, _L, _P, _I, _I, _F, _I, _P, _P, _P, _P]),
    "asr_ce_finish": (_I, [_P, _I, _P, _P, _P, _P]),
    "asr_argmax_rows": (_I, [_P, _L, _I, _I, _P, _P]),
    "asr_edit_distance_batch": (_I, [_P, _P, _P, _P, _I, _P]),
    "asr_logsoftmax_topk": (_I, [_P, _L, _I, _I, _I, _P, _P, _P]),
    "asr_dec_gemm": (_I, [_P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P, _P, _P, _P, _F, _P, _P, _P, _P, _F, _P, _P]),
    "asr_dec_attn": (_I, [_P, _L, _P, _P, _L, _P, _P, _L, _L, _I, _P, _L, _I, _I, _I, _F, _I, _P, _P]),
    "asr_dec_attn_fused": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _F, _P, _P, _P, _L, _L, _I, _P, _L, _I, _I, _I, _F, _I,
                                _P, _P]),
    "asr_dec_finish": (_I, [_P, _L, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "asr_ce_bwd": (_I, [_P, _L, _P, _P, _I, _I, _F, _I, _P, _P, _P, _L, _I, _P]),
    "asr_adam_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P, _P]),
    "asr_step_advance": (_I, [_P, _P]),
    "asr_adam_noam_step": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P]),
    "asr_sumsq_acc": (_I, [_P, _L, _P, _P]),
    "asr_clip_coef": (_I, [_P, _F, _P, _P]),
    "asr_grad_coef": (_I, [_P, _F, _P, _P, _P]),
    "asr_length_mask": (_I, [_P, _I, _I, _P, _P]),
    "asr_ratio": (_I, [_P, _P, _P, _P]),
    "asr_conv1_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "asr_conv1_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "asr_conv_pack_weight": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "asr_conv_pack_weight_multi": (_I, [_I, _P, _P, _P, _P, _P, _I, _P]),
    "asr_conv3x3_igemm": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "asr_relu_bits_bytes": (_L, [_I, _I, _I, _I]),
    "asr_conv3x3_igemm_bits": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "asr_conv3x3_relu_pool": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "asr_maxpool_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "asr_maxpool_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "asr_maxpool_fwd_code": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "asr_maxpool_bwd_code": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "asr_conv3x3_relu_pool_code": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "asr_conv3x3_relu_pool_tcf_code": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "asr_conv3x3_relu_pool_tcf_codecl": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "asr_gemm_nn_poolbwd": (_I, [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "asr_permute_cols_tcf": (_I, [_P, _L, _P, _L, _I, _I, _I, _I, _P]),
    "asr_conv3x3_wgrad_workspace": (_L, [_I, _I, _I, _I, _I]),
    "asr_conv3x3_wgrad_nhwc": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "asr_conv3x3_wgrad_partials": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "asr_conv3x3_wgrad_reduce": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "asr_vgg_level0_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "asr_vgg_level0_bwd_workspace": (_L, [_I, _I, _I]),
    "asr_vgg_level0_dgrad": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "asr_vgg_level0_wgrad": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "asr_stft_frames": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _P]),
    "asr_spect_finish": (_I, [_P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "asr_im2col": (_I, [_P, _P] + [_I] * 12 + [_L, _L, _I, _I, _P]),
    "asr_col2im": (_I, [_P, _P] + [_I] * 12 + [_L, _I, _P]),
    "asr_window_sum": (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _P]),
    "asr_bn_stats": (_I, [_P, _L, _L, _I, _P, _P, _I, _I, _P]),
    "asr_bn_stats_blocks": (_L, [_L]),
    "asr_bn_stats_partial": (_I, [_P, _L, _L, _I, _P, _P, _P, _I, _I, _P]),
    "asr_bn_batch_stats": (_I, [_P, _L, _L, _I, _P, _P, _P, _F, _F, _P, _P, _P, _I, _I, _P]),
    "asr_bn_batch_stats_v": (_I, [_P, _L, _L, _I, _P, _P, _P, _F, _F, _P, _P, _P, _I, _I, _P, _I, _P]),
    "asr_bn_act_bwd_reduce_v": (_I, [_P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _P, _P, _I, _I, _P]),
    "asr_bn_act_bwd_v": (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P]),
    "asr_bn_act_fwd": (_I, [_P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _I, _P]),
    "asr_bn_act_bwd_reduce": (_I, [_P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _P, _I, _P]),
    "asr_bn_act_bwd": (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
}


def header_symbols():
    """Every function name declared in include/asr_hip.h."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asr_[a-z0-9_]+)\s*\(", src)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libasr_hip.so is missing (%s): run `python __graft_entry__.py build` -- there is no "
                           "fallback path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    _forward_env_tuning(lib)
    return lib


# A/B switches of the library.  The library itself reads no environment variables: the ASR_<NAME>
# variables present when the library is loaded are forwarded ONCE through asr_set_tuning(); set_tuning() changes them later.
TUNING_NAMES = ("ATTN_GENERIC", "IGEMM_TH", "IGEMM_TPS", "IGEMM_WBUF", "CONV1_WGRAD_MFMA", "IGEMM_ABLATE", "C64", "CONV_POOL",
                "WGRAD_ABLATE", "WGRAD_DMA", "CONV1_WGRAD_WGS", "C64_PER_CU", "C64_ABLATE", "C64_SHAPE", "GEMM_NS", "GEMM_TILE",
                "GEMM_GENERIC", "TN_WGS", "TN_128", "TN_128_MIN", "TN_128_RM", "TN_NBUF", "NN_BIG", "NN_RING", "TN_GROUP_SLICE_MIN", "GEMM_BIG_MIN", "NT_RING", "TN_PIPE", "TN_PIPE_MIN", "GEMM_ABLATE", "ATTN_SHORT", "ATTN_SHORT_BWD", "ATTN_BOTH", "NNTN_STAGES",
                "ATTN_PP", "ATTN_PP_MIN", "ATTN_PP_TAIL", "ATTN_PP_PRIO", "ATTN_PP_STAGGER", "ATTN_PP_STAGGER_SEL", "TN_GROUP_TILE", "TN_GROUP_MROWS",
                "TN_GROUP_WGS", "WGRAD_XCD", "IGEMM_XCD", "C64_SPLIT", "GEMM_BIG", "GEMM_BIG_NS", "GEMM_BIG_NN", "L0_WSPLIT", "WS128", "WS64", "WS64_PER_CU", "WS_PAIR", "WS_BITS", "NN_ROWDOT", "ATTN_BWD_FUSED")


def _forward_env_tuning(lib):
    for name in TUNING_NAMES:
        v = os.environ.get("ASR_" + name)
        if v is None:
            continue
        try:
            iv = int(v)
        except ValueError:
            iv = 1                      # presence switches (ASR_ATTN_GENERIC=yes)
        rc = lib.asr_set_tuning(name.encode(), iv)
        if rc != 0:
            raise RuntimeError("asr_set_tuning(%s) refused" % name)


def set_tuning(name, value):
    """Set (value is not None) or clear a tuning switch of the library, e.g. set_tuning("GEMM_TILE", 2)."""
    lib = load()
    rc = lib.asr_clear_tuning(name.encode()) if value is None else lib.asr_set_tuning(name.encode(), int(value))
    check(rc, "asr_set_tuning(%s)" % name)


class AsrHipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise AsrHipError("%s failed: %s (%d)" % (what, load().asr_strerror(rc).decode(), rc))


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise AsrHipError("unsupported dtype %s" % t.dtype)


def dt_of(dtype):
    return {torch.float32: F32, torch.bfloat16: BF16}[dtype]


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: the HIP path is the only path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise AsrHipError("asr_hip ops need tensors on a HIP device (got %s); there is no CPU path" % t.device)
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def stream():
    """The current HIP stream's handle.  torch's raw accessor when it exists: torch.cuda.current_stream() builds a Stream object and
    resolves the device on every call (~3 us, once per launch on the eager path)."""
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


_fn_cache = {}


def call(name, *args):
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        check(rc, name)
