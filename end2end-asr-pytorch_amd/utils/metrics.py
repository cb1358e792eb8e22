"""Loss and error-rate metrics with the reference's call signatures (reference: utils/metrics.py).

calculate_loss / calculate_metrics run the label-smoothed cross entropy + argmax + num_correct kernel
(asr_ce_fwd / asr_ce_bwd).  CER / WER use our own Levenshtein distance (asr_hip.text) because python-Levenshtein
(reference: metrics.py:3) is a third-party C extension that is not part of this build.
"""
import torch

from asr_hip import functions as F_
from asr_hip.text import edit_distance
from utils import constant


def calculate_cer(s1, s2):
    """Character edit distance between hypothesis s1 and gold s2 (reference: metrics.py:48-56)."""
    return edit_distance(s1, s2)


def calculate_wer(s1, s2):
    """Word edit distance (reference: metrics.py:58-76; the word->char remapping there only exists to satisfy
    python-Levenshtein's string-only API)."""
    return edit_distance(s1.split(), s2.split())


def _is_chinese(ch):
    return '一' <= ch <= '鿿'


def calculate_cer_en_zh(s1, s2):
    """Mixed English / Chinese CER (reference: metrics.py:9-46 with data/helper.py:43-98): words containing a CJK
    character are scored per language.  Returns (en_dist, zh_dist, len_en_gold, len_zh_gold)."""
    def split(s):
        en, zh = [], []
        for seg in s.split():
            (zh if any(_is_chinese(c) for c in seg) else en).append(seg)
        return " ".join(en), " ".join(zh)
    en1, zh1 = split(s1)
    en2, zh2 = split(s2)
    return calculate_cer(en1, en2), calculate_cer(zh1, zh2), len(en2), len(zh2)


def calculate_loss(pred, gold, input_lengths=None, target_lengths=None, smoothing=0.0, loss_type="ce", global_count=None):
    """pred (B,T,V), gold (B,T) -> scalar loss (reference: metrics.py:102-132).
    ce: mean over non-PAD tokens of the label-smoothed row loss, smoothing mass eps/V on every class (metrics.py:124).
    global_count: optional device scalar replacing the local non-PAD count (exact loss under data parallelism)."""
    if loss_type == "ctc":
        # F.log_softmax + F.ctc_loss(reduction="mean") with blank = PAD (reference: metrics.py:133-154), one fused HIP path
        if input_lengths is None or target_lengths is None:
            raise ValueError("loss_type='ctc' needs input_lengths and target_lengths")
        return F_.CTCFn.apply(pred, gold, input_lengths, target_lengths, constant.PAD_TOKEN)
    if loss_type != "ce":
        raise NotImplementedError("loss_type must be 'ce' or 'ctc'")
    loss, _, _ = F_.CEFn.apply(pred, gold, float(smoothing), constant.PAD_TOKEN, global_count)
    return loss


def calculate_metrics(pred, gold, input_lengths=None, target_lengths=None, smoothing=0.0, loss_type="ce", sync=True,
                      global_count=None, with_argmax=False):
    """-> (loss, num_correct)  (reference: metrics.py:78-100).  num_correct is a Python int as in the reference
    (one device sync); pass sync=False to get the fp32 device tensor [loss_sum, count, num_correct] instead.
    with_argmax (ce, sync=False): -> (loss, sums, argmax (B, T) int64): the row arg-max the loss kernel finds anyway (lowest index on
    ties, like asr_argmax_rows) -- the captured training step takes its hyp_seq from here instead of a second pass over the logits."""
    if loss_type == "ctc":              # the reference returns (loss, None) (metrics.py:96-97)
        return calculate_loss(pred, gold, input_lengths, target_lengths, smoothing, "ctc"), None
    if loss_type != "ce":
        raise NotImplementedError("loss_type must be 'ce' or 'ctc'")
    loss, sums, am = F_.CEFn.apply(pred, gold, float(smoothing), constant.PAD_TOKEN, global_count)
    if sync:
        return loss, int(sums[2].item())
    if with_argmax:
        return loss, sums, am.view(pred.shape[0], pred.shape[1])
    return loss, sums
