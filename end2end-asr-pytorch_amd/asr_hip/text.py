"""Host-side text metric used by CER / WER (replaces the python-Levenshtein C extension the reference imports,
reference: utils/metrics.py:3,56,76)."""


def edit_distance(a, b):
    """Levenshtein distance between two sequences (strings or lists), O(len(a) * len(b)) two-row DP."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        ca = a[i - 1]
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cost = prev[j - 1] + (ca != b[j - 1])
            up = prev[j] + 1
            left = cur[j - 1] + 1
            cur[j] = cost if cost < up and cost < left else (up if up < left else left)
        prev = cur
    return prev[-1]
