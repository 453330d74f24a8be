"""ROUGE-L as the reference's evaluation scripts compute it (host side, like the reference's: evaluation/Rouge.py:65-108 LCS,
:186-206 F-measure, :209-245 sentence-level ROUGE-L, evaluation/Eval_Rouge.py:13-68 max over ground truths, mean, x100, 2 decimals).

Used by the acceptance harness (north star: ROUGE-L within 0.2 of the reference on a dev set): tests/test_parity_gpu.py decodes a
synthetic dev set with the HIP path and with the CPU oracle and compares the two scores.  The LCS table is two rolling numpy
rows instead of the reference's dict of (i, j) cells."""
import numpy as np


def lcs_length(x, y):
    """Length of the longest common subsequence of two token sequences."""
    if len(x) == 0 or len(y) == 0:
        return 0
    if len(y) > len(x):
        x, y = y, x
    ids = {}
    a = np.fromiter((ids.setdefault(t, len(ids)) for t in x), dtype=np.int64, count=len(x))
    b = np.fromiter((ids.setdefault(t, len(ids)) for t in y), dtype=np.int64, count=len(y))
    prev = np.zeros(len(b) + 1, dtype=np.int64)
    for tok in a:
        match = prev[:-1] + (b == tok)          # diagonal + 1 where the tokens agree
        cur = np.maximum(prev[1:], match)       # vs the cell above
        cur = np.maximum.accumulate(cur)        # vs the cell to the left (the recurrence is monotone along the row)
        prev = np.concatenate(([0], cur))
    return int(prev[-1])


def rouge_l(hypothesis, reference):
    """(F, P, R) of ROUGE-L between two whitespace-tokenised strings (or token lists), with the reference's F-measure:
    beta = P / (R + 1e-12), F = (1 + beta^2) R P / (R + beta^2 P + 1e-12)."""
    hyp = hypothesis.split(" ") if isinstance(hypothesis, str) else list(hypothesis)
    ref = reference.split(" ") if isinstance(reference, str) else list(reference)
    llcs = lcs_length(hyp, ref)
    r, p = llcs / len(ref), llcs / len(hyp)
    beta = p / (r + 1e-12)
    f = (1 + beta ** 2) * r * p / (r + beta ** 2 * p + 1e-12)
    return f, p, r


def eval_rouge_l(run, ref):
    """``run``: one hypothesis string per item; ``ref``: a list of ground-truth strings per item.  Per item the best F over its
    ground truths, x100; the mean over items rounded to 2 decimals (what Run_Evaluation.py prints as ROUGE_L_F1)."""
    assert len(run) == len(ref), "the length of predicted span and ground_truths span should be same"
    total = 0.0
    for hyp, truths in zip(run, ref):
        total += max(rouge_l(hyp, t)[0] * 100 for t in truths)
    return round(total / len(run), 2)
