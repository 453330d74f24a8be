"""CPU restatement of the reference's ROUGE-L (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).

evaluation/Rouge.py:83-108 (_lcs: full DP table), :65-80 (_len_lcs), :186-206 (_f_p_r_lcs), :209-245 (rouge_l_sentence_level),
evaluation/Eval_Rouge.py:13-22 (max over ground truths), :49-68 (cal_rouge / eval_rouge: x100, mean, round to 2)."""


def _len_lcs(x, y):
    n, m = len(x), len(y)
    table = {}
    for i in range(n + 1):
        for j in range(m + 1):
            if i == 0 or j == 0:
                table[i, j] = 0
            elif x[i - 1] == y[j - 1]:
                table[i, j] = table[i - 1, j - 1] + 1
            else:
                table[i, j] = max(table[i - 1, j], table[i, j - 1])
    return table[n, m]


def rouge_l(hypothesis, reference):
    ref_words = reference.split(" ")
    hyp_words = hypothesis.split(" ")
    llcs = _len_lcs(hyp_words, ref_words)
    r_lcs = llcs / len(ref_words)
    p_lcs = llcs / len(hyp_words)
    beta = p_lcs / (r_lcs + 1e-12)
    f_lcs = ((1 + beta ** 2) * r_lcs * p_lcs) / (r_lcs + (beta ** 2) * p_lcs + 1e-12)
    return f_lcs, p_lcs, r_lcs


def eval_rouge_l(run, ref):
    total = 0.0
    for i, pre in enumerate(run):
        total += max(rouge_l(pre, truth)[0] * 100 for truth in ref[i])
    return round(total / len(run), 2)
