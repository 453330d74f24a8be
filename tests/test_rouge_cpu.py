"""ROUGE-L: the host-side implementation used by the acceptance harness against the fixture captured from the
reference's own evaluation/Rouge.py + evaluation/Eval_Rouge.py (tests/golden/rouge_l.npz)."""
import numpy as np
import torch

import cases
from helpers import load_golden


def test_rouge_l_matches_reference_fixture():
    import case_rg_amd
    rec = cases.CASES["rouge_l"](case_rg_amd.namespace(), torch.device("cpu"))
    golden = load_golden("rouge_l")
    assert np.allclose(rec["fpr"].numpy(), golden["fpr"], rtol=0, atol=1e-12)
    assert rec["rouge_l_f1"].item() == golden["rouge_l_f1"][0]


def test_lcs_edge_cases():
    from case_rg_amd.evaluation import lcs_length, rouge_l
    assert lcs_length([], ["a"]) == 0 and lcs_length(["a"], ["a"]) == 1
    assert lcs_length("a b c d".split(), "b d".split()) == 2
    assert lcs_length("x a y b z c".split(), "a b c".split()) == 3
    f, p, r = rouge_l("a b c", "a b c")
    assert abs(f - 1.0) < 1e-9 and p == 1.0 and r == 1.0
    assert rouge_l("a", "b")[0] == 0.0
