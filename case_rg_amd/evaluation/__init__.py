"""Answer-quality metrics of the reference's evaluation scripts that the acceptance bar names (ROUGE-L)."""
from .rouge import eval_rouge_l, lcs_length, rouge_l  # noqa: F401
