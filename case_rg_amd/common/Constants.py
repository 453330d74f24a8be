"""Special-token strings (reference: common/Constants.py:1-7; strings only, they are data)."""
PAD_WORD = '[PAD]'
BOS_WORD = '[unused0]'
UNK_WORD = '[UNK]'
EOS_WORD = '[unused1]'
SEP_WORD = '[SEP]'
CLS_WORD = '[CLS]'
MASK_WORD = '[MASK]'
