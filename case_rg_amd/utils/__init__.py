from .fill import fill_params, make_vocab  # noqa: F401
from .synth import copy_task_batch, synth_batch  # noqa: F401
