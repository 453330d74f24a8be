import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU-marked tests are skipped (not failed) on a box without a device, so a bare
    ``pytest tests`` in the CPU container stays green; ``-m gpu`` on the GPU box runs them."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Dump the measured parity errors of this session (GPU runs only record any)."""
    import json
    from helpers import ERRORS
    if not ERRORS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    worst = {c: {m: max((v["rel_err"] for v in keys.values()), default=0.0) for m, keys in modes.items()} for c, modes in ERRORS.items()}
    with open(os.path.join(out, "parity_errors.json"), "w") as fh:
        json.dump({"note": "max |got - reference| / max |reference| per tensor, HIP product on MI355X vs fixtures captured from the "
                           "reference (tests/golden/gen_golden.py); modes: see tests/test_parity_prod_gpu.py",
                   "worst_per_case_and_mode": worst, "tensors": ERRORS}, fh, indent=1, sort_keys=True)
