"""Shared comparison helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def to_np(t):
    return t.detach().float().cpu().numpy() if t.is_floating_point() else t.detach().cpu().numpy()


def compare(name, got, want, rtol, atol):
    """Compare one tensor with its golden value.  Integer / bool arrays must match exactly;
    floats within rtol*|want| + atol, with identical +-inf / NaN placement."""
    got = to_np(got) if torch.is_tensor(got) else np.asarray(got)
    assert got.shape == want.shape, "%s: shape %s != golden %s" % (name, got.shape, want.shape)
    if want.dtype.kind in "biu":
        assert np.array_equal(got, want), "%s: integer/bool mismatch" % name
        return 0.0
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), "%s: finite-ness pattern differs" % name
    assert np.array_equal(got[~fin], want[~fin], equal_nan=True), "%s: inf/nan values differ" % name
    err = np.abs(got[fin] - want[fin])
    tol = rtol * np.abs(want[fin]) + atol
    worst = float((err / np.maximum(tol, 1e-30)).max()) if err.size else 0.0
    assert worst <= 1.0, "%s: max err %.3e (%.2fx tolerance), max |want| %.3e" % (
        name, float(err.max()), worst, float(np.abs(want[fin]).max()))
    return worst


def check_case(name, rec, rtol, atol, grad_rtol=None, grad_atol=None, skip=(), override=None):
    """``override``: {key: (rtol, atol)} for single tensors that carry a documented looser bar."""
    golden = load_golden(name)
    assert set(rec) == set(golden), "case %s: keys differ: %s" % (name, set(rec) ^ set(golden))
    for k, want in golden.items():
        if k in skip:
            continue
        if override and k in override:
            compare(name + "/" + k, rec[k], want, *override[k])
        elif k.startswith("in_"):  # regenerated inputs must be the committed inputs
            compare(name + "/" + k, rec[k], want, 0.0, 0.0)
        elif k.startswith("g"):
            compare(name + "/" + k, rec[k], want, grad_rtol or rtol, grad_atol or atol)
        else:
            compare(name + "/" + k, rec[k], want, rtol, atol)


# ---------------------------------------------------------------------------------------------
# measured-error ledger: every GPU parity comparison records its worst error relative to the tensor's scale; conftest.py
# writes the ledger to gpurun_out/parity_errors.json at the end of the session (copied to profiles/ per round).
# ---------------------------------------------------------------------------------------------
ERRORS = {}


def scaled_error(name, got, want):
    """max |got - want| / (max |want| + 1e-6) over the finite entries, after checking shape and the inf / nan pattern."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), "%s: inf/nan pattern differs" % name
    if not fin.any():
        return 0.0
    return float(np.abs(got[fin] - want[fin]).max() / (np.abs(want[fin]).max() + 1e-6))


def l2_error(got, want):
    """||got - want|| / ||want|| over the finite entries."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    fin = np.isfinite(want)
    den = np.linalg.norm(want[fin])
    return float(np.linalg.norm(got[fin] - want[fin]) / den) if den > 0 else float(np.linalg.norm(got[fin]))


def record_error(case, mode, key, rel, tol, l2=None):
    rec = {"rel_err": float("%.3e" % rel), "tol": tol}
    if l2 is not None:
        rec["l2_err"] = float("%.3e" % l2)
    ERRORS.setdefault(case, {}).setdefault(mode, {})[key] = rec
