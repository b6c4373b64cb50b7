import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "interpreting-video-features_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Parity tests run the plans with the built-in per-layer kernel choice: the autotuner picks by timing, so its choice
# (and with it the last bits of every sum) varies from box to box, and a gate that sits at the edge of the Adam
# trajectory's sensitivity would pass on one GPU and fail on the next.  What the tuner may pick is covered by
# test_every_conv_variant_matches_torch (each variant against torch fp64) and by
# test_tuned_plan_matches_builtin_choice (a tuned plan against the untuned one, whole network).
os.environ.setdefault("IVF_AUTOTUNE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]
    return load


def rel_err(a, b):
    """max |a-b| / max |b| over the whole tensor (the '1e-3 relative fp32' gate)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def ranking_consistent(rank_got, mask_want, tol):
    """A frame ranking (stable argsort of -mask) is an integer output, bit-exact where the mask
    separates the frames; frames whose reference mask values differ by less than `tol` (exact
    ties up to rounding: e.g. the interior of a run in a 'reverse' search, which only the
    regulariser moves) may come in any order.  True iff `rank_got` is a permutation that sorts
    `mask_want` non-increasingly up to `tol`."""
    rank_got = np.asarray(rank_got).astype(np.int64)
    mask_want = np.asarray(mask_want, dtype=np.float64)
    if sorted(rank_got.tolist()) != list(range(mask_want.size)):
        return False
    v = mask_want[rank_got]
    return bool(np.all(v[:-1] >= v[1:] - tol))


def rel_err_elem(a, b, floor):
    """max over elements of |a-b| / max(|b|, floor): every entry against its own magnitude."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def note(msg):
    """Measured error figures: printed (pytest -s) and appended to gpurun_out/parity_measured.log,
    the file the round's profiles/rNN_parity_measured.txt is copied from."""
    print("[parity]", msg)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_measured.log"), "a") as f:
            f.write(msg + "\n")
    except OSError:
        pass
