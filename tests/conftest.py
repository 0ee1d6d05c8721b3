import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def scores_data():
    """scoresLr_test_{fg,bg}.table as the reference compiles them
    (compile_training_data_scores, scoresLr_data.go:248-258: fg rows first, labels true)."""
    import numpy as np
    from oracle import oracle as O
    _, fg = O.read_scores_table(os.path.join(GOLDEN, "scoresLr_test_fg.table"))
    _, bg = O.read_scores_table(os.path.join(GOLDEN, "scoresLr_test_bg.table"))
    x = np.concatenate([fg, bg], axis=0)
    labels = np.array([1] * len(fg) + [0] * len(bg), dtype=np.uint8)
    return O.Csr.from_dense(x), labels, x
