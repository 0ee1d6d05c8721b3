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


@pytest.fixture()
def setopt(monkeypatch):
    """setopt("KLR_X", value | None): a KLR_* switch for the rest of this test.  Contexts read the environment only when
    they are created (klr_ctx_create), so the switch is put into the environment (for contexts and processes created
    later) AND set on every live Context (klr_ctx_set_option); both are undone when the test ends."""
    import kmerlr_amd as K
    touched = []

    def _set(name, value):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(value))
        touched.append(name)
        for c in K.Context.live():
            c.set_option(name, value)
    yield _set
    monkeypatch.undo()
    for c in K.Context.live():
        c.reload_env()
