"""GPU: two real processes, each with part of the rows on the same card, exchange through the library's
all-reduce hook (gloo underneath): gradient bitwise equal to the single-process run, same selection, same solve.
Second case: the two shards disagree on the layout (one has the packed integer stream, one does not) and must
still post matching collectives (sweep_protocol, klr_sweep.hip)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_two(port, mode=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", PYTHONPATH=ROOT)
    if mode:
        env["KLR_TWO_RANK_MODE"] = mode
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-4000:])
    return outs


def test_two_ranks_one_gpu():
    outs = run_two(29533)
    assert "two ranks ok" in outs[0]


def test_two_ranks_mixed_layouts():
    outs = run_two(29537, "mixed")
    assert "two ranks ok (mixed layouts)" in outs[0]


def test_two_ranks_reduce_scatter_collective():
    env_extra = {"KLR_RS_MIN_DIM": "1"}
    os.environ.update(env_extra)
    try:
        outs = run_two(29541, "scatter")
    finally:
        for k in env_extra:
            os.environ.pop(k, None)
    assert "two ranks ok (scatter)" in outs[0]
