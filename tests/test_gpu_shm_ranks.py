"""GPU: the library's built-in shared-memory collectives (klr_ctx_init_shm) between real processes on one card:
all-reduce against numpy on awkward sizes, and the reduced solve with one dense-panel shard next to a shard that
is not one (the Gram path must be chosen jointly: ADVICE round 2, klr_solve.hip)."""
import ctypes as C
import os
import subprocess
import sys
from multiprocessing import shared_memory

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(case, world=2):
    import kmerlr_amd as K
    nbytes = K.lib().klr_shm_region_bytes(world)
    shm = shared_memory.SharedMemory(create=True, size=nbytes)
    try:
        addr = C.addressof(C.c_char.from_buffer(shm.buf))
        assert K.lib().klr_shm_region_init(C.c_void_p(addr), shm.size, world) == 0
        env = dict(os.environ, WORLD_SIZE=str(world), KLR_SHM_NAME=shm.name, PYTHONPATH=ROOT, KLR_SHM_TIMEOUT_S="60")
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shm_rank_worker.py"), case], env=dict(env, RANK=str(r)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
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
    finally:
        del addr
        shm.close()
        try:
            shm.unlink()
        except FileNotFoundError:
            pass


def test_shm_collectives_against_numpy():
    outs = run_ranks("collectives", world=3)
    assert "collectives ok" in outs[0]


def test_dense_panel_gram_is_chosen_jointly():
    outs = run_ranks("dense")
    assert "dense round 0 ok" in outs[0] and "dense round 1 ok" in outs[0]
