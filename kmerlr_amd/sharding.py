"""Row sharding helpers for one-process-per-GPU runs (samples shard naturally: every quantity on the
hot path is a sum over samples; DESIGN.md "multi-GPU")."""


def shard_rows(n, rank, world):
    """Contiguous, balanced row block [r0, r1) of rank `rank` out of `world`."""
    r0 = n * rank // world
    r1 = n * (rank + 1) // world
    return r0, r1


def init_rccl(ctx, rank, world, broadcast):
    """Gives the library its own RCCL communicator: rank 0 creates the ncclUniqueId, `broadcast`
    (callable: bytes-or-None -> bytes) distributes it, every rank joins.  The library then issues one
    all-reduce per sweep on its own stream with no Python in the loop."""
    from . import _capi
    uid = _capi.rccl_unique_id() if rank == 0 else None
    uid = broadcast(uid)
    ctx.init_rccl(uid, rank, world)
