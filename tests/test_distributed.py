"""The N>1 path on CPU: world_size-2 gloo processes shard the environments and aggregate throughput exactly as bench.py
does on GPUs over RCCL (no data-path collective: only a SUM of units and a MAX of elapsed time)."""
import os
import socket
import sys

import torch.multiprocessing as mp

from helpers import REPO


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from muzero_amd import pipeline

    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = pipeline.shard_range(8191, rank, world)
    # rank r "plays" (hi - lo) envs x 50 sims x 3 moves in (1 + r) seconds
    rate, units, secs = pipeline.aggregate_throughput((hi - lo) * 50 * 3, 1.0 + rank)
    dist.barrier()
    q.put((rank, lo, hi, rate, units, secs, pipeline.rank_env()))
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, rate0, units0, secs0, env0), (r1, lo1, hi1, rate1, units1, secs1, env1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 4096, 4096, 8191)  # disjoint, contiguous, sizes differ by at most one
    assert units0 == units1 == 8191 * 150 and secs0 == secs1 == 2.0  # SUM of units, MAX of time
    assert rate0 == rate1 == 8191 * 150 / 2.0
    assert env0 == (0, 0, 2) and env1 == (1, 1, 2)


def test_shard_range_covers_everything():
    from muzero_amd.pipeline import shard_range

    for total in (1, 7, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
