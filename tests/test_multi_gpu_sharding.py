"""The N>1 control path of bench.py on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.
The data path has no collective (streams are independent), so what needs covering is the
barrier + max-over-ranks timing reduction and the stream -> device binding."""
import os
import socket
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import bench
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dist.barrier()
    elapsed = 0.5 if rank == 0 else 0.8            # rank 1 is the slow one
    mx = bench.reduce_max(dist, elapsed)
    dist.barrier()
    q.put((rank, mx, [s for s in range(8) if bench.stream_to_device(s, world) == rank]))
    dist.destroy_process_group()


def test_two_rank_timing_reduction_and_stream_binding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [0.8, 0.8]                       # every rank sees the slowest rank's time
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5, 7]  # stream s -> device s mod N, disjoint cover
    import bench
    # whole-job throughput: all ranks' pixels over the max time
    assert bench.whole_job_gpix(2, 1e9, 4, 0.8) == pytest.approx(2 * 1e9 * 4 / 0.8 / 1e9)
    assert bench.reduce_max(None, 1.25) == 1.25
