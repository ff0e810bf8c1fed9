"""not-gpu: the N>1 path of bench.py (stream sharding, barrier, max-over-ranks) with
world_size-2 gloo on CPU.  The per-stream compute stand-in is the oracle's CPU pipe."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_streams, n, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from cariboulite_amd import shard, synth
    from oracle import oracle as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    taps = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "taps.npz"))
    mine = shard.assign_streams(n_streams, world, rank)
    pipes = {s: orc.RxPipeF32(0, taps["fir64_c2"], taps["rs_3_2"], 3, 2, n) for s in mine}
    data = {s: synth.smi_stream_bytes(n, 0, stream=s)[0] for s in mine}
    sums = {}

    def step():
        for s in mine:
            sums[s] = float(np.abs(pipes[s].run(data[s])).sum())

    dt = shard.timed_steps(step, 2, dist=dist)
    # every rank sees the same max-over-ranks time
    t = torch.tensor([dt], dtype=torch.float64)
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    q.put((rank, mine, sums, dt, float(lo), float(hi)))
    dist.destroy_process_group()


def test_two_rank_stream_sharding_gloo():
    import torch.multiprocessing as mp
    from cariboulite_amd import shard, synth
    from oracle import oracle as orc
    world, n_streams, n = 2, 5, 8192
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_streams, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(s for r in res for s in r[1])
    assert owned == list(range(n_streams))                       # every stream exactly once
    for rank, mine, sums, dt, lo, hi in res:
        assert mine == [s for s in range(n_streams) if shard.owner_of(s, world) == rank]
        assert lo == hi == dt and dt > 0                         # max-over-ranks agreed by all ranks
    # sharded results == single-process results: no data-path collective is needed
    taps = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "taps.npz"))
    allsums = {s: v for r in res for s, v in r[2].items()}
    for s in range(n_streams):
        p = orc.RxPipeF32(0, taps["fir64_c2"], taps["rs_3_2"], 3, 2, n)
        b = synth.smi_stream_bytes(n, 0, stream=s)[0]
        p.run(b)                                                  # two steps, state carried
        assert float(np.abs(p.run(b)).sum()) == allsums[s]
    assert shard.job_throughput(100, 2, 0.5, 2) == 800.0


def test_assign_streams_config4():
    from cariboulite_amd import shard
    for world in (1, 2, 4, 8):
        per = [shard.assign_streams(256, world, r) for r in range(world)]
        assert all(len(p) == 256 // world for p in per)
        assert sorted(sum(per, [])) == list(range(256))
    with pytest.raises(ValueError):
        shard.assign_streams(4, 2, 2)


def _fanout_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from cariboulite_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_streams, n = 5, 4096
    root_buf = None
    if rank == 0:
        root_buf = (torch.arange(n_streams * n, dtype=torch.int32).reshape(n_streams, n) * 7 + 3)
    local = shard.fanout_streams(root_buf, n_streams, dist, world, rank, root=0, device=torch.device("cpu"),
                                 dtype=torch.int32, n_elems=n)
    back = shard.gather_streams(local * 2, n_streams, dist, world, rank, root=0)        # fan-in of per-stream results
    q.put((rank, shard.assign_streams(n_streams, world, rank), local.numpy().copy(), back.numpy().copy() if rank == 0 else None))
    dist.destroy_process_group()


def test_fanout_streams_gloo():
    """Root holds every stream's raw buffer; each rank ends up with exactly the rows it owns."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_fanout_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = (np.arange(5 * 4096, dtype=np.int64).reshape(5, 4096) * 7 + 3).astype(np.int32)
    for rank, rows, local, back in res:
        assert np.array_equal(local, full[rows])
        if rank == 0:
            assert np.array_equal(back, full * 2)


def test_time_slices_cover_the_stream():
    from cariboulite_amd import shard
    for n, world, align in ((1_000_000, 3, 2), (131072, 8, 4), (10, 4, 4), (7, 1, 2)):
        sl = shard.time_slices(n, world, align)
        assert len(sl) == world and sl[0][0] == 0 and sl[-1][1] == n
        for (a, b), (c, d) in zip(sl, sl[1:]):
            assert b == c and a <= b
        assert all(a % align == 0 for a, _ in sl if a < n)


def _phase_worker(rank, world, port, n, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from cariboulite_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator(); g.manual_seed(3)
    msg = 0.3 * torch.randn(n, generator=g)                       # every rank can see the stream; it sums only its own slice
    slices = shard.time_slices(n, world, 3)
    q.put((rank, shard.fm_slice_phases(msg, slices, 75e3, 4e6, dist=dist)))
    dist.destroy_process_group()


def test_fm_slice_phase_hand_off_two_ranks_gloo():
    """The sliced TX path's one exchange: every rank sums its own slice's phase increments (fp64) and ONE all_gather of a
    double gives every rank the phase at the start of every slice -- the same values a single process computes."""
    import torch
    import torch.multiprocessing as mp
    from cariboulite_amd import shard
    world, n = 2, 300_003
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_phase_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator(); g.manual_seed(3)
    msg = 0.3 * torch.randn(n, generator=g)
    want = shard.fm_slice_phases(msg, shard.time_slices(n, world, 3), 75e3, 4e6)
    assert res[0] == res[1] == want and want[0] == 0.0 and want[1] != 0.0
