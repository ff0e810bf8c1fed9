"""Stream sharding and timing for one-process-per-GPU runs.

Channel streams are independent units (the reference models each channel as its
own Soapy device), so N GPUs need no data-path collective: stream s runs on
rank s mod N with all of its state (sync offsets, FIR / resampler history, IIR
and FM state).  torch.distributed (RCCL on GPUs, gloo in the CPU tests) carries
only the timing barrier and the max-over-ranks reduction.
"""
import time


def assign_streams(n_streams, world, rank):
    """Streams owned by `rank`: s with s mod world == rank (config 4: 256 streams -> 32 per GPU)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_streams, world))


def time_slices(n, world, align):
    """Cut ONE stream of n input samples into `world` contiguous slices whose starts are multiples of `align`
    (lcm of the decimation M and 2, so that every slice starts on polyphase phase 0 and on an even index);
    returns [(start, stop)] per rank.  Slice owners prime their pipe with the halo before `start`
    (RxPipe.seek + one discarded run), see clhip_rx_pipe_seek."""
    if world < 1 or align < 1:
        raise ValueError("world and align must be positive")
    per = -(-n // world)
    per = -(-per // align) * align
    out = []
    for r in range(world):
        a, b = min(r * per, n), min((r + 1) * per, n)
        out.append((a, b))
    return out


def run_time_slice(pipe, in_kind, d_in, start, stop, d_out, out_count_fn, scratch_out):
    """Process input samples [start, stop) of one long device-resident stream `d_in` with `pipe` as if the pipe
    had seen everything before `start`: reset, seek, run the halo (outputs discarded into scratch_out), run the slice
    into d_out.  Returns the number of outputs written."""
    halo = pipe.halo()
    pipe.reset()
    h0 = max(start - halo, 0)
    # a stream start inside the halo keeps the zero history of a fresh pipe for the missing part
    pipe.seek(h0)
    if start > h0:
        pipe.run(in_kind, d_in[h0:], 0, start - h0, scratch_out, 0)
    return pipe.run(in_kind, d_in[start:], 0, stop - start, d_out, 0)


def owner_of(stream, world):
    return stream % world


def timed_steps(step_fn, steps, sync_fn=None, dist=None, device=None):
    """Time EXACTLY `steps` calls of step_fn, bracketed by barrier + device sync on both
    sides; returns the MAX wall time over ranks (seconds)."""
    import torch
    sync_fn = sync_fn or (lambda: None)
    if dist is not None:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    if dist is not None:
        dist.barrier()
    sync_fn()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def job_throughput(units_per_rank_per_step, steps, dt_max, world):
    """Whole-job units per second: every rank processed the same per-step units (weak scaling)."""
    return world * units_per_rank_per_step * steps / dt_max


def fanout_streams(root_buf, n_streams, dist, world, rank, root=0, device=None, dtype=None, n_elems=None):
    """The one case with a real exchange step: `root` holds the raw SMI
    buffers of ALL streams ([n_streams, n_elems]) and hands every rank the streams it owns.

    xGMI is point-to-point (7 links per GPU), so this is a batch of direct sends -- one per
    (peer, stream block), all posted at once so every link carries traffic concurrently -- not a
    ring broadcast, which would be bound by a single link.  Returns this rank's [n_local, n_elems]
    tensor (the root keeps a view of its own rows).
    """
    import torch
    mine = assign_streams(n_streams, world, rank)
    if rank == root:
        n_elems, dtype, device = root_buf.shape[1], root_buf.dtype, root_buf.device
    local = torch.empty((len(mine), n_elems), dtype=dtype, device=device)
    ops = []
    if rank == root:
        for peer in range(world):
            rows = assign_streams(n_streams, world, peer)
            if not rows:
                continue
            block = root_buf[rows]                      # gather the peer's rows into one contiguous message
            if peer == root:
                local.copy_(block)
            else:
                ops.append(dist.P2POp(dist.isend, block, peer))
    elif mine:
        ops.append(dist.P2POp(dist.irecv, local, root))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return local
