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


def run_iir_time_slice(filt, d_iq, start, stop, d_out, scratch):
    """Filter samples [start, stop) of one long device-resident CS16 stream `d_iq` (rows of (i, q) int16) with the IIR object
    `filt` as if it had filtered everything before `start`: the filter has a finite memory (IIR.memory_samples(): nothing of
    an older state is left above 1e-12), so it starts from rest that many samples earlier, its outputs over that halo go to
    `scratch`, and from `start` on it is in the whole stream's state -- a halo, not a state hand-off, and no collective.
    Returns the number of samples written to d_out."""
    mem = filt.memory_samples()
    if mem == 0:
        raise ValueError("this filter's memory is longer than the single-pass kernel's horizon: slice with a state hand-off")
    filt.state = None                                   # at rest
    h0 = max(start - mem, 0)
    if start > h0:
        filt.run(d_iq[h0:], start - h0, out=scratch)
    filt.run(d_iq[start:], stop - start, out=d_out)
    return stop - start


def fm_slice_phases(d_msg, slices, kf_hz, fs_hz, dist=None):
    """The modulator's phase (radians, wrapped) at the start of every time slice of ONE long message stream: an exclusive
    prefix over the slices' phase sums 2 pi kf / fs * sum(m) (fp64) -- 8 bytes per slice boundary.  Single process: all
    slices are summed here; with torch.distributed every rank sums its own slice and the sums meet in one all_gather of a
    double (the only exchange the sliced TX path has)."""
    import math
    import torch
    w = 2.0 * math.pi * kf_hz / fs_hz
    if dist is None:
        sums = [float(d_msg[a:b].double().sum()) * w for (a, b) in slices]
    else:
        a, b = slices[dist.get_rank()]
        mine = (d_msg[a:b].double().sum() * w).reshape(1)
        gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, mine)
        sums = [float(g) for g in gathered]
    out, acc = [], 0.0
    for v in sums:
        out.append(math.remainder(acc, 2.0 * math.pi))
        acc += v
    return out


def run_fm_time_slice(pipe, d_msg, start, stop, phase_at_start, kf_hz, fs_hz, d_bytes, scratch_bytes, halo=64):
    """FM-modulate, resample, quantise and pack messages [start, stop) of one long device-resident message stream with the TX
    pipe `pipe` as if it had seen everything before `start`: the modulator's phase there is `phase_at_start`
    (fm_slice_phases: the hand-off), the resampler's history is rebuilt from the `halo` messages before the slice (a
    multiple of the decimation M; their outputs go to scratch_bytes).  Returns the number of output words written."""
    import math
    from . import hip
    h0 = max(start - halo, 0)
    w = 2.0 * math.pi * kf_hz / fs_hz
    ph0 = phase_at_start - (float(d_msg[h0:start].double().sum()) * w if start > h0 else 0.0)
    pipe.seek(h0, math.remainder(ph0, 2.0 * math.pi))
    if start > h0:
        pipe.run(hip.TXPIPE_IN_FM_MESSAGE, d_msg[h0:], 0, start - h0, scratch_bytes, scratch_bytes.numel())
    no = pipe.out_count(stop - start)
    return pipe.run(hip.TXPIPE_IN_FM_MESSAGE, d_msg[start:], 0, stop - start, d_bytes, 4 * no)


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


_fan_comm = None


def _fanout_comm(dist, world, rank, device):
    """The process's RCCL point-to-point communicator behind the C ABI (clfan_*): rank 0 makes the id,
    torch.distributed only carries its 128 bytes to the other ranks."""
    global _fan_comm
    if _fan_comm is None:
        import torch
        from . import fanout
        ident = torch.tensor(fanout.Comm.unique_id() if rank == 0 else [0] * fanout.ID_BYTES, dtype=torch.uint8, device=device)
        if world > 1:
            dist.broadcast(ident, 0)
        _fan_comm = fanout.Comm(world, rank, ident.cpu().tolist())
    return _fan_comm


def fanout_streams(root_buf, n_streams, dist, world, rank, root=0, device=None, dtype=None, n_elems=None):
    """The one case with a real exchange step: `root` holds the raw SMI
    buffers of ALL streams ([n_streams, n_elems]) and hands every rank the streams it owns.

    xGMI is point-to-point (7 links per GPU), so this is a set of direct sends -- one per (peer, stream), all in
    one group so every link carries traffic concurrently -- not a ring broadcast, which would be bound by a single
    link.  On GPUs the exchange is the C ABI's (include/cariboulite_fanout.h: grouped ncclSend / ncclRecv, this
    module is only a caller); with a CPU backend (the gloo rehearsal of the logic) the same schedule is posted through
    torch.distributed.  Returns this rank's [n_local, n_elems] tensor.
    """
    import torch
    mine = assign_streams(n_streams, world, rank)
    if rank == root:
        n_elems, dtype, device = root_buf.shape[1], root_buf.dtype, root_buf.device
    local = torch.empty((len(mine), n_elems), dtype=dtype, device=device)
    on_gpu = torch.device(device).type == "cuda"
    if on_gpu and (dist is None or dist.get_backend() == "nccl"):
        comm = _fanout_comm(dist, world, rank, device)
        row = n_elems * local.element_size()
        comm.scatter(root, root_buf.data_ptr() if rank == root else None, row, row, n_streams,
                     local.data_ptr() if len(mine) else None, row, torch.cuda.current_stream().cuda_stream)
        return local
    # (gloo moves host memory only: device tensors of the one-GPU rehearsal -- every rank on GPU 0, where RCCL refuses -- go through
    # host copies of the messages; the schedule is the same)
    ops, landing = [], None
    if rank == root:
        for peer in range(world):
            rows = assign_streams(n_streams, world, peer)
            if not rows:
                continue
            block = root_buf[rows]                      # gather the peer's rows into one contiguous message
            if peer == root:
                local.copy_(block)
            else:
                ops.append(dist.P2POp(dist.isend, block.cpu() if on_gpu else block, peer))
    elif mine:
        landing = torch.empty(local.shape, dtype=dtype, device="cpu") if on_gpu else local
        ops.append(dist.P2POp(dist.irecv, landing, root))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if landing is not None and landing is not local:
        local.copy_(landing)
    return local


def gather_streams(local, n_streams, dist, world, rank, root=0):
    """Fan-in: every rank's per-stream results ([n_local, n_elems], same n_elems everywhere) back to `root` as
    [n_streams, n_elems] (None elsewhere).  GPU path = clfan_gather_streams (grouped ncclSend / ncclRecv)."""
    import torch
    n_elems = local.shape[1]
    out = torch.empty((n_streams, n_elems), dtype=local.dtype, device=local.device) if rank == root else None
    mine = assign_streams(n_streams, world, rank)
    if local.device.type == "cuda" and (dist is None or dist.get_backend() == "nccl"):
        comm = _fanout_comm(dist, world, rank, local.device)
        row = n_elems * local.element_size()
        comm.gather(root, local.data_ptr() if len(mine) else None, row, row, n_streams,
                    out.data_ptr() if rank == root else None, row, torch.cuda.current_stream().cuda_stream)
        return out
    ops, parts = [], {}
    on_gpu = local.device.type == "cuda"                # (a CPU backend under device tensors: messages through host copies, as in fanout_streams)
    if rank == root:
        for peer in range(world):
            rows = assign_streams(n_streams, world, peer)
            if not rows:
                continue
            if peer == root:
                out[rows] = local
            else:
                parts[peer] = torch.empty((len(rows), n_elems), dtype=local.dtype, device="cpu" if on_gpu else local.device)
                ops.append(dist.P2POp(dist.irecv, parts[peer], peer))
    elif mine:
        ops.append(dist.P2POp(dist.isend, local.cpu() if on_gpu else local.contiguous(), root))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for peer, blk in parts.items():
        out[assign_streams(n_streams, world, peer)] = blk.to(out.device)
    return out
