#!/usr/bin/env python3
"""bench.py -- Msamples/s through the RX pipe unpack + FIR(64) + resample(3/2)
(BASELINE.json metric, config 2) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) it launches that second form itself: the parent
touches no GPU (it has not even imported torch), starts torch.distributed.run as a child process on 127.0.0.1 with a
free port, relays the child's output -- rank 0's JSON line -- and exits with its code.

A "step" is one pass of the hot path over one batch of synthetic SMI bytes that
is already resident in HBM: ONE launch of the fused kernel (which also verifies the
sync words of each of the 2048 native 512 KiB chunks, caribou_smi.c:235-292) over a
2^28-sample stream (1 GiB in, 3 GiB CF32 out -- larger than the 256 MiB
Infinity Cache).  Streams are independent, so N GPUs run N such streams with
no data-path collective (weak scaling); the only collectives are the timing
barrier and the max-over-ranks reduction.

--workload c1|c3|c4|c5|iir|tags times the other configs of BASELINE.json (the a6 IIR, the pps tag compaction) through the same contract;
the default (c2) is the headline metric.  --pcie prints the PCIe-INCLUSIVE companion of the headline line in the same JSON shape
(host SMI bytes in, host samples out, --pcie-streams Soapy devices per GPU through cl_group_readStream; roofline.bound "pcie"
against the ceiling measured in the same run): reported beside the headline, never instead of it.

One JSON line on rank 0.  `roofline` prices the fused kernel against HBM
(16 algorithmic bytes per input sample: 4 read + 12 written); `cpu_baseline`
is the oracle's fp32 CPU pipe (oracle/cl_oracle.c, "port") on this host's cores.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "active")     # libgomp barriers spin (sandboxed futexes are slow)
# torch moves this script's numpy arrays to and from the device; with the threshold (MiB) out of reach the HIP runtime stages
# pageable copies >= 1 MiB through its own pinned buffers instead of pinning numpy's heap pages in place.
# A tool's own setting: the product libraries neither set nor need it.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

import numpy as np  # noqa: E402

torch = None          # imported by main() once it is clear that THIS process is a rank (the self-launching parent never does)

ALGO_BYTES_PER_SAMPLE = 16.0      # R 4 B + W 8*3/2 B
FLOP_PER_SAMPLE = 304.0           # 64*2*2 (FIR) + 1.5*8*2*2 (3/2 polyphase)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3
NATIVE_CHUNK_SAMPLES = 131072     # caribou_smi.c:78 / cariboulite_radio.c:1310-1315


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--log2-samples", type=int, default=28, help="samples per GPU per step (2^k)")
    ap.add_argument("--samples", type=int, default=0, help="c2 only: samples per GPU per step, any multiple of 4 (e.g. 4000000 = one second of "
                    "one stream, SURVEY.md section 8d's cache-resident case); overrides --log2-samples")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--settle", type=int, default=40, help="untimed steps before the warm-up (DVFS settle)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal on one GPU)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c4", "c1", "c3", "c5", "iir", "tags"],
                    help="c2 (default, the BASELINE.json metric) | c4: 256 streams x 2^24, FIR128 + 5/4, sharded (strong "
                         "scaling) | c1 / c3 / c5 / iir: the other BASELINE.json configs and the a6 filter, one stream set "
                         "per GPU (weak scaling), same JSON shape with their own algorithmic bytes")
    ap.add_argument("--streams", type=int, default=256, help="c4 only: total streams (32 = the share of one GPU of the 8-GPU job)")
    ap.add_argument("--fanout", action="store_true", help="c4 only: rank 0 holds all raw buffers and scatters them over xGMI first")
    ap.add_argument("--pcie", action="store_true",
                    help="c2 only: the PCIe-INCLUSIVE figure instead of the headline one -- host SMI bytes in, host CF32 samples out, "
                         "--pcie-streams Soapy devices of one GPU read through cl_group_readStream (one MTU per stream and step); "
                         "roofline.bound = \"pcie\" against the box's own concurrent H2D + D2H ceiling.  Never the headline `value`.")
    ap.add_argument("--pcie-streams", type=int, default=32)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even at world size 1 (under torch.distributed.run --nproc-per-node 1): the "
                         "RCCL calls of the N > 1 path -- init with a device id, barrier, max-reduction on a device tensor, destroy -- "
                         "on a one-GPU box")
    ap.add_argument("--no-pcie-extra", action="store_true",
                    help="default run only: do not append the PCIe-inclusive figure (`pcie_inclusive`, a child `bench.py --pcie` behind the timed region)")
    ap.add_argument("--print-launch", action="store_true",
                    help="with --gpus N > 1 and no WORLD_SIZE: print the launcher command this process would start, and exit")
    return ap.parse_args()


def self_launch_argv(n_gpus, argv, port):
    """The command a plain `python bench.py --gpus N ...` turns itself into: one rank per GPU of this node over
    torch.distributed.run, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def self_launch(a):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [x for x in sys.argv[1:] if x != "--print-launch"]
    cmd = self_launch_argv(a.gpus, argv, port)
    if a.print_launch:
        print(json.dumps({"argv": cmd, "torch_imported": "torch" in sys.modules,
                          "hip_imported": "cariboulite_amd.hip" in sys.modules}), flush=True)
        return 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a child process, never an exec: this process stays the one the caller waits for and times
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(taps, d_words, budget_s):
    """Oracle fp32 pipe on host cores over a bounded sample of the SAME bytes."""
    from oracle import oracle as orc
    orc.lib()
    n = 1 << 24                                    # 16.8 M samples = 128 native chunks (4.2 s of stream)
    b = d_words[:n].cpu().numpy().view(np.uint8)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)             # a one-GPU box owns a 16-CPU share of the host
    out, bufs = orc.rx_pipe_f32_mt(0, b, taps["fir64_c2"], taps["rs_3_2"], 3, 2, 1)          # warm (page faults)
    res = {}
    for label, nt in (("1t", 1), ("all", cores)):   # 1 thread first: libgomp's pool spins once it exists
        reps, t0 = 0, time.perf_counter()
        while True:
            out, bufs = orc.rx_pipe_f32_mt(0, b, taps["fir64_c2"], taps["rs_3_2"], 3, 2, nt, bufs=bufs)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget_s * (0.6 if label == "all" else 0.4):
                break
        res[label] = reps * n / dt / 1e6
    # (i) of SURVEY.md 8(d): the reference-semantics plumbing alone (chunked sync search + unpack + /4096), one thread
    reps, t0 = 0, time.perf_counter()
    while True:
        _, iq, _ = orc.smi_read(0, b, n, 524288)
        _ = orc.cs16_to_cf32(iq)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 1.5:
            break
    res["unpack_1t"] = reps * n / dt / 1e6
    # the same plumbing through the REFERENCE's own code where it has been compiled (oracle/_ref/libref_smi.so = its caribou_smi.c:
    # caribou_smi_read's chunk loop -- read() of 512 KiB native batches from a file that stands where /dev/smi stands, sync search,
    # caribou_smi_rx_data_analyze), then the plugin's /4096 loop (restated: the Soapy plugin itself cannot be compiled here); one thread
    ref_1t = None
    if orc.have_ref():
        import ctypes as C
        import tempfile
        try:
            with tempfile.NamedTemporaryFile(suffix=".smi", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as f:
                f.write(b.tobytes()); f.flush()
                iq = np.zeros((n + 2, 2), np.int16)
                meta = np.zeros(n + 2, np.uint8)
                reps, t0 = 0, time.perf_counter()
                while True:
                    ret = orc.ref().ref_smi_read_file(f.name.encode(), 0, orc._p(iq, C.c_int16), orc._p(meta, C.c_uint8), C.c_size_t(n), C.c_size_t(524288))
                    if ret <= 0:
                        raise RuntimeError(f"caribou_smi_read returned {ret}")
                    _ = orc.cs16_to_cf32(iq[:n])
                    reps += 1
                    dt = time.perf_counter() - t0
                    if dt > 1.5:
                        break
                ref_1t = round(reps * n / dt / 1e6, 1)
        except Exception as e:                                 # (a reported extra: the line stands without it)
            ref_1t = f"{type(e).__name__}: {e}"[:120]
    return {"value": round(res["all"], 1), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "value_1_thread": round(res["1t"], 1), "unpack_scale_only_1_thread": round(res["unpack_1t"], 1),
            "reference_caribou_smi_read_plus_scale_1_thread": ref_1t,
            "sample": f"first 2^24 samples (128 native chunks) of the GPU input, oracle/cl_oracle.c "
                      f"orc_rx_pipe_f32_mt (unpack+sync -> /4096 -> FIR64 -> 3/2, fp32 AVX2, OpenMP)"}, out


def pmc_traffic(workload):
    """HBM bytes per step measured with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections;
    tools/collect_workload_profiles.sh) and committed under profiles/: (bytes, source) or (None, None)."""
    for rnd in ("r04", "r03", "r02", "r01"):
        f = os.path.join(ROOT, "profiles", rnd, f"{workload}_pmc.json")
        if os.path.exists(f):
            try:
                d = json.load(open(f))["_derived"]
                return round(d.get("traffic_bytes_per_step", d.get("traffic_bytes_per_launch"))), f"profiles/{rnd}/{workload}_pmc.json"
            except Exception:
                pass
    return None, None


def bench_pcie(a, world, rank, dev, dist, red_dev, arch, taps):
    """Config 2's stages at the drop-in boundary, PCIe included: every rank feeds --pcie-streams Soapy devices of its GPU with host
    SMI bytes (cl_smi_feed_bytes: their pinned FIFOs stand where /dev/smi's kfifo stands) and reads them through cl_group_readStream
    into pageable numpy buffers, one MTU per stream and step (soapy_api/CaribouliteStreamFunctions.cpp:239-254 x N devices,
    SoapyCariboulite.cpp:46-69).  Same JSON shape as the headline line; the roof is the PCIe link, measured on this box in this
    run (tools/microbench/pcie_duplex: the same bytes in and out on two streams, pinned memory, copy engine)."""
    from cariboulite_amd import soapy as S, synth, shard
    MTU, NB = NATIVE_CHUNK_SAMPLES, 4 * NATIVE_CHUNK_SAMPLES
    n, K, W = a.pcie_streams, a.steps, a.warmup
    words = [synth.smi_stream_bytes((K + W) * MTU, i % 2, stream=1000 * rank + i)[0] for i in range(min(n, 4))]

    def leg(registered):
        devs = []
        for i in range(n):
            d = S.Device(dict(driver="Cariboulite", channel="S1G" if i % 2 == 0 else "HiF", gpu=str(dev.index or 0)))
            d.activateStream(d.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, args={"FIR": "64:1000000", "RESAMP": "3/2"}))
            devs.append(d)
        # (every step's bytes are queued beforehand: room for all of them in the group's pinned slab, so that the members' FIFOs stay in it)
        grp = S.Group(devs, {"SLAB_MB": str(((K + W) * NB >> 20) + 1)})
        bufs = [np.zeros((MTU * 3 // 2 + 8, 2), np.float32) for _ in range(n)]
        if registered:
            grp.registerBuffers(bufs)                       # cl_group_register_buffers: the launches store into the clients' buffers themselves
        for i, d in enumerate(devs):
            d.feedSmiBytes(words[i % len(words)])

        def step():
            nd, rets = grp.readStream(bufs, MTU)
            assert nd == n, (nd, grp.lastError())

        for _ in range(W):
            step()
        t = shard.timed_steps(step, K, sync_fn=torch.cuda.synchronize, dist=dist, device=red_dev)
        stats = grp.stats()
        first = bufs[0][: MTU * 3 // 2].copy()
        grp.close()
        for d in devs:
            d.close()
        return t, stats, first

    dt, st, got0 = leg(False)
    dt_reg, st_reg, got0_reg = leg(True) if not a.no_pcie_extra else (None, None, None)
    if rank == 0:
        value = world * n * MTU * K / dt / 1e6
        res = {"metric": "Msamples/s through unpack+FIR(64)+resample(3/2) pipe, host bytes in -> host samples out (PCIe-inclusive)",
               "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": W,
               "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"config 2's stages at the Soapy boundary: {n} streams per GPU (S1G + HiF), one MTU = {MTU} samples per stream and step "
                                      f"through cl_group_readStream, host SMI bytes in (pinned FIFOs), CF32 samples out into pageable buffers",
                          "streams_per_gpu": n, "arch": arch, "group": {k: st[k] for k in ("batched_reads", "single_reads", "launches", "copies_2d", "errors")},
                          "parallelism": f"{world} independent group(s), no data-path collective"}}
        exe = os.path.join(ROOT, "tools", "microbench", "pcie_duplex")
        sh = None
        if os.path.exists(exe):
            try:
                o = subprocess.run([exe, str(n * NB >> 20), str(n * MTU * 12 >> 20)], capture_output=True, text=True, timeout=120, check=True).stdout
                sh = json.loads(o[o.index("{"):])["shape"]
            except Exception:
                sh = None
        per_gpu = value / world
        if dt_reg is not None:
            # the same calls with the clients' buffers registered beforehand (cl_group_register_buffers): no mirror, no last-hop memcpy on
            # the host -- the figure that does not depend on how busy the box's cores are; never `value`
            assert got0_reg.tobytes() == got0.tobytes(), "registered route delivered other samples than the default route"
            res["registered_client_buffers"] = {"value": round(world * n * MTU * K / dt_reg / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt_reg / K * 1e3, 4),
                                                "direct_reads": st_reg.get("direct_reads"), "same_samples_as_default_route": True}
        if sh:
            peak = n * MTU / (sh["duplex_ms"] * 1e-3) / 1e6
            res["roofline"] = {"bound": "pcie", "achieved": round(per_gpu, 1), "peak": round(peak, 1), "unit": "Msamples/s per GPU", "frac": round(per_gpu / peak, 4),
                               "achieved_GBs": round(16.0 * per_gpu / 1e3, 2), "peak_GBs": round(16.0 * peak / 1e3, 2), "traffic": None,
                               "peak_note": f"measured here: {sh['in_MiB']} MiB H2D + {sh['out_MiB']} MiB D2H queued together on two HIP streams (pinned memory, copy engine) "
                                            f"take {sh['duplex_ms']:.3f} ms; alone {sh['h2d_GBs']} / {sh['d2h_GBs']} GB/s",
                               "kernel": "rx_pipe_fused_kernel over sub-batches of 4 streams, stores into the mapped pinned mirror",
                               "algorithmic_bytes_per_sample": 16.0}
            if dt_reg is not None:
                res["registered_client_buffers"]["frac"] = round(res["registered_client_buffers"]["value"] / world / peak, 4)
        else:
            res["roofline"] = {"bound": "pcie", "achieved": round(per_gpu, 1), "peak": None, "unit": "Msamples/s per GPU", "frac": None, "traffic": None,
                               "peak_note": "tools/microbench/pcie_duplex not built: __graft_entry__.build() compiles it"}
        if world == 1 and not a.no_cpu:
            n_cpu = 1 << 24
            dwords = torch.from_numpy(np.tile(words[0][: min(words[0].size, 4 * n_cpu)], -(-4 * n_cpu // words[0].size))[: 4 * n_cpu].view(np.int32).copy())
            cb, cpu_out = cpu_baseline(taps, dwords, a.cpu_seconds)
            res["cpu_baseline"] = cb
            res["gpu_over_cpu"] = round(value / cb["value"], 2)
            # parity of the delivered samples: stream 0's last batch against the oracle's pipe over the same stream (carried state included)
            from oracle import oracle as orc
            fir, rs = orc.FIR(taps["fir64_c2"]), orc.Resampler(taps["rs_3_2"], 3, 2)
            want = None
            for k in range(K + W):
                _, iq, _ = orc.smi_read(0, words[0][k * NB:(k + 1) * NB], MTU, NB)
                want = rs.f64(fir.f64(orc.cs16_to_cf32(iq[:MTU])))
            res["max_abs_diff_vs_oracle_last_batch"] = float(np.max(np.abs(got0[: want.shape[0]] - want)))
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def bench_c4(a, world, rank, dev, dist, red_dev, arch, taps):
    """Config 4 (secondary, not the BASELINE metric): 256 independent 4 MS/s streams, FIR128 + 5/4,
    stream s on rank s mod N, no data-path collective; --fanout adds the one real exchange step
    (root scatters the raw buffers with direct sends) and reports it separately."""
    from cariboulite_amd import hip, synth, shard
    n_streams = a.streams
    n = 1 << (24 if a.log2_samples == 28 else a.log2_samples)
    mine = shard.assign_streams(n_streams, world, rank)
    fan_s = None
    if a.fanout and dist is not None:
        root_buf = None
        if rank == 0:
            root_buf = torch.stack([synth.torch_smi_words(n, dev, 0, 100 + s) for s in range(n_streams)])
        torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
        words = shard.fanout_streams(root_buf, n_streams, dist, world, rank, 0, dev, torch.int32, n)
        torch.cuda.synchronize(); dist.barrier(); fan_s = time.perf_counter() - t0
        del root_buf
    else:
        words = torch.stack([synth.torch_smi_words(n, dev, 0, 100 + s) for s in mine])
    pipe = hip.RxPipe(len(mine), hip.CHANNEL_S1G, taps["fir128_c4"], taps["rs_5_4"], 5, 4, hip.PIPE_OUT_IQ)
    no = pipe.out_count(n)
    out = torch.empty((len(mine), no, 2), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        pipe.run(hip.PIPE_IN_SMI_WORDS, words, n, n, out, no, stream)

    for _ in range(a.settle + a.warmup):
        step()
    dt = shard.timed_steps(step, a.steps, sync_fn=torch.cuda.synchronize, dist=dist, device=red_dev)
    if rank == 0:
        value = n_streams * n * a.steps / dt / 1e6
        print(json.dumps({
            "metric": f"Msamples/s through unpack+FIR(128)+resample(5/4) pipe, {n_streams} streams", "value": round(value, 1),
            "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config 4: {n_streams} streams x 2^{int(np.log2(n))} samples, FIR128 + 5/4, stream s on rank s mod N",
                       "streams_per_gpu": len(mine), "arch": arch, "fanout_scatter_s": fan_s},
            "roofline": {"bound": "hbm", "achieved": round(14.0 * n_streams * n * a.steps / dt / 1e9 / world, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                         "frac": round(14.0 * n_streams * n * a.steps / dt / 1e9 / world / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic("c4")[0] if (n_streams == 32 * world and n == 1 << 24) else None,
                         "traffic_source": pmc_traffic("c4")[1], "kernel": "rx_pipe_fused_kernel<PipeCfg<128,5,4,8,MODE_IQ,16,256,FFA>>",
                         "algorithmic_frac_of_fp32_valu_peak": round(552.0 * n_streams * n * a.steps / dt / 1e12 / world / VALU_PEAK_TFLOPS, 4)}}),
            flush=True)
    if dist is not None:
        dist.destroy_process_group()


def bench_secondary(a, world, rank, dev, dist, red_dev, arch, taps):
    """The other configs of BASELINE.json (parity-test cases, not the headline metric) through the same timing
    contract: one independent copy of the workload per GPU, no data-path collective."""
    from cariboulite_amd import hip, synth, shard
    stream = torch.cuda.current_stream().cuda_stream
    checks = []                                   # verdicts asked after the timed region (a failed one voids the line)
    if a.workload == "c1":
        n = 1 << a.log2_samples
        nch = max(n // NATIVE_CHUNK_SAMPLES, 1)
        words = synth.torch_smi_words(n, dev, 0, rank)
        offs = torch.zeros(nch, dtype=torch.int32, device=dev)
        out = torch.empty((n, 2), dtype=torch.float32, device=dev)
        def step():
            hip.smi_find_offsets(words, 4 * n, 524288, 524288, nch, offs, stream)
            hip.smi_unpack(0, words, 4 * n, 524288, 524288, nch, offs, hip.FORMAT_CF32, out, None, stream)
        units, bytes_per, metric = n, 12.0, "Msamples/s through sync check + int13 unpack + /4096 (CF32 out)"
        desc = f"config 1 on the GPU: one 2^{a.log2_samples}-sample SMI buffer, chunked sync check + unpack -> CF32"
        kern = "smi_find_offsets_kernel + smi_unpack_kernel<CF32>"
    elif a.workload == "c3":
        n = 1 << (a.log2_samples - 1)
        w2 = [synth.torch_smi_words(n, dev, ch, 2 * rank + ch) for ch in (0, 1)]
        pipes = [hip.RxPipe(1, ch, taps["fir64_c3"], None, 1, 1, hip.PIPE_OUT_FM_DEMOD) for ch in (0, 1)]
        outs = [torch.empty(n, dtype=torch.float32, device=dev) for _ in (0, 1)]
        def step():
            for ch in (0, 1):
                pipes[ch].run(hip.PIPE_IN_SMI_WORDS, w2[ch], 0, n, outs[ch], 0, stream)
        units, bytes_per, metric = 2 * n, 8.0, "Msamples/s through unpack+FIR(64)+FM demod, S1G + HiF"
        desc = f"config 3: two channels (S1G, HiF) x 2^{a.log2_samples - 1} samples, FIR64 + phase-difference FM demod, fp32 out"
        kern = "rx_pipe_fused_kernel<PipeCfg<64,1,1,1,MODE_FM,16,256,FFA>> x2"
    elif a.workload == "c5":
        n = 1 << (a.log2_samples - 1)
        msg = torch.randn(n, device=dev) * 0.3
        pipe = hip.TxPipe(1, 75e3, 4e6, taps["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
        no = pipe.out_count(n) + 4
        by = torch.empty(4 * no, dtype=torch.uint8, device=dev)
        def step():
            pipe.run(hip.TXPIPE_IN_FM_MESSAGE, msg, 0, n, by, 4 * no, None, 0, stream)
        units, bytes_per, metric = n, 4.0 + 8.0 / 3.0, "Msamples/s through FM mod + 2/3 resample + int13 pack (TX)"
        desc = f"config 5: 2^{a.log2_samples - 1} fp32 messages -> FM modulate -> 2/3 polyphase -> (int16)(f*4096) -> SMI TX words"
        kern = "tx_fm_chain_kernel<TxCfg<2,3,8>>"
    elif a.workload == "tags":
        n = 1 << a.log2_samples
        meta = ((torch.arange(n, device=dev) % 4_000_000) == 0).to(torch.uint8)         # one pps marker per second of stream
        ws = torch.empty(hip.lib().clhip_sync_tags_ws_bytes(n), dtype=torch.uint8, device=dev)
        idx = torch.empty(4096, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        def step():
            hip.sync_tags(meta, n, idx, 4096, cnt, ws, stream)
        units, bytes_per, metric = n, 1.0, "Msamples/s through the pps tag compaction of the meta plane"
        desc = f"GNU Radio source's tag loop (caribouLiteSource_impl.cc:113-119): 2^{a.log2_samples} meta bytes -> ordered positions of meta == 1"
        kern = "sync_tags_count_kernel + sync_tags_emit_kernel"
        checks.append(lambda: int(cnt.item()) == (n + 3_999_999) // 4_000_000)
    else:
        n = 1 << (a.log2_samples - 2)
        iq = torch.randint(-4096, 4096, (n, 2), dtype=torch.int16, device=dev)
        from cariboulite_amd import soapy as S
        f = hip.IIR(S.design_butter_lowpass(6, 4e6, 50e3))        # the reference's 100 kHz-bandwidth filter (fc = bw / 2)
        def step():
            f.run(iq, n, None, stream)
        units, bytes_per, metric = n, 8.0, "Msamples/s through the Butterworth-6 IIR on CS16 (fp64, in place)"   # in place: R 4 + W 4
        desc = f"a6: 2^{a.log2_samples - 2} CS16 samples filtered in place, both rails, state carried"
        kern = "iir_rail_kernel<3, 64, true> (one launch per step)"
        checks.append(lambda: f.status() == 0 and not f.on_scan_path())
    for _ in range(a.settle + a.warmup):
        step()
    dt = shard.timed_steps(step, a.steps, sync_fn=torch.cuda.synchronize, dist=dist, device=red_dev)
    if not all(c() for c in checks):
        raise SystemExit(f"bench.py --workload {a.workload}: a timed step reported an overrun; the line would be invalid")
    if rank == 0:
        value = world * units * a.steps / dt / 1e6
        ach = bytes_per * units * a.steps / dt / 1e9
        print(json.dumps({
            "metric": metric, "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"iir": "f64", "tags": "u8"}.get(a.workload, "f32"), "data": "synthetic",
            "config": {"workload": desc, "arch": arch, "parallelism": f"{world} independent copy(ies), no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                         "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic(a.workload)[0] if a.log2_samples == 28 else None, "traffic_unit": "HBM bytes per step",
                         "traffic_source": pmc_traffic(a.workload)[1], "kernel": kern,
                         "algorithmic_bytes_per_sample": round(bytes_per, 3),
                         "note": "whole step (all launches of the workload), HIP-event free: wall clock of the timed region"}}),
            flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a)
    global torch
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with --nproc-per-node {a.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    if os.environ.get("CLHIP_BENCH_ALL_ON_GPU0"):          # rehearsal of the N>1 path on a one-GPU box (gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist, red_dev = None, dev
    if world > 1 or (a.force_dist and "RANK" in os.environ):
        import torch.distributed as dist
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL; used only for barrier + max
        else:
            # (gloo's C++ side prints its "Rank 0 is connected to ..." lines on stdout: point fd 1 at stderr while it
            # connects, so that rank 0's JSON line stays the only thing on stdout)
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(a.dist_backend)
                dist.barrier()
            finally:
                os.dup2(keep, 1)
                os.close(keep)
            red_dev = torch.device("cpu")
        # build the communicator NOW: a lazy first barrier right before the timed region would idle the GPU
        # for seconds and restart the DVFS transient
        dist.barrier()
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=red_dev), op=dist.ReduceOp.MAX)

    from cariboulite_amd import hip, synth
    arch = hip.require_gpu()
    taps = np.load(os.path.join(ROOT, "tests", "golden", "taps.npz"))
    if a.pcie:
        if a.workload != "c2":
            raise SystemExit("--pcie is config 2's stages at the Soapy boundary: --workload c2")
        if a.steps == 100 and a.warmup == 30:
            a.steps, a.warmup = 40, 10                  # (every step's bytes are queued in the members' pinned FIFOs beforehand: 512 KiB per stream and step)
        return bench_pcie(a, world, rank, dev, dist, red_dev, arch, taps)
    if a.workload == "c4":
        return bench_c4(a, world, rank, dev, dist, red_dev, arch, taps)
    if a.workload != "c2":
        return bench_secondary(a, world, rank, dev, dist, red_dev, arch, taps)

    n = a.samples if a.samples > 0 else 1 << a.log2_samples
    n_chunks = -(-n // NATIVE_CHUNK_SAMPLES)
    words = synth.torch_smi_words(n, dev, channel=0, stream=rank)             # int32 RX words in HBM
    pipe = hip.RxPipe(1, hip.CHANNEL_S1G, taps["fir64_c2"], taps["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    n_out = pipe.out_count(n)
    out = torch.empty((n_out, 2), dtype=torch.float32, device=dev)
    offs = torch.full((max(n_chunks, 1),), -1, dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    # the sync check of every native chunk rides inside the fused kernel (four scalar loads per tile)
    pipe.set_sync_check(None, NATIVE_CHUNK_SAMPLES, bad)
    assert pipe.uses_fused(n), "fused gfx950 kernel not selected"
    stream = torch.cuda.current_stream().cuda_stream
    L = hip.lib()

    def step():
        got = pipe.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0, stream)
        assert got == n_out

    # Everything the timed region needs is created BEFORE the warm-up: the kernel is VALU-bound and the
    # chip is power-managed -- after >= 50 ms of idle the first steps run boosted (~0.9 ms), the clock
    # then dips (~1.4 ms) and takes ~20 steps to settle (tools/dvfs_probe.py).  A
    # plain synchronize does not trigger that, a host-side pause does, so nothing slow sits between the
    # warm-up and the timed steps, and an untimed settle phase precedes the W warm-up steps: `value` is
    # the SUSTAINED rate whatever K and W the caller picks.
    from cariboulite_amd import shard
    ev0, ev1 = L.clhip_event_create(), L.clhip_event_create()
    for _ in range(a.settle):
        step()
    for _ in range(a.warmup):
        step()
    counter = [0]

    def timed_step():
        # HIP events on the launch stream bracket the timed region itself: the first step records the start event in
        # front of its launch, the last one the stop event behind it.  Their distance / K is the fused kernel's
        # average launch duration including the dispatch gap between back-to-back launches (an event pair around
        # EVERY launch puts two more packets between kernels and lengthens the very thing it measures).
        k = counter[0]; counter[0] += 1
        if k == 0:
            L.clhip_event_record(ev0, stream)
        pipe.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0, stream)
        if k == a.steps - 1:
            L.clhip_event_record(ev1, stream)

    # barrier + synchronize on both sides, EXACTLY `steps` steps, max over ranks
    dt = shard.timed_steps(timed_step, a.steps, sync_fn=torch.cuda.synchronize, dist=dist, device=red_dev)
    kern_avg_s = float(L.clhip_event_elapsed_ms(ev0, ev1)) / a.steps / 1e3
    # per-launch durations (min, spread) from a second, untimed pass with an event pair around every launch, queued
    # right behind the timed region (a host-side pause in between would restart the DVFS transient)
    evs = [(L.clhip_event_create(), L.clhip_event_create()) for _ in range(min(a.steps, 10))]
    for e0, e1 in evs:
        L.clhip_event_record(e0, stream)
        pipe.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0, stream)
        L.clhip_event_record(e1, stream)
    torch.cuda.synchronize()
    kern_ms = [L.clhip_event_elapsed_ms(e0, e1) for e0, e1 in evs]
    for e0, e1 in evs + [(ev0, ev1)]:
        L.clhip_event_destroy(e0); L.clhip_event_destroy(e1)
    # every launch verified its chunks' sync words on the device; the byte-granular search agrees
    hip.smi_find_offsets(words, 4 * n, 4 * NATIVE_CHUNK_SAMPLES, 4 * NATIVE_CHUNK_SAMPLES, max(n_chunks, 1), offs, stream)
    assert int(bad.item()) == 0 and int(offs.abs().max().item()) == 0, "synthetic stream lost sync?"

    if rank == 0:
        # HBM traffic of the fused kernel from PMC counters: measured in separate rocprofv3 --pmc passes
        # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; tools/pmc_summ.py) and committed under profiles/
        full = a.log2_samples == 28 and not a.samples
        traffic, traffic_src = pmc_traffic("c2") if full else (None, None)
        if traffic is None and full:
            traffic, traffic_src = pmc_traffic("c")            # round 1 named its files c_*
        value = shard.job_throughput(n, a.steps, dt, world) / 1e6
        achieved = ALGO_BYTES_PER_SAMPLE * n / kern_avg_s / 1e9
        res = {
            "metric": "Msamples/s through unpack+FIR(64)+resample(3/2) pipe",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "settle_steps": a.settle,
            "config": {"workload": f"config 2: 1 ch/GPU, 4 MS/s stream replayed as one {(str(n) if a.samples else '2^' + str(a.log2_samples))}-sample "
                                   f"buffer per GPU ({n_chunks} native 512 KiB chunks), sync check + int13 unpack + "
                                   f"64-tap FIR + 3/2 polyphase resample, CF32 out",
                       "samples_per_gpu_per_step": n, "fir_taps": 64, "resample": "3/2", "arch": arch,
                       "parallelism": f"{world} independent stream(s), no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # the same algorithmic bytes priced on the WALL step (sync search + launch + gaps), the clock `value` uses
                         "frac_on_step_clock": round(ALGO_BYTES_PER_SAMPLE * n / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                         "traffic_source": traffic_src,
                         "kernel": "rx_pipe_fused_kernel<PipeCfg<64,3,2,8,MODE_IQ,16,256,FFA>, SMI_WORDS, S1G>",
                         "kernel_ms_avg": round(kern_avg_s * 1e3, 4),
                         "kernel_ms_avg_note": "HIP events on the launch stream around the K timed launches / K (dispatch gaps included)",
                         "kernel_ms_single_launch_avg": round(float(np.mean(kern_ms)), 4), "kernel_ms_min": round(float(np.min(kern_ms)), 4),
                         "algorithmic_bytes_per_sample": ALGO_BYTES_PER_SAMPLE,
                         "read_only_frac": round(4.0 * n / kern_avg_s / 1e9 / HBM_PEAK_GBS, 4),   # 4 B read per sample alone
                         "algorithmic_tflops": round(FLOP_PER_SAMPLE * n / kern_avg_s / 1e12, 2),
                         "algorithmic_frac_of_fp32_valu_peak": round(FLOP_PER_SAMPLE * n / kern_avg_s / 1e12 / VALU_PEAK_TFLOPS, 4)},
        }
        if world == 1 and not a.no_cpu:
            cb, cpu_out = cpu_baseline(taps, words, a.cpu_seconds)
            res["cpu_baseline"] = cb
            res["gpu_over_cpu"] = round(value / cb["value"], 1)
            # the CPU leg doubles as a parity check of the benchmarked GPU path (zero state, same bytes)
            pipe.reset()
            step()
            torch.cuda.synchronize()
            g = out[: cpu_out.shape[0]].cpu().numpy()
            res["max_abs_diff_vs_cpu_port"] = float(np.max(np.abs(g - cpu_out)))
        if world == 1 and not a.no_cpu and not a.no_pcie_extra and full:
            # The other truth of the same stages, next to the headline and never in its place (`value` is the HBM-resident figure):
            # host SMI bytes in -> host CF32 samples out through the drop-in boundary (32 Soapy devices, cl_group_readStream), against
            # this box's own PCIe ceiling -- `python bench.py --pcie` in a child process, after the timed region, its line condensed.
            try:
                o = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pcie", "--no-cpu", "--steps", "24", "--warmup", "6"],
                                   capture_output=True, text=True, timeout=240,
                                   env={k: v for k, v in os.environ.items() if not (k.startswith("ROCP") or k in ("LD_PRELOAD", "HSA_TOOLS_LIB"))})   # (a profiler around this run profiles this run)
                line = [ln for ln in o.stdout.splitlines() if ln.startswith("{")][-1]
                c = json.loads(line)
                res["pcie_inclusive"] = {"value": c["value"], "unit": c["unit"], "ms_per_step": c["ms_per_step"], "streams": c["config"]["streams_per_gpu"],
                                         "what": "the same stages behind the SoapySDR boundary: host SMI bytes in (pinned FIFOs) -> cl_group_readStream -> host CF32 samples in pageable "
                                                 "buffers; python bench.py --pcie prints the full line",
                                         "roofline": {k: c["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac")},
                                         "registered_client_buffers": c.get("registered_client_buffers")}
            except Exception as e:                             # (the headline line stands whatever happens here)
                res["pcie_inclusive"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
