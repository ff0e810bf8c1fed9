#!/usr/bin/env python3
"""PCIe-inclusive AGGREGATE rate of the Soapy boundary over many streams: host SMI bytes in (cl_smi_feed_bytes into every
device's pinned FIFO), host samples out into the clients' pageable numpy buffers through cl_group_readStream (one MTU =
131072 samples per stream and call), priced against the box's own concurrent H2D + D2H ceiling for the same message shape
(tools/microbench/pcie_duplex, measured in the same run).  Never bench.py's headline `value` (inputs resident in HBM).

    python tools/bench_group.py [--streams 32] [--calls 12] [--cases cf32_fir64_rs_3_2,cs16,cf32] [--sub 8] [--threads 4]

One JSON object: per case the group's rate (default route: pinned mirror + copy threads), the same with registered client
buffers (the copy engine writes them directly), the same N devices read ONE BY ONE through cl_readStream (round 3's way), and
`roofline: {bound: "pcie", ...}`."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MTU, NB = 131072, 524288

CASES = {
    "cs16": ("CS16", "int16", 2, None, 4.0),
    "cf32": ("CF32", "float32", 2, None, 8.0),
    "cf32_fir64_rs_3_2": ("CF32", "float32", 2, {"FIR": "64:1000000", "RESAMP": "3/2"}, 12.0),
    "cf32_fir64_fm_demod": ("CF32", "float32", 1, {"FIR": "64:100000", "DEMOD": "FM"}, 4.0),
    # BASELINE config 4's per-GPU share (256 streams over 8 GPUs = 32 per GPU) behind the boundary: 128-tap FIR + 5/4
    "cf32_fir128_rs_5_4": ("CF32", "float32", 2, {"FIR": "128:1200000", "RESAMP": "5/4"}, 10.0),
    # the reference's own low-pass selected on every member (setBandwidth 100 kHz: Butterworth-6, CaribouliteStream.cpp:282-301): such
    # members are read through their own devices inside the group call, their chains queued together
    "cs16_iir": ("CS16", "int16", 2, None, 4.0, 100e3),
}


def pcie_shape(in_mib, out_mib):
    exe = os.path.join(ROOT, "tools", "microbench", "pcie_duplex")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, str(in_mib), str(out_mib)], capture_output=True, text=True, timeout=120, check=True).stdout
        return json.loads(out[out.index("{"):])["shape"]
    except Exception as e:                                 # the rows stay, unpriced
        print("pcie_duplex failed:", e, file=sys.stderr)
        return None


# name: (format, dtype, input bytes per element, stream kwargs, words out per element)
TX_CASES = {"tx_cs16": ("CS16", "int16", 4, None, 1.0), "tx_cf32": ("CF32", "float32", 8, None, 1.0),
            "tx_cf32_fm_rs_2_3": ("CF32", "float32", 8, {"MOD": "FM:75000", "RESAMP": "2/3"}, 2.0 / 3.0),      # config 5's stages behind writeStream
            "tx_cf32_rs_2_3": ("CF32", "float32", 8, {"RESAMP": "2/3"}, 2.0 / 3.0)}


def run_tx_case(name, a):
    """cl_group_writeStream: the clients' samples (pageable) in, packed SMI words in the members' pinned TX FIFOs out (drained between
    the timed repetitions), against the same devices written one by one."""
    import numpy as np
    from cariboulite_amd import soapy as S
    fmt, dt, in_b, args, out_per = TX_CASES[name]
    out_words = lambda k: -(-k * MTU * 2 // 3) if out_per < 1.0 else k * MTU          # (2/3: ceil over the whole run of messages)
    n, K = a.streams, a.calls
    rng = np.random.default_rng(5)
    res = {}
    for mode in ("default", "one_by_one"):
        devs, sts = [], []
        for i in range(n):
            d = S.Device(dict(driver="Cariboulite", channel="S1G" if i % 2 == 0 else "HiF"))
            sts.append(d.setupStream(S.SOAPY_SDR_TX, fmt, args=args))
            d.activateStream(sts[-1])
            devs.append(d)
        bufs = [(rng.integers(-4096, 4096, (MTU, 2)).astype(np.int16) if fmt == "CS16" else (rng.random((MTU, 2), dtype=np.float32) - 0.5)) for _ in range(n)]
        kw = {"COPY_THREADS": str(a.threads), "INGEST_STREAMS": str(a.ingest)}
        if a.sub:
            kw["SUBBATCH"] = str(a.sub)
        if a.tx_copy_mb >= 0:
            kw["TX_COPY_MB"] = str(a.tx_copy_mb)
        grp = S.Group(devs, kw) if mode == "default" else None
        best = None
        for rep in range(a.reps + 1):
            t0 = time.perf_counter()
            for k in range(K):
                if grp is not None:
                    nd, rets = grp.writeStream(bufs, MTU)
                    assert nd == n, (k, nd, grp.lastError())
                else:
                    for i in range(n):
                        assert devs[i].writeStream(sts[i], [bufs[i]], MTU).ret == MTU
            if grp is not None:
                assert grp.flush() == 0                    # (write-behind: the last call's launches belong to the timed region)
            dt_s = time.perf_counter() - t0
            for d in devs:
                got = d.drainSmiBytes().size
                assert got == 4 * (out_words((rep + 1) * K) - out_words(rep * K)), (got, rep)
            if rep and (best is None or dt_s < best):
                best = dt_s
        res[mode] = {"msps_in": round(n * K * MTU / best / 1e6, 1), "ms_per_group_call": round(best / K * 1e3, 4), "us_per_stream_call": round(best / K / n * 1e6, 2)}
        if grp is not None:
            res[mode]["stats"] = grp.stats()
            grp.close()
        for d in devs:
            d.close()
    link_in_b = 4 if args and "MOD" in args else in_b      # (MOD=FM: only the I rail, the message, crosses PCIe -- the copy threads take it out of the clients' samples)
    in_mib, out_mib = n * MTU * link_in_b >> 20, int(n * MTU * 4 * out_per) >> 20
    sh = pcie_shape(in_mib, out_mib)
    if sh:
        ceiling = n * MTU / (sh["duplex_ms"] * 1e-3) / 1e6
        res["roofline"] = {"bound": "pcie", "unit": "Msamples/s", "peak": round(ceiling, 1),
                           "peak_note": f"{in_mib} MiB H2D and {out_mib} MiB D2H queued together on two streams, pinned memory, copy engine: {sh['duplex_ms']:.3f} ms",
                           **{f"frac_{m}": round(res[m]["msps_in"] / ceiling, 3) for m in ("default", "one_by_one")}}
    return res


def run_case(name, a):
    import numpy as np
    from cariboulite_amd import soapy as S, synth
    if name in TX_CASES:
        return run_tx_case(name, a)
    fmt, dt, width, args, out_b = CASES[name][:5]
    bw = CASES[name][5] if len(CASES[name]) > 5 else None
    n, K = a.streams, a.calls
    words = [synth.smi_stream_bytes(K * MTU, i % 2, stream=i)[0] for i in range(min(n, 4))]      # a few distinct streams, reused
    res = {}
    up, down = (int(x) for x in args["RESAMP"].split("/")) if args and "RESAMP" in args else (1, 1)
    n_out = MTU * up // down + 8 if up != down else MTU
    shape = (n_out, width) if width > 1 else (n_out,)

    def devices():
        devs, sts = [], []
        for i in range(n):
            d = S.Device(dict(driver="Cariboulite", channel="S1G" if i % 2 == 0 else "HiF"))
            sts.append(d.setupStream(S.SOAPY_SDR_RX, fmt, args=args))
            if bw:
                d.setBandwidth(S.SOAPY_SDR_RX, 0, bw)
            d.activateStream(sts[-1])
            devs.append(d)
        return devs, sts

    def feed(devs):
        for i, d in enumerate(devs):
            d.feedSmiBytes(words[i % len(words)])        # (device i and words i % 4 have the same channel type)

    for mode in a.modes.split(","):
        devs, sts = devices()
        bufs = [np.zeros(shape, dt) for _ in range(n)]
        grp = None
        if mode != "one_by_one":
            kw = {"COPY_THREADS": str(a.threads), "SINK": a.sink, "INGEST_STREAMS": str(a.ingest)}
            if a.sub:
                kw["SUBBATCH"] = str(a.sub)                # (0: the group's default for the lane's route)
            grp = S.Group(devs, kw)
            if mode == "registered":
                grp.registerBuffers(bufs)
        best, got_total, cpu_best = None, 0, None
        for rep in range(a.reps + 1):
            feed(devs)
            c0 = sum(os.times()[:2])                       # user + system seconds of the process, all threads: what the calls cost the HOST
            t0 = time.perf_counter()
            got = 0
            for k in range(K):
                if grp is not None:
                    nd, rets = grp.readStream(bufs, MTU)
                    assert nd == n, (mode, k, nd, grp.lastError())
                    got += sum(rets)
                else:
                    for i in range(n):
                        r = devs[i].readStream(sts[i], [bufs[i]], MTU).ret
                        assert r > 0
                        got += r
            dt_s = time.perf_counter() - t0
            if rep and (best is None or dt_s < best):
                best, got_total, cpu_best = dt_s, got, sum(os.times()[:2]) - c0
        res[mode] = {"msps_in": round(n * K * MTU / best / 1e6, 1), "ms_per_group_call": round(best / K * 1e3, 4),
                     "us_per_stream_call": round(best / K / n * 1e6, 2), "out_elems_per_call": got_total // K,
                     "host_cpu_s_per_wall_s": round(cpu_best / best, 2)}
        if grp is not None:
            res[mode]["stats"] = grp.stats()
            grp.close()
        for d in devs:
            d.close()
    in_mib, out_mib = n * NB >> 20, max(int(n * MTU * out_b) >> 20, 1)
    sh = pcie_shape(in_mib, out_mib)
    if sh:
        ceiling = n * MTU / (sh["duplex_ms"] * 1e-3) / 1e6
        res["roofline"] = {"bound": "pcie", "unit": "Msamples/s", "peak": round(ceiling, 1),
                           "peak_note": f"{in_mib} MiB H2D and {out_mib} MiB D2H queued together on two streams, pinned memory, copy engine: "
                                        f"{sh['duplex_ms']:.3f} ms (alone: H2D {sh['h2d_GBs']} GB/s, D2H {sh['d2h_GBs']} GB/s)",
                           **{f"frac_{m}": round(res[m]["msps_in"] / ceiling, 3) for m in res if m in ("default", "registered", "one_by_one")}}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=32)
    ap.add_argument("--calls", type=int, default=12)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sub", type=int, default=0)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--sink", default="mapped", choices=["mapped", "copy"])
    ap.add_argument("--ingest", type=int, default=2, help="ingest HIP streams (1 .. 8)")
    ap.add_argument("--tx-copy-mb", type=int, default=-1, help="TX cases: the group's TX_COPY_MB kwarg (0: one copy in per sub-batch; default: the group's, 8)")
    ap.add_argument("--cases", default="cf32_fir64_rs_3_2,cs16,cf32")
    ap.add_argument("--modes", default="default,registered,one_by_one")
    ap.add_argument("--only", default=None, help="(internal) one case in this process")
    a = ap.parse_args()
    if a.only:
        print(json.dumps({a.only: run_case(a.only, a)}))
        return 0
    # one fresh process per case: what a call costs must not depend on what the process registered or freed before
    out = {"streams": a.streams, "calls": a.calls, "subbatch": a.sub, "copy_threads": a.threads, "sink": a.sink, "ingest_streams": a.ingest,
           "runtime_env": {"GPU_PINNED_MIN_XFER_SIZE": os.environ.get("GPU_PINNED_MIN_XFER_SIZE", "(unset: the runtime's default)")}}
    for c in a.cases.split(","):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--only", c] + [x for x in sys.argv[1:]], capture_output=True, text=True)
        if r.returncode:
            out[c] = {"error": r.stderr[-1500:]}
            continue
        out.update(json.loads(r.stdout[r.stdout.index("{"):]))
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
