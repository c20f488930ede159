#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s end-to-end (detect -> downmix -> demod) on 10 MHz cf32.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (K1..K7) over one chunk of
synthetic 10 MHz cf32 IQ that is already resident in HBM (complex
AWGN + 10 DQPSK bursts per Msample) -- by default in the chunk's slot of the context's
history ring (irdm_ingest_ptr: where a producer that feeds in place writes it; every slot
is filled before the timed region), fed with one chunk of look-ahead at pipeline_depth 3 (four batch contexts);
the drain of the last chunks' per-burst chains is inside the timed region.  Every rank
processes its own stream (the path
partitions by stream / time-chunk, no data-path collective); per step the demodulated
frame records are gathered to rank 0 over RCCL.  `value` = samples all ranks processed
/ max-over-ranks wall time.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline       dominant KERNEL (the decimator, or K1 on scenes with few bursts: the longer device span per
                 launch of the two; the detector scan is a chain of ~10 short launches and is reported under
                 stage_ms only).  ms_per_launch = the kernel's OWN clock: its wavefronts stamp the device's
                 100 MHz clock when they start and end (irdm_kernel_clock), first wavefront in to last
                 wavefront out, averaged over the launches of the timed region -- i.e. in run, with the other
                 stages of neighbouring chunks beside it.  achieved = algorithmic bytes per launch /
                 ms_per_launch, vs the 8 TB/s HBM peak; kernel_ms_rocprof = the same kernel's average in the
                 newest committed rocprofv3 summary of this configuration (profiles/); stage_ms = HIP-event
                 brackets of the stages on their streams (they include what a launch waits in its queue);
                 stage_ms_alone / kernel_clock_ms_alone = the same from a few extra steps at pipeline_depth 0
                 (one kernel on the chip at a time).
  cpu_baseline   the CPU oracle (C port of the reference, the decimating FIR in the order of the reference's
                 AVX2 kernel like the product's default) in the reference's thread layout
                 (1 detector + 4 downmix + 1 demod thread, main.c:175) on a bounded prefix of the
                 same stream, N=1 / rank 0 only; cpu_baseline_1core: the same oracle on one core.
  parity_checked the records of one chunk through the HIP path compared field by field with the
                 oracle's for the same samples (burst indices and hard bits exact, soft 1e-4).
  detect_only    BASELINE config 2 on the same chunk (K1 + scan, burst records only).
  file_to_raw    the C99 binary (iridium-sniffer-hip -f) on the chunk written to a file: wall clock
                 of the whole process, file in the page cache.
  chunk_x2       this script run once more with chunks of twice the samples (same generator, density and options;
                 its own process), its first chunk checked against the oracle: what a host gets that hands the
                 library 128 Mi-sample chunks -- the detector scan's per-chunk launches and waits spread over
                 twice the samples.  Reported beside the headline, never as `value`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "iridium-sniffer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_scene(torch, device, fs, n, density_per_msample, seed):
    """Noise on the device, bursts synthesised on the host (float64) and added in place."""
    import siggen
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.randn((n, 2), generator=g, device=device, dtype=torch.float32)
    x.mul_(0.002)
    rng = np.random.default_rng(seed + 1000)
    fft = 1 << int(round(np.log2(fs / 1000.0)))
    first = 520 * fft
    nb = int(round(density_per_msample * n / 1e6))
    span = n - first - int(0.012 * fs)
    if span <= 0:
        raise SystemExit("--samples %d is too small: the detector primes on the first %d samples (512 frames of %d)"
                         % (n, first, fft))
    starts = np.sort(rng.integers(0, span, size=nb)) + first
    half_ch = int((fs / 2 - 60e3) // (1e6 / 24.0))
    lens = 0
    for s in starts:
        p = int(rng.integers(119, 180))
        quads = siggen.frame_quadrants(rng.integers(0, 4, size=p).tolist())
        ch = int(rng.integers(-half_ch, half_ch + 1)) or 1
        sig = siggen.make_burst(fs, quads, siggen.channel_freq(ch), rng.uniform(0, 2 * np.pi))
        e = min(n, int(s) + len(sig))
        t = torch.from_numpy(np.ascontiguousarray(sig[:e - int(s)]).view(np.float32).reshape(-1, 2)).to(device)
        x[int(s):e] += t
        lens += e - int(s)
    return x, nb


def cpu_reference_layout(orc, host, fs, fmt, workers=4):
    """The oracle's stage functions in the reference's thread layout (main.c:175, :667-694): one detector thread feeding
    32768-sample blocks, `workers` downmix threads, one demod thread, bounded queues in between.  ctypes releases the
    GIL inside the C calls, so the stages really run in parallel.  Returns wall seconds and counts."""
    import ctypes as C
    import queue
    import threading
    L = orc.lib()
    det = L.orc_detector_create(1622000000.0, int(fs), 0.0, 0)
    n_fft = L.orc_detector_fft_size(det)
    q_burst, q_frame = queue.Queue(2048), queue.Queue(512)
    counts = {"bursts": 0, "demods": 0}

    def cb(rec, samples, user):
        r = orc.BurstRec()
        C.memmove(C.byref(r), rec, C.sizeof(orc.BurstRec))
        buf = np.ctypeslib.as_array(samples, shape=(2 * r.num_samples,)).copy()
        counts["bursts"] += 1
        q_burst.put((r, buf))

    cbf = orc.BURST_CB(cb)

    def detector():
        per = 1 if fmt == 2 else 2
        flat = host.view(np.float32) if fmt == 2 else host
        step = 32768
        nn = len(host) if fmt == 2 else len(host) // 2
        for o in range(0, nn, step):
            cnt = min(step, nn - o)
            if fmt == 2:
                blk = flat[2 * o:2 * (o + cnt)]
                L.orc_detector_feed_cf32(det, orc.fptr(blk), cnt, cbf, None)
            else:
                blk = flat[per * o:per * (o + cnt)]
                i8 = (blk >> 8).astype(np.int8) if fmt == 1 else blk
                L.orc_detector_feed_i8(det, i8.ctypes.data_as(C.c_void_p), cnt, cbf, None)
        for _ in range(workers):
            q_burst.put(None)

    def worker():
        dm = L.orc_downmix_create()
        while True:
            it = q_burst.get()
            if it is None:
                break
            r, buf = it
            fr = orc.Frame()
            if L.orc_downmix_process(dm, C.byref(r), orc.fptr(buf), 1622000000.0, int(fs), n_fft, 1700000000 * 10**9, C.byref(fr)) > 0:
                q_frame.put(fr)
        L.orc_downmix_destroy(dm)
        q_frame.put(None)

    def demod():
        done = 0
        buf = C.create_string_buffer(4096)
        t0 = C.c_uint64(0)
        while done < workers:
            fr = q_frame.get()
            if fr is None:
                done += 1
                continue
            d = orc.Demod()
            if L.orc_qpsk_demod(C.byref(fr), 1, C.byref(d)) > 0:
                L.orc_format_raw(C.byref(d), b"bench", C.byref(t0), buf, 4096)
                counts["demods"] += 1

    ths = [threading.Thread(target=detector)] + [threading.Thread(target=worker) for _ in range(workers)] + \
          [threading.Thread(target=demod)]
    t = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t
    L.orc_detector_destroy(det)
    return dict(seconds=dt, bursts=counts["bursts"], demods=counts["demods"])


def cpu_model():
    """the host CPU's model string (/proc/cpuinfo), for the cpu_baseline entry"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def record_msg_bytes(irdm, cap):
    """bytes of one rank's record message: a count word + cap compact records (pack_records)"""
    return 8 + cap * (irdm.Demod.bits.offset + irdm.Demod.bits.size // 8)


def unpack_demods(irdm, packed):
    """[n, 176] irdm_demod_packed_t bytes -> [n, 4544] irdm_demod_t bytes with the LLRs zero (for the parity check)"""
    head, nbits = irdm.Demod.bits.offset, irdm.Demod.bits.size
    full = np.zeros((len(packed), C.sizeof(irdm.Demod)), np.uint8)
    if len(packed):
        full[:, :head] = packed[:, :head]
        full[:, head:head + nbits] = np.unpackbits(packed[:, head:head + nbits // 8], axis=1)
    return full


def pack_records(irdm, demods, hb, cap):
    """What frame_output_print needs of the demodulated frames of one step (frame_output.c:168-197) into the message
    buffer `hb` (numpy uint8 view of pinned memory): a count word, then per frame the record's head (id, timestamp,
    frequency, magnitude, noise, confidence, level, symbol counts) and the hard bits packed 8 per byte = 176 bytes.
    More frames than the message holds is an error: nothing is ever dropped silently."""
    head, nbits = irdm.Demod.bits.offset, irdm.Demod.bits.size
    recp = head + nbits // 8
    k = len(demods)
    if k > cap:
        raise SystemExit("bench: %d demodulated frames in one step exceed the gather message (%d)" % (k, cap))
    hb[:8] = np.array([k], dtype=np.int64).view(np.uint8)
    if k and demods.shape[1] == recp:              # already irdm_demod_packed_t (option packed_records)
        hb[8:8 + k * recp] = demods.reshape(-1)
    elif k:
        rec = hb[8:8 + k * recp].reshape(k, recp)
        rec[:, :head] = demods[:, :head]
        rec[:, head:] = np.packbits(demods[:, head:head + nbits], axis=1)
    return k


def bench_time_shard(args, torch, dist, irdm, rank, world, local, device, cdev, backend):
    """BASELINE config 4: ONE 12 MHz cf32 stream, 16384-point detect, consecutive time-chunks on consecutive ranks.
    Rank 0 holds one period of the stream (world chunks) in HBM and scatters [overlap | chunk] slices every super-step;
    the detector state travels rank to rank (sharding.TimeShard); demodulated-frame records are gathered to rank 0.
    value = world * chunk * steps / max-over-ranks time."""
    import sharding
    fs = 12_000_000 if args.sample_rate == 10_000_000 else args.sample_rate
    nfft = 1 << int(round(np.log2(fs / 1000.0)))
    chunk = args.samples // 32768 * 32768
    ov = (sharding.required_overlap(fs, nfft) + 15) // 16 * 16
    if ov >= chunk:
        raise SystemExit("--samples %d is smaller than the chunk overlap %d" % (chunk, ov))
    bps = 8
    nccl = backend == "nccl"
    parts = None
    nb = 0
    if rank == 0:
        x, nb = build_scene(torch, device, fs, world * chunk, args.density, seed=4)
        flat = x.view(torch.uint8).reshape(-1)
        parts = []
        for k in range(world):
            lo = (k * chunk - ov) * bps
            hi = (k + 1) * chunk * bps
            piece = torch.cat([flat[lo:], flat[:hi]]) if lo < 0 else flat[lo:hi].clone()
            parts.append(piece if nccl else piece.cpu())
        del x, flat
    # two slices' worth of landing buffers: the slice of super-step s + 1 travels while super-step s computes (a slice is
    # (overlap + chunk) * 8 bytes = 739 MB at 12 MHz: 10-12 ms on one xGMI link, more than the N scans of a super-step)
    n_buf = 2 if world > 1 else 1
    bufs = [torch.empty((ov + chunk) * bps, dtype=torch.uint8, device=device) for _ in range(n_buf)]
    stages = None if nccl else [torch.empty((ov + chunk) * bps, dtype=torch.uint8) for _ in range(n_buf)]
    # (several ranks: a rank works on one chunk per super-step, its per-burst chain overlaps the other ranks' scans --
    # pipeline_depth 1; one rank: the ordinary chained feed, chunk k + 1 begun before chunk k ends)
    pipe = irdm.Pipeline(fs, fmt=irdm.FMT_CF32, max_chunk_samples=chunk, max_bursts_per_chunk=8192, device=local,
                         pipeline_depth=min(args.depth, 1) if world > 1 else args.depth)
    for kv in args.opt:
        key, val = kv.split("=")
        pipe.set_option(key, int(val))
    ts = sharding.TimeShard(dist, pipe, torch, device, chunk, bps, ov) if world > 1 else None
    ingest = world == 1 and args.depth >= 1
    if ingest:
        # one rank: the producer writes every chunk where the context keeps it (irdm_ingest_ptr), as in the stream mode --
        # every chunk-sized slot of the history ring holds the chunk before the timed region
        ring_ptr, ring_len = pipe.ring()
        if ring_len % chunk == 0:
            for k in range(ring_len // chunk):
                rc = irdm.lib().irdm_device_copy(C.c_void_p(ring_ptr + k * chunk * bps), C.c_void_p(parts[0].data_ptr() + ov * bps), chunk * bps)
                assert rc == 0, rc
            torch.cuda.synchronize()
        else:
            ingest = False
    cap = 8192                           # = max_bursts_per_chunk above: more frames in one chunk are an error, not truncated
    msg = record_msg_bytes(irdm, cap)
    # (the record gather is double-buffered and asynchronous like the stream mode's: a step's message travels while the next
    # step computes)
    ghost = [torch.zeros((msg,), dtype=torch.uint8).pin_memory() for _ in range(2)] if world > 1 else None
    gbuf = [torch.zeros((msg,), dtype=torch.uint8, device=cdev) for _ in range(2)] if world > 1 else None
    glist = [[torch.zeros((msg,), dtype=torch.uint8, device=cdev) for _ in range(world)] for _ in range(2)] if (world > 1 and rank == 0) else [None, None]
    counts = torch.zeros((4,), dtype=torch.int64, device=cdev)       # bursts, frames produced, frames sent, frames gathered
    step_no = [0]
    scatter = {"work": None}
    gather = {"work": None, "slot": 0, "record": False}
    parts_t = {"scatter_wait": 0.0, "collect": 0.0}      # host seconds per part of a super-step beside TimeShard.t (rank 0's are printed)

    def issue_scatter(i):
        return dist.scatter(bufs[i] if nccl else stages[i], parts if rank == 0 else None, src=0, async_op=True)

    def finish_gather():
        if gather["work"] is not None:
            gather["work"].wait()
            if gather["record"] and rank == 0:
                counts[3] += torch.stack([l[:8] for l in glist[gather["slot"]]]).view(torch.int64).sum()
            gather["work"] = None

    def collect(record):
        """the records that have come out of the pipeline so far: counted, packed, sent towards rank 0"""
        nbst = len(pipe.poll_bursts_raw())
        pipe.drop_frames()
        demods = pipe.poll_demods_raw()
        k = 0
        if world > 1:
            finish_gather()                          # (the message before the last one: its buffers are free again)
            slot = gather["slot"] ^ 1
            k = pack_records(irdm, demods, ghost[slot].numpy(), cap)
            gbuf[slot].copy_(ghost[slot])
            if cdev.type == "cuda":
                torch.cuda.current_stream().synchronize()
            gather.update(work=dist.gather(gbuf[slot], glist[slot], dst=0, async_op=True), slot=slot, record=record)
        if record:
            counts[0] += nbst
            counts[1] += len(demods)
            counts[2] += k

    def super_step(record):
        if world > 1:
            i = step_no[0] % 2
            tq = time.perf_counter()
            if scatter["work"] is None:
                scatter["work"] = issue_scatter(i)              # (the very first slice: nothing to hide it behind)
            scatter["work"].wait()
            if not nccl:
                bufs[i].copy_(stages[i])
            # the stream the slice arrived on, not the device: the previous super-step's per-burst chain is still running
            torch.cuda.current_stream().synchronize()
            scatter["work"] = issue_scatter(1 - i)              # the next super-step's slices, under this one's compute
            parts_t["scatter_wait"] += time.perf_counter() - tq
            ts.step(bufs[i], first_of_stream=True)
        elif not args.depth:
            pipe.feed_device(parts[0].data_ptr() + ov * bps, chunk, None)
        else:
            pipe.feed_begin(pipe.ingest_ptr(chunk) if ingest else parts[0].data_ptr() + ov * bps, chunk, None)
            if step_no[0] > 0:
                pipe.feed_end()
        step_no[0] += 1
        tq = time.perf_counter()
        collect(record)
        parts_t["collect"] += time.perf_counter() - tq

    for _ in range(args.warmup):
        super_step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if ts is not None:
        ts.t = dict.fromkeys(ts.t, 0.0)
        ts.steps_timed = 0
    parts_t = dict.fromkeys(parts_t, 0.0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        super_step(True)
    if world == 1 and args.depth:
        # drain: the last chunk's scan and the chains in flight belong to the timed work
        pipe.feed_end()
        step_no[0] = 0
        pipe.flush()
        counts[0] += len(pipe.poll_bursts_raw())
        pipe.drop_frames()
        counts[1] += len(pipe.poll_demods_raw())
    if world > 1:
        # drain: the last super-step's chains, their records, the gathers in flight and the slice that was sent ahead for a
        # super-step that does not come all belong to the timed work
        ts.drain()
        collect(True)
        finish_gather()
        scatter["work"].wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    K = max(args.steps, 1)
    t = pipe.timings()
    if rank == 0:
        state_mb = pipe.state_bytes() / 1e6
        out = {
            "metric": "IQ Msamples/s end-to-end (detect->demod), %d MHz cf32" % (fs // 1_000_000),
            "value": round(world * chunk * K / dt / 1e6, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg4: ONE %d MHz cf32 stream, %d-pt detect, full pipeline; a step = %d consecutive time-chunks of "
                                   "%d samples, one per GPU, scattered from rank 0 with a %d-sample overlap, detector state "
                                   "(%.1f MB) handed rank to rank, %.0f bursts/Msample"
                                   % (fs // 1_000_000, nfft, world, chunk, ov, state_mb, args.density),
                       "samples_per_step_per_gpu": chunk, "parallelism": "time-chunks x%d" % world,
                       "job_bursts_per_step": int(counts[0].item()) / K, "raw_frames_per_step": int(counts[1].item()) / K,
                       "records": ({"produced": int(counts[1].item()), "sent": int(counts[2].item()),
                                    "gathered_on_rank0": int(counts[3].item())} if world > 1 else None),
                       "pipeline_depth": args.depth, "backend": backend,
                       # host milliseconds per super-step on rank 0, part by part: waiting for this step's slice (sent under the
                       # previous step), overlap + K1 enqueue, the previous rank's head, scan enqueue, the history, the scan
                       # itself, the sends, enqueueing the chain (irdm_advance), collecting / gathering records
                       "parts_ms_rank0": ({**{k: round(v / K * 1e3, 3) for k, v in parts_t.items()},
                                           **{k: round(v / max(ts.steps_timed, 1) * 1e3, 3) for k, v in ts.t.items()}}
                                          if ts is not None else None),
                       "ring_waits": pipe.stat("ring_waits"),
                       "scan": {k: pipe.stat(k) for k in ("scan_fast_chunks", "scan_fallbacks", "band_chunks", "band_rounds", "band_aborts")},
                       # rotator checkpoint rows: prebuilt runs per bin (0: on demand), builds / checkpoints chains had to make
                       "rot": {k: pipe.stat(k) for k in ("rot_prebuilt_runs", "rot_builds", "rot_ckpts", "rot_grows", "scratch_grows", "tiles_grows")},
                       "host_us": {k: pipe.stat("host_us_%d" % i) for i, k in enumerate(
                           ("k1_ring_enqueue", "settle", "chain_enqueue", "scan_enqueue", "wait_older_chain", "final_sync",
                            "settle_wait_scan", "settle_counters", "settle_records", "build_records"))}},
            "roofline": {"bound": "hbm", "kernel": "fir_decimate_kernel_f", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None, "traffic": None, "stage_ms_rank0_last_step": {k: round(v, 4) for k, v in t.items()}},
            "cpu_baseline": None,
        }
        if world > 1 and not (out["config"]["records"]["produced"] == out["config"]["records"]["sent"] ==
                              out["config"]["records"]["gathered_on_rank0"]):
            raise SystemExit("bench: record gather lost frames: %r" % (out["config"]["records"],))
        print(json.dumps(out), flush=True)
    pipe.close()


def rocprof_avg_ms(kernel, fs, density):
    """average duration of `kernel` in the newest committed rocprofv3 --kernel-trace --stats summary of THIS configuration
    (profiles/r<round>_kernel_stats.csv: 10 MHz, 10 bursts per Msample; ..._cfg5_12mhz_d40.csv: 12 MHz, 40), or None"""
    import glob
    import re
    if fs == 10_000_000 and density == 10:
        pat = r"r(\d+)_kernel_stats\.csv$"
    elif fs == 12_000_000 and density == 40:
        pat = r"r(\d+)_kernel_stats_cfg5_12mhz_d40\.csv$"
    else:
        return None
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats*.csv")):
        m = re.search(pat, os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return None
    try:
        for line in open(best[1]).read().splitlines()[1:]:
            name, calls, total_ms, avg_us = line.rsplit(",", 4)[:4]
            if kernel in name:
                return {"avg_ms": round(float(avg_us) / 1e3, 4), "calls": int(calls), "file": os.path.basename(best[1])}
    except Exception:
        return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--samples", type=int, default=64 * 1024 * 1024, help="samples per chunk per GPU")
    ap.add_argument("--density", type=float, default=10.0, help="bursts per Msample")
    ap.add_argument("--sample-rate", type=int, default=10_000_000)
    ap.add_argument("--format", choices=("cf32", "ci16", "ci8"), default="cf32",
                    help="device sample format of the chunk (the headline metric is cf32; ci16 / ci8 = the reference's "
                         "integer file formats, quantised from the same scene)")
    ap.add_argument("--depth", type=int, default=3,
                    help="pipeline_depth: 1 .. 5 (default 3: four batch contexts) = the detector scan of chunk k stays in flight "
                         "while the per-burst chains of the previous depth + 1 chunks and chunk k+1's FFT run (results later, "
                         "identical); 0 = every feed returns its own chunk's results.  (A chain is ~3 ms of dependent launches "
                         "in run: with three contexts the period was that latency / 3, profiles/r5_depth_sweep.json)")
    ap.add_argument("--cpu-samples", type=int, default=64 * 1024 * 1024,
                    help="prefix of the stream the CPU oracle is timed on (0 = skip)")
    ap.add_argument("--cpu-passes", type=int, default=3, help="one-core oracle passes over that prefix (~3 s each)")
    ap.add_argument("--shard", choices=("streams", "time"), default="streams",
                    help="streams (default): one independent stream per GPU (BASELINE config 5 / the headline metric at N=1); "
                         "time: ONE 12 MHz stream, consecutive time-chunks on consecutive ranks with the detector state handed "
                         "rank to rank over RCCL (BASELINE config 4)")
    ap.add_argument("--alone-steps", type=int, default=4,
                    help="extra steps at pipeline_depth 0 (stage times with one kernel on the chip at a time, and the records "
                         "for the parity check; 0 = skip)")
    ap.add_argument("--detect-steps", type=int, default=10, help="extra detect-only steps (BASELINE config 2; 0 = skip)")
    ap.add_argument("--file-run", type=int, default=2,
                    help="1: also time the C99 binary on the chunk written to a file; 2: and on a 600 Msample recording")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT",
                    help="irdm_set_option before the run (kernel-variant A/B: fir_generic=1, fft_radix2=1, scan_mode=1)")
    ap.add_argument("--packed", type=int, default=1,
                    help="1: the timed context queues compact frame records (irdm_demod_packed_t: no LLRs, bits 8 per byte)")
    ap.add_argument("--ingest", type=int, default=1,
                    help="pipeline_depth >= 1: 1 = the chunk lives in its slot of the history ring (irdm_ingest_ptr; the ring is "
                         "filled with the synthetic chunk before the timed region), 0 = fed from a separate buffer and copied")
    ap.add_argument("--lookahead", type=int, default=1,
                    help="pipeline_depth >= 1: 1 = irdm_feed_begin(k+1) before irdm_feed_end(k); 0 = irdm_feed_device")
    ap.add_argument("--host-steps", type=int, default=6,
                    help="extra, separately timed steps fed from pinned HOST memory (PCIe-inclusive rate; 0 = skip)")
    ap.add_argument("--cpu-layout", type=int, default=1,
                    help="0: skip the thread-layout CPU runs (cpu_baseline = the one-core pass): what the chunk_x2 sub-run uses")
    ap.add_argument("--parity-chunks", type=int, default=2, choices=(1, 2),
                    help="parity_checked over the stream's first two chunks (default) or the first one only")
    ap.add_argument("--big-chunk-steps", type=int, default=20,
                    help="N = 1, cf32: a second, separately timed run of this script with chunks of twice the samples (same scene "
                         "generator and density, same options), N steps, parity-checked on its first chunk -> `chunk_x2` on the "
                         "JSON line (never `value`); 0 = skip")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import irdm

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # IRDM_BENCH_BACKEND=gloo + IRDM_BENCH_SHARE_GPU=1: functional check of the multi-rank code path on a one-GPU box
    # (all ranks on device 0, collectives through host tensors); never used for reported numbers
    backend = os.environ.get("IRDM_BENCH_BACKEND", "nccl")
    if os.environ.get("IRDM_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    cdev = device if backend == "nccl" else torch.device("cpu")      # where collective buffers live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    fs = args.sample_rate
    n = args.samples // 32768 * 32768
    irdm.build()
    if args.shard == "time":
        bench_time_shard(args, torch, dist, irdm, rank, world, local, device, cdev, backend)
        if world > 1:
            dist.destroy_process_group()
        return
    x, nb = build_scene(torch, device, fs, n, args.density, seed=2 + rank)
    torch.cuda.synchronize()

    fmt = {"cf32": irdm.FMT_CF32, "ci16": irdm.FMT_CI16, "ci8": irdm.FMT_CI8}[args.format]
    bps = {"cf32": 8, "ci16": 4, "ci8": 2}[args.format]
    if args.format == "ci16":       # siggen.to_ci16: x * 131072, rounded, clipped
        x = torch.clamp(torch.round(x * 131072.0), -32768, 32767).to(torch.int16)
    elif args.format == "ci8":      # siggen.to_ci8 with headroom for the burst peaks
        x = torch.clamp(torch.round(x * 512.0 * 4), -128, 127).to(torch.int8)
    torch.cuda.synchronize()
    pipe = irdm.Pipeline(fs, fmt=fmt, max_chunk_samples=n, max_bursts_per_chunk=8192,      # (= the gather message's cap below)
                         device=local, pipeline_depth=args.depth)
    pipe.L.irdm_feed_device.restype = C.c_int
    for kv in args.opt:
        key, val = kv.split("=")
        pipe.set_option(key, int(val))
    packed = bool(args.packed)
    if packed:
        pipe.set_option("packed_records", 1)
    pipe.set_option("kernel_clock", 1)      # the chip-filling kernels' own device spans (irdm_kernel_clock): roofline below
    poll_demods = pipe.poll_demods_packed_raw if packed else pipe.poll_demods_raw
    stream = None        # the chunk is complete in HBM before the timed region: nothing to order against
    # record gather to rank 0 (RCCL over xGMI): what frame_output_print needs of a demodulated frame (frame_output.c:
    # 168-197) -- the record's head (id, timestamp, frequency, magnitude, noise, confidence, level, symbol counts) and
    # the hard bits packed 8 per byte: 176 bytes per frame, a count word in front.  Fixed-size messages sized for the
    # context's max_bursts_per_chunk (a step with more frames than that is an error, never a silent truncation),
    # double-buffered and asynchronous so the collective of step i overlaps the detector scan of step i+1.
    cap = 8192                                                         # = max_bursts_per_chunk above
    MSG = record_msg_bytes(irdm, cap)
    gather_host = [torch.zeros((MSG,), dtype=torch.uint8).pin_memory() for _ in range(2)] if world > 1 else None
    gather_bufs = [torch.zeros((MSG,), dtype=torch.uint8, device=cdev) for _ in range(2)] if world > 1 else None
    gather_lists = ([[torch.zeros((MSG,), dtype=torch.uint8, device=cdev) for _ in range(world)] for _ in range(2)]
                    if (world > 1 and rank == 0) else [None, None])
    gather_work = [None, None]
    gathered = torch.zeros((1,), dtype=torch.int64, device=cdev)        # rank 0: records received, from the count words

    def gather_wait(slot):
        if gather_work[slot] is None:
            return
        gather_work[slot].wait()
        gather_work[slot] = None
        if rank == 0:
            gathered.add_(torch.stack([l[:8] for l in gather_lists[slot]]).view(torch.int64).sum())

    def gather_send(slot, demods):
        k = pack_records(irdm, demods, gather_host[slot].numpy(), cap)
        gather_bufs[slot].copy_(gather_host[slot], non_blocking=True)
        gather_work[slot] = dist.gather(gather_bufs[slot], gather_lists[slot], dst=0, async_op=True)
        return k
    step_no = [0]
    counts = torch.zeros((3,), dtype=torch.int64, device=cdev)

    stage = {k: 0.0 for k in ("fft_mag", "scan", "fir", "post", "demod", "total")}
    totals = dict(bursts=0, demods=0, burst_samples=0)

    host = {"feed_call": 0.0, "poll": 0.0}

    ingest = bool(args.depth and args.ingest)
    look = bool(args.depth and args.lookahead)
    if ingest:
        # inputs resident in HBM before the timed region: every chunk-sized slot of the history ring holds the chunk
        # (what a producer that writes in place -- an H2D copy, a conversion kernel -- leaves behind)
        ring_ptr, ring_len = pipe.ring()
        assert ring_len % n == 0, (ring_len, n)
        torch.cuda.synchronize()
        for k in range(ring_len // n):
            rc = irdm.lib().irdm_device_copy(C.c_void_p(ring_ptr + k * n * bps), C.c_void_p(x.data_ptr()), n * bps)
            assert rc == 0, rc
    pending = [0]

    def feed_one():
        ptr = pipe.ingest_ptr(n) if ingest else x.data_ptr()
        assert ptr
        if not look:
            return pipe.feed_device(ptr, n, stream)
        pipe.feed_begin(ptr, n, stream)
        pending[0] += 1
        if pending[0] > 1:          # (one chunk begun ahead of the one that is ended)
            pending[0] -= 1
            return pipe.feed_end()
        return 0

    def drain_feed():
        nb = 0
        while pending[0]:
            pending[0] -= 1
            nb += pipe.feed_end()
        return nb

    # the records of the stream's FIRST TWO chunks as the timed context produces them (pipeline_depth 3, fed in place with
    # look-ahead, the other chunks' stages running beside them): what parity_checked compares with the oracle run over a
    # two-chunk prefix of the stream.  Everything the context emits during the warm-up is kept (records leave in stream order).
    head = {"bursts": [], "demods": [], "open": True}

    def step(record):
        ta = time.perf_counter()
        nb_step = feed_one()
        tb = time.perf_counter()
        bursts = pipe.poll_bursts_raw()          # [n, 72] bytes
        pipe.drop_frames()
        demods = poll_demods()          # [n, 176] bytes (packed records; 4544 with the full ones): everything frame_output_print needs
        tc = time.perf_counter()
        if head["open"] and not record:
            head["bursts"].append(bursts.copy()); head["demods"].append(demods.copy())
        if record:
            host["feed_call"] += (tb - ta) * 1e3
            host["poll"] += (tc - tb) * 1e3
        if world > 1:
            slot = step_no[0] & 1
            step_no[0] += 1
            gather_wait(slot)
            k = gather_send(slot, demods)
            if record:
                counts[0] += nb_step
                counts[1] += len(demods)
                counts[2] += k
        if record:
            t = pipe.timings()
            for kk in stage:
                stage[kk] += t[kk]
            totals["bursts"] += len(bursts)
            totals["demods"] += len(demods)
            ns_off = irdm.Burst.num_samples.offset
            totals["burst_samples"] += int(bursts[:, ns_off:ns_off + 8].copy().view(np.uint64).sum()) if len(bursts) else 0

    for _ in range(args.warmup):
        step(False)
    if args.depth:
        drain_feed()
        pipe.flush()
        wb = pipe.poll_bursts_raw(); pipe.drop_frames(); wd = poll_demods()
        # (the warm-up chunks' records that were still in flight leave together here, in chunk order)
        head["bursts"].append(wb.copy()); head["demods"].append(wd.copy())
    head["open"] = False
    head["chunks"] = args.warmup
    if world > 1:
        for slot in (0, 1):
            gather_wait(slot)
        gathered.zero_()
        dist.barrier()
    torch.cuda.synchronize()
    pipe.kernel_clock(0, reset=True)            # (waits for the device: outside the timed region)
    pipe.kernel_clock(1, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    if args.depth:
        # drain: the last chunk's detector scan and per-burst stages belong to the timed work
        nb_tail = drain_feed()
        pipe.flush()
        tb_ = pipe.poll_bursts_raw()
        pipe.drop_frames()
        tail = poll_demods()
        totals["demods"] += len(tail)
        totals["bursts"] += len(tb_)
        if world > 1:
            counts[0] += len(tb_)
            counts[1] += len(tail)
            slot = step_no[0] & 1
            step_no[0] += 1
            gather_wait(slot)
            counts[2] += gather_send(slot, tail)
        ns_off = irdm.Burst.num_samples.offset
        totals["burst_samples"] += int(tb_[:, ns_off:ns_off + 8].copy().view(np.uint64).sum()) if len(tb_) else 0
    if world > 1:
        for slot in (0, 1):
            gather_wait(slot)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)      # whole-job burst / record counts
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    K = max(args.steps, 1)
    # what a SCALE record can be checked with: the size of the group the collectives really ran over (RCCL when the backend
    # is nccl) and every rank's own rate over its own clock (the job's rate divides by the slowest rank's time)
    group_world = dist.get_world_size() if world > 1 else 1
    rank_rates = [n * K / dt / 1e6]
    if world > 1:
        mine = torch.tensor([dt], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_rates = [n * K / float(t.item()) / 1e6 for t in every]
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    ms = {k: v / K for k, v in stage.items()}
    value = world * n * K / dt / 1e6

    # ---- roofline of the dominant kernel (DESIGN.md "Algorithmic bytes") ----
    decim = int(round(fs / 250000.0))
    lb = totals["burst_samples"] / K
    alg_bytes = {
        "fft_mag": float(bps) * n,                # one read of every sample (B_det's b_in: 8 / 4 / 2 bytes)
        "scan": 8.0 * n,                          # history row read + write per bin-frame (B_det = 16 B/sample with K1)
        "fir": float(bps) * lb + 8.0 * lb / decim,   # burst-window re-read + decimated (cf32) write
    }
    opts = dict((kv.split("=", 1)[0], int(kv.split("=", 1)[1])) for kv in args.opt)
    # (the kernels the library picks for this configuration: DESIGN.md section 5; options fir_order / k1_kernel)
    fir_name = "fir_decimate_kernel"                   # (the any-M kernel: 2 / 4 MHz)
    if decim in (40, 48) and not opts.get("fir_generic", 0):
        fir_name = "fir_decimate_kernel_f" if opts.get("fir_order", 1) else "fir_decimate_kernel_r"
    k1_name = "fft_mag_p32_kernel" if pipe.fft_size >= 8192 else "fft_mag_r16_kernel" if pipe.fft_size == 4096 else "fft_mag_kernel"
    kernels = {"fft_mag": k1_name, "scan": "band_* (scan_band.hip passes)", "fir": fir_name}
    # the dominant KERNEL: the scan is a chain of ~25 short launches of six kernels (its stage time is their sum plus
    # what they wait for each other), so it is reported in stage_ms / stage_GBps but not as "the" kernel
    # (the decimator does 57 % of the step's algorithmic bytes and all of its arithmetic; K1 only where bursts are so few
    # that the decimator's launch is less than half of K1's)
    # ms_per_launch = the kernel's own span on the device, first wavefront in to last wavefront out (s_memrealtime stamps
    # inside the kernel, irdm_kernel_clock), averaged over the launches of the timed region -- not the HIP-event bracket
    # (stage_ms), which also holds the strip-geometry kernel and whatever the launch waited for in the queue.  The
    # dominant kernel is the one of the two chip-filling kernels with the longer span per launch.
    kclk = {}
    for which, key in ((0, "fir"), (1, "fft_mag")):
        sm, nl, _ = pipe.kernel_clock(which)
        kclk[key] = {"ms": sm / nl if nl else 0.0, "launches": nl}
    has_clock = kernels["fir"] != "fir_decimate_kernel" and kclk["fir"]["launches"] > 0
    if has_clock:
        dom = "fir" if kclk["fir"]["ms"] >= kclk["fft_mag"]["ms"] else "fft_mag"
    else:
        dom = "fir" if ms["fir"] >= 0.5 * ms["fft_mag"] else "fft_mag"
    dom_ms = kclk[dom]["ms"] if kclk[dom]["launches"] > 0 else ms[dom]
    ach = alg_bytes[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM traffic per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in their own rocprofv3 runs and
    # corrected as MI355X_MICROARCH.md prescribes; profiles/summarize.py -> profiles/<round>_pmc.json)
    traffic = None
    try:
        import glob
        import re
        files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")) if re.search(r"r\d+_pmc\.json$", os.path.basename(f))]
        files.sort(key=lambda f: int(re.search(r"r(\d+)_pmc", os.path.basename(f)).group(1)))
        if fs == 12_000_000 and args.density == 40:
            files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_cfg5.json"))]
            files.sort(key=lambda f: int(re.search(r"r(\d+)_pmc", os.path.basename(f)).group(1)))
        elif not (fs == 10_000_000 and args.density == 10):
            files = []
        if files:
            pmc = json.load(open(files[-1]))
            for kname, d in pmc.items():
                if kernels[dom] in kname:
                    traffic = round(d["traffic_bytes"])
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": kernels[dom], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes": round(alg_bytes[dom]),
                "ms_per_launch": round(dom_ms, 4),
                "ms_per_launch_source": ("device clock inside the kernel: first wavefront in .. last wavefront out, mean of %d launches"
                                         % kclk[dom]["launches"]) if kclk[dom]["launches"] > 0 else "HIP-event bracket (no kernel clock in this kernel)",
                "dominant_rule": "longer device span per launch of {decimator, K1}",
                "kernel_clock_ms": {k: round(v["ms"], 4) for k, v in kclk.items()},
                "kernel_ms_rocprof": rocprof_avg_ms(kernels[dom], fs, args.density),
                "stage_ms": {k: round(v, 4) for k, v in ms.items()},
                "stage_ms_alone": None,
                "host_ms": {k: round(v / K, 3) for k, v in host.items()},
                "stage_GBps": {k: round(alg_bytes[k] / (ms[k] * 1e-3) / 1e9, 2) for k in alg_bytes if ms[k] > 0},
                # SURVEY 8(d) B_full = B_det * N_samples + sum_bursts 8 * L_b (+ 8 * L_b / M): all algorithmic bytes of one
                # step over the whole step time -- the path's HBM fraction as a whole (the sequential scan bounds it)
                "pipeline": {"algorithmic_bytes": round(sum(alg_bytes.values())),
                             "achieved_GBps": round(sum(alg_bytes.values()) / (dt / K) / 1e9, 2),
                             "frac": round(sum(alg_bytes.values()) / (dt / K) / 1e9 / HBM_PEAK_GBS, 5)}}

    # ---- PCIe-inclusive rates: the same scene fed from pinned HOST memory (never `value`), in the three sample formats the
    #      reference's files come in -- cf32 (8 B/sample over PCIe), ci16 (4 B) and ci8 (2 B: SURVEY 8f.1, main.c:223-284,
    #      simd_generic.c:147-153: the conversion is part of K1's and the decimator's load stage) ----
    pcie = None
    if rank == 0 and world == 1 and args.host_steps > 0:
        xf = x if args.format == "cf32" else None

        def quantised(name):
            if name == args.format:
                return x
            if xf is None:
                return None
            if name == "ci16":
                return torch.clamp(torch.round(xf * 131072.0), -32768, 32767).to(torch.int16)
            return torch.clamp(torch.round(xf * 512.0 * 4), -128, 127).to(torch.int8)

        pcie = {}
        for name, code, nb_ in (("cf32", irdm.FMT_CF32, 8), ("ci16", irdm.FMT_CI16, 4), ("ci8", irdm.FMT_CI8, 2)):
            xq = quantised(name)
            if xq is None:
                continue
            if name == args.format:
                ph, own = pipe, False
            else:
                ph = irdm.Pipeline(fs, fmt=code, max_chunk_samples=n, max_bursts_per_chunk=8192, device=local, pipeline_depth=args.depth)
                if packed:
                    ph.set_option("packed_records", 1)
                own = True
            pollh = ph.poll_demods_packed_raw if packed else ph.poll_demods_raw
            nbytes = n * nb_
            hptr, hview = irdm.host_alloc(nbytes)
            hview[:] = xq.view(torch.uint8).reshape(-1).cpu().numpy()
            for _ in range(2):
                ph.feed_host_ptr(hptr, n)
                ph.poll_bursts_raw(); ph.drop_frames(); pollh()
            torch.cuda.synchronize()
            th = time.perf_counter()
            nd = 0
            for _ in range(args.host_steps):
                ph.feed_host_ptr(hptr, n)
                ph.poll_bursts_raw(); ph.drop_frames(); nd += len(pollh())
            if args.depth:
                ph.flush()
                ph.poll_bursts_raw(); ph.drop_frames(); nd += len(pollh())
            torch.cuda.synchronize()
            hdt = time.perf_counter() - th
            pcie[name] = {"value": round(n * args.host_steps / hdt / 1e6, 2), "unit": "Msamples/s",
                          "h2d_GBps": round(nbytes * args.host_steps / hdt / 1e9, 2), "steps": args.host_steps,
                          "frames_per_step": round(nd / args.host_steps, 1),
                          "note": "irdm_feed_host from pinned host memory, %s (%d B/sample over PCIe); H2D of chunk k+1 "
                                  "overlaps the detector scan of chunk k" % (name, nb_)}
            irdm.host_free(hptr)
            if own:
                ph.close()
            del xq

    # ---- the same stages with one kernel on the chip at a time (pipeline_depth 0), and the chunk's records for the
    #      parity check below ----
    alone = None
    alone_clock = {}
    gpu_recs = None
    if rank == 0 and world == 1 and args.alone_steps > 0:
        p0 = irdm.Pipeline(fs, fmt=fmt, max_chunk_samples=n, max_bursts_per_chunk=8192, device=local, pipeline_depth=0)
        for kv in args.opt:
            key, val = kv.split("=")
            p0.set_option(key, int(val))
        p0.set_option("kernel_clock", 1)
        acc = {k: 0.0 for k in stage}
        for i in range(args.alone_steps):
            p0.feed_device(x.data_ptr(), n, None)
            if i == 0:
                gpu_recs = (p0.poll_bursts(), p0.poll_demods())
            else:
                p0.poll_bursts_raw(); p0.poll_demods_raw()
            p0.drop_frames()
            if i > 0 or args.alone_steps == 1:
                t = p0.timings()
                for kk in acc:
                    acc[kk] += t[kk]
        d = max(args.alone_steps - 1, 1)
        alone = {k: round(v / d, 4) for k, v in acc.items()}
        alone_clock = {}
        for which, key in ((0, "fir"), (1, "fft_mag")):
            sm, nl, last = p0.kernel_clock(which)
            alone_clock[key] = round(last, 4)           # (the last launch: the first ones include module loading)
        p0.close()

    # ---- BASELINE config 2: detect-only (K1 + scan, burst records) on the same chunk ----
    detect_only = None
    if rank == 0 and world == 1 and args.detect_steps > 0:
        pd = irdm.Pipeline(fs, fmt=fmt, max_chunk_samples=n, max_bursts_per_chunk=8192, device=local, pipeline_depth=args.depth)
        pd.set_option("detect_only", 1)
        for _ in range(2):
            pd.feed_device(x.data_ptr(), n, None); pd.poll_bursts_raw()
        torch.cuda.synchronize()
        td = time.perf_counter()
        nbd = 0
        for _ in range(args.detect_steps):
            pd.feed_device(x.data_ptr(), n, None)
            nbd += len(pd.poll_bursts_raw())
        if args.depth:
            pd.flush()
            nbd += len(pd.poll_bursts_raw())
        torch.cuda.synchronize()
        ddt = time.perf_counter() - td
        detect_only = {"value": round(n * args.detect_steps / ddt / 1e6, 2), "unit": "Msamples/s",
                       "ms_per_step": round(ddt / args.detect_steps * 1e3, 3), "steps": args.detect_steps,
                       "bursts_per_step": nbd / args.detect_steps,
                       "algorithmic_GBps": round((bps + 8.0) * n * args.detect_steps / ddt / 1e9, 1),
                       "note": "cfg2: K1 + prefilter + band scan, burst records only; B_det = %d B/sample" % (bps + 8)}
        pd.close()

    # ---- file -> RAW lines with the C99 binary (wall clock of the whole process, file in the page cache) ----
    file_to_raw = None
    if rank == 0 and world == 1 and args.file_run:
        try:
            import subprocess
            import tempfile
            shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
            ext = {"cf32": "cf32", "ci16": "ci16", "ci8": "ci8"}[args.format]
            path = os.path.join(shm, "irdm_bench_%d.%s" % (os.getpid(), ext))
            x.cpu().numpy().tofile(path)
            exe = os.path.join(ROOT, "iridium-sniffer_amd", "iridium-sniffer-hip")
            tf = time.perf_counter()
            r = subprocess.run([exe, "-f", path, "-r", str(fs), "--format", args.format, "--file-info", "bench", "--timing"],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            fdt = time.perf_counter() - tf
            os.remove(path)
            lines = r.stdout.count(b"\nRAW:") + (1 if r.stdout.startswith(b"RAW:") else 0)

            def timing_of(err):
                import re
                mt = re.search(rb"startup ([0-9.]+) s .* stream ([0-9.]+) s", err)
                return (float(mt.group(1)), float(mt.group(2))) if mt else (None, None)
            st_up, st_run = timing_of(r.stderr)
            file_to_raw = {"value": round(n / fdt / 1e6, 2), "unit": "Msamples/s", "wall_s": round(fdt, 3),
                           "startup_s": st_up, "stream_s": st_run,
                           "raw_lines": lines, "rc": r.returncode,
                           "note": "iridium-sniffer-hip -f <%d samples %s>, process start to exit (HIP init, "
                                   "context creation, fread from the page cache, H2D, GPU, RAW lines to a pipe)" % (n, args.format)}
            # BASELINE.md's only published workload: a 60 s / 600 Msample cf32 recording (12.0 s on an i7-11800H, AVX2).
            # Here: the chunk nine times over (604 Msamples, 4.8 GB) through the same binary.
            if args.file_run >= 2 and args.format == "cf32":
                import shutil
                reps = max(1, int(round(600e6 / n)))
                big_dir = shm if shutil.disk_usage(shm).free > (reps + 1) * n * bps else tempfile.gettempdir()
                big = os.path.join(big_dir, "irdm_bench_%d_big.cf32" % os.getpid())
                host_chunk = x.cpu().numpy()
                with open(big, "wb") as fbig:
                    for _ in range(reps):
                        host_chunk.tofile(fbig)
                del host_chunk
                tf = time.perf_counter()
                r = subprocess.run([exe, "-f", big, "-r", str(fs), "--format", "cf32", "--file-info", "bench", "--timing"],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                bdt = time.perf_counter() - tf
                os.remove(big)
                st_up, st_run = timing_of(r.stderr)
                file_to_raw["recording_600Msamples"] = {
                    "samples": reps * n, "wall_s": round(bdt, 3), "startup_s": st_up, "stream_s": st_run,
                    "value": round(reps * n / bdt / 1e6, 2), "unit": "Msamples/s", "rc": r.returncode,
                    "raw_lines": r.stdout.count(b"\nRAW:") + (1 if r.stdout.startswith(b"RAW:") else 0),
                    "published_reference_s": 12.0,
                    "note": "process start to exit; the reference's README publishes 12.0 s wall for its 600 Msample recording on "
                            "an i7-11800H (AVX2, no GPU); file read from %s" % big_dir}
        except Exception as e:
            file_to_raw = {"error": str(e)[:200]}

    # ---- CPU baseline: the oracle on a bounded prefix of the same stream (rank 0, N=1) ----
    cpu = None
    cpu1 = None
    parity_checked = None
    if rank == 0 and world == 1 and args.cpu_samples > 0:
        import orc
        m = min(n, args.cpu_samples) // 32768 * 32768
        if args.format == "cf32":
            host = x[:m].cpu().numpy().view(np.complex64).reshape(-1)
        else:
            host = x[:m].cpu().numpy().reshape(-1)
        # (a) one core, the whole path in one thread
        t1 = time.perf_counter()
        for _ in range(max(args.cpu_passes, 1)):
            ref = orc.run_stream(host, fs, fmt=int(fmt), cap_bursts=8192)
        cdt = time.perf_counter() - t1
        cpu1 = {"value": round(max(args.cpu_passes, 1) * m / cdt / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                "kind": "port",
                "sample": "%d passes over the first %d samples of the rank-0 stream (%.1f s of CPU), the C oracle in one thread "
                          "(reference --no-gpu algorithm, pinned FFT; the dispatched kernels in simd_avx2.c's operation ORDER but "
                          "written as scalar fmaf loops, not intrinsics -- a port's artefact: slower than the oracle's "
                          "--no-simd forms, 16 vs 23 Msamples/s), %d bursts -> %d RAW frames per pass"
                          % (max(args.cpu_passes, 1), m, cdt, ref.n_tagged, len(ref.demods))}
        # (b) the reference's thread layout: 1 detector thread -> 4 downmix workers -> 1 demod / output thread
        #     (main.c:175, :667-694); the oracle's stage functions release the GIL while they run
        try:
            if not args.cpu_layout:
                raise RuntimeError("skipped (--cpu-layout 0)")
            lay = cpu_reference_layout(orc, host, fs, int(fmt), workers=4)
            cpu = {"value": round(m / lay["seconds"] / 1e6, 3), "unit": "Msamples/s", "cores": 6, "kind": "port",
                   "sample": "one pass over the first %d samples in the reference's thread layout (1 detector + 4 downmix + "
                             "1 demod thread around the oracle's stage functions), %.1f s wall, %d bursts -> %d frames"
                             % (m, lay["seconds"], lay["bursts"], lay["demods"])}
            cpu["cpu_model"] = cpu_model()
            cpu["host_cores"] = os.cpu_count()
            # (b') "all host cores" (SURVEY 8d): downmix workers = cores - 2 beside the detector and the demod thread -- the
            #      detector thread is the serial part, so this saturates where the reference's layout does unless the host is small
            nw = max(1, (os.cpu_count() or 6) - 2)
            if nw != 4:
                lay2 = cpu_reference_layout(orc, host, fs, int(fmt), workers=nw)
                cpu["all_cores"] = {"value": round(m / lay2["seconds"] / 1e6, 3), "unit": "Msamples/s", "cores": nw + 2,
                                    "sample": "the same pass with %d downmix workers (%d busy threads), %.1f s wall" % (nw, nw + 2, lay2["seconds"])}
            else:
                cpu["all_cores"] = {"value": cpu["value"], "unit": "Msamples/s", "cores": 6, "sample": "this host has 6 cores: the layout above"}
        except Exception as e:
            cpu = dict(cpu1)
            cpu["sample"] += " [thread-layout run failed: %s]" % str(e)[:80]
        # (c) parity of the benchmark scene itself: the records of the stream's first TWO chunks through the HIP path -- the TIMED
        #     context's, when the warm-up fed at least three chunks (the bursts still active at the end of chunk 1 leave with
        #     chunk 2; the oracle, whose stream ends there, never emits them) -- against the oracle run over the two-chunk
        #     prefix; else the first chunk's, from the timed context or the pipeline_depth 0 context above
        which = None
        timed_llr = True
        chunks_checked = [0]
        hb = np.concatenate([a for a in head["bursts"] if len(a)]) if any(len(a) for a in head["bursts"]) else None
        hd = np.concatenate([a for a in head["demods"] if len(a)]) if any(len(a) for a in head["demods"]) else None
        if hb is not None and m == n:
            if head.get("chunks", 0) >= 3 and args.depth and args.parity_chunks >= 2:
                ref = orc.run_stream(np.concatenate([host, host]), fs, fmt=int(fmt), cap_bursts=16384)
                chunks_checked = [0, 1]
            gb_raw = hb[:len(ref.bursts)]
            ids = set(int(v) for v in gb_raw[:, irdm.Burst.id.offset:irdm.Burst.id.offset + 8].copy().view(np.uint64).reshape(-1))
            if hd is not None and len(hd):
                did = hd[:, irdm.Demod.id.offset:irdm.Demod.id.offset + 8].copy().view(np.uint64).reshape(-1)
                fd = hd[np.array([int(v) in ids for v in did])]
            else:
                fd = np.zeros((0, 1), np.uint8)
            if packed:
                fd = unpack_demods(irdm, fd) if len(fd) else fd      # (no LLRs in the compact records: the level is compared,
                timed_llr = False                                      # the LLRs by the whole-stream test of this configuration)
            gpu_recs = ([irdm.Burst.from_buffer_copy(bytes(r)) for r in gb_raw],
                        [irdm.Demod.from_buffer_copy(bytes(r)) for r in fd])
            which = "the timed context (pipeline_depth %d, in place %s, look-ahead %s)" % (args.depth, ingest, look)
        elif gpu_recs is not None:
            which = "a pipeline_depth 0 context on the same chunk"
        if gpu_recs is not None and m == n:
            gb, gd = gpu_recs
            ok = len(gb) == len(ref.bursts) and len(gd) == len(ref.demods)
            max_soft = 0.0
            first_bad = None

            def bad(what):
                nonlocal first_bad
                if first_bad is None:
                    first_bad = what
                return False
            if ok:
                for g, r_ in zip(gb, ref.bursts):
                    for fld in ("id", "start", "stop", "last_active", "center_bin", "num_samples"):
                        if getattr(g, fld) != getattr(r_, fld):
                            ok = bad("burst id %d: %s %r != %r" % (r_.id, fld, getattr(g, fld), getattr(r_, fld)))
                    for fld in ("magnitude", "noise"):
                        if np.float32(getattr(g, fld)).view(np.uint32) != np.float32(getattr(r_, fld)).view(np.uint32):
                            ok = bad("burst id %d: %s %r != %r" % (r_.id, fld, getattr(g, fld), getattr(r_, fld)))
                for g, r_ in zip(gd, ref.demods):
                    for fld in ("id", "timestamp", "n_symbols", "n_bits", "confidence", "direction"):
                        if getattr(g, fld) != getattr(r_, fld):
                            ok = bad("frame id %d: %s %r != %r" % (r_.id, fld, getattr(g, fld), getattr(r_, fld)))
                    if bytes(g.bits[:g.n_bits]) != bytes(r_.bits[:r_.n_bits]):
                        ok = bad("frame id %d: hard bits differ" % r_.id)
                    soft = max(abs(g.level - r_.level),
                               float(np.max(np.abs(np.array(g.llr[:g.n_bits], np.float32) - np.array(r_.llr[:r_.n_bits], np.float32))))
                               if (g.n_bits and timed_llr) else 0.0)
                    max_soft = max(max_soft, soft)
                if max_soft > 1e-4:
                    ok = bad("soft outputs differ by %g" % max_soft)
            else:
                bad("record counts differ")
            parity_checked = {"ok": bool(ok), "bursts": len(gb), "frames": len(gd), "oracle_bursts": len(ref.bursts),
                              "oracle_frames": len(ref.demods), "max_soft": max_soft, "records_of": which, "chunks": chunks_checked,
                              "first_mismatch": first_bad,
                              "what": "ids / indices / centre bins / dB fields / hard bits / confidence exact, level%s within 1e-4"
                                      % (" and LLR" if timed_llr else " (compact records carry no LLRs)")}

    # ---- the same pipeline fed chunks of twice the size (what a host that can afford 13 ms of latency per chunk gets: the
    #      scan's per-chunk launches and waits are spread over twice the samples) -- its own process, its own scene of the
    #      same density, parity-checked on its first chunk; reported beside the headline, never as `value` ----
    chunk_x2 = None
    if rank == 0 and world == 1 and args.big_chunk_steps > 0 and args.format == "cf32" and args.shard == "streams":
        try:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--samples", str(2 * n), "--steps", str(args.big_chunk_steps),
                   "--warmup", "5", "--density", str(args.density), "--sample-rate", str(fs), "--depth", str(args.depth),
                   "--packed", str(args.packed), "--ingest", str(args.ingest), "--lookahead", str(args.lookahead),
                   "--cpu-samples", str(2 * n if args.cpu_samples > 0 else 0), "--cpu-passes", "1", "--cpu-layout", "0",
                   "--parity-chunks", "1", "--host-steps", "0", "--alone-steps", "0", "--detect-steps", "0", "--file-run", "0",
                   "--big-chunk-steps", "0"]
            for kv in args.opt:
                cmd += ["--opt", kv]
            r2 = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
            line = [ln for ln in r2.stdout.decode().splitlines() if ln.startswith("{")]
            j2 = json.loads(line[-1])
            pc2 = j2.get("parity_checked") or {}
            chunk_x2 = {"value": j2["value"], "unit": "Msamples/s", "ms_per_step": j2["ms_per_step"], "steps": j2["steps"],
                        "samples_per_step": 2 * n, "bursts_per_step": j2["config"].get("bursts_per_step"),
                        "parity_ok": pc2.get("ok"), "parity_bursts": pc2.get("bursts"), "parity_frames": pc2.get("frames"),
                        "note": "this script run again with --samples %d (chunks of twice the size, same generator, density and "
                                "options), first chunk checked against the oracle; the headline `value` stays on %d-sample chunks"
                                % (2 * n, n)}
        except Exception as e:
            chunk_x2 = {"error": str(e)[:200]}

    if rank == 0:
        roofline["stage_ms_alone"] = alone
        if alone:
            roofline["kernel_clock_ms_alone"] = alone_clock
            a_ms = alone_clock.get(dom, 0) or alone.get(dom, 0)
            if a_ms > 0:
                roofline["achieved_alone"] = round(alg_bytes[dom] / (a_ms * 1e-3) / 1e9, 2)
                roofline["frac_alone"] = round(roofline["achieved_alone"] / HBM_PEAK_GBS, 5)
        out = {
            "metric": "IQ Msamples/s end-to-end (detect->demod), %d MHz %s" % (fs // 1_000_000, args.format),
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # ranks the collectives ran over and the backend that carried them ("nccl" = RCCL); per-rank Msamples/s
            "rccl_world": group_world if backend == "nccl" else 0, "collective_backend": backend if world > 1 else None,
            "collective_world": group_world,
            "rank_Msamples_per_s": {"min": round(min(rank_rates), 2), "max": round(max(rank_rates), 2)},
            "config": {"workload": "cfg3: %d MHz %s full pipeline (detect + downmix/FIR/CFO + Gardner DQPSK), "
                                   "%d-pt detect, %d samples/GPU/step resident in HBM, %.0f bursts/Msample"
                                   % (fs // 1_000_000, args.format, pipe.fft_size, n, args.density),
                       "samples_per_step_per_gpu": n, "bursts_per_step": totals["bursts"] / K,
                       "raw_frames_per_step": totals["demods"] / K, "parallelism": "streams x%d" % world,
                       "job_bursts_per_step": (int(counts[0].item()) / K) if world > 1 else totals["bursts"] / K,
                       "pipeline_depth": args.depth, "ingest_in_place": ingest, "lookahead": (1 if look else 0),
                       "packed_records": packed,      # the timed context returns 176-byte frame records (what frame_output_print reads: no LLRs)
                       "records": ({"produced": int(counts[1].item()), "sent": int(counts[2].item()),
                                    "gathered_on_rank0": int(gathered.item())} if world > 1 else None),
                       "scan": {k: pipe.stat(k) for k in ("scan_fast_chunks", "scan_fallbacks", "scan_dense_frames", "band_chunks",
                                                          "band_rounds", "band_retries", "band_aborts", "band_last_flags", "k1_lists",
                                                          "scan_chained", "scan_chain_undone",
                                                          "band_steps",      # (band_steps: update steps of the scans' last rounds, summed)
                                                          # scans that opened with round 1 behind a speculation pass (band_spec) / passes
                                                          "spec_scans", "spec_passes",
                                                          # later rounds' sums passes that began behind an unchanged prefix
                                                          "sum_restarts")},
                       # the plan pass's own phase stamps, us per chunk (round 1: verdict loops, boundaries, bitmaps read,
                       # step count scan, step list, slot scan, end; last verdict: loops, boundaries)
                       # (--opt band_timeline=1) device timeline of the scan's passes, us per chunk: [time from the first
                       # workgroup's start to the last one's end, idle time in front of the pass, launches per chunk]
                       "scan_timeline_us": ({("%s%d" % (("plan", "sums", "cross", "walk")[i % 4], i // 4) if i < 24 else
                                              ("commit", "history")[i - 24]):
                                             [round(pipe.stat("tl_dur_%d" % i) / 100.0 / max(1, pipe.stat("band_chunks")), 1),
                                              round(pipe.stat("tl_gap_%d" % i) / 100.0 / max(1, pipe.stat("band_chunks")), 1),
                                              round(pipe.stat("tl_n_%d" % i) / max(1, pipe.stat("band_chunks")), 2)]
                                             for i in range(26) if pipe.stat("tl_n_%d" % i) > 0} or None),
                       "walk_events": ({"round0_longest_lane": pipe.stat("tl_dur_26") / max(1, pipe.stat("band_chunks")),
                                        "round0_all": pipe.stat("tl_dur_27") / max(1, pipe.stat("band_chunks")),
                                        "later_longest_lane": pipe.stat("tl_dur_28") / max(1, pipe.stat("band_chunks")),
                                        "later_all": pipe.stat("tl_dur_29") / max(1, pipe.stat("band_chunks")),
                                        "later_busiest_wave": pipe.stat("tl_dur_30") / max(1, pipe.stat("band_chunks")),
                                        "later_longest_wave_us": pipe.stat("tl_dur_31") / 100.0 / max(1, pipe.stat("band_chunks"))}
                                       if pipe.stat("tl_n_0") > 0 else None),
                       "plan_phase_us": [round(pipe.stat("plan_tp_%d" % i) / 100.0 / max(1, pipe.stat("band_chunks")), 1)
                                         for i in range(16)],
                       "host_us_total": {k: pipe.stat("host_us_%d" % i) for i, k in enumerate(
                           ("k1_ring_enqueue", "settle", "chain_enqueue", "scan_enqueue", "wait_older_chain", "final_sync",
                            "settle_wait_scan", "settle_counters", "settle_records", "build_records"))}},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_1core": cpu1,
            "parity_checked": parity_checked,
            "detect_only": detect_only,
            "file_to_raw": file_to_raw,
            "pcie_inclusive": pcie,
            "chunk_x2": chunk_x2,
        }
        if world > 1 and not (out["config"]["records"]["produced"] == out["config"]["records"]["sent"] ==
                              out["config"]["records"]["gathered_on_rank0"]):
            raise SystemExit("bench: record gather lost frames: %r" % (out["config"]["records"],))
        print(json.dumps(out), flush=True)
    pipe.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
