"""Multi-GPU plumbing for the hot path, one process per GPU over
torch.distributed ("nccl" == RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Two ways the path shards:
  streams      independent IQ streams, one per rank: no data-path collective, only the
               gather of demodulated-frame records to rank 0 (what bench.py --gpus N runs).
  time chunks  ONE stream cut into contiguous chunks of whole feed blocks.  The per-frame
               FFT/|.|^2 and all per-burst work are independent; the detector state machine is
               not (512-frame noise ring, active bursts, burst ids), so rank k receives the
               detector state from rank k-1 (point-to-point send/recv: 16-32 MiB, ~0.2 ms on one
               xGMI link) and each rank also loads the `overlap` samples that precede its
               chunk so burst windows can reach back across the boundary.
"""
import numpy as np


def chunk_plan(total_samples, world, block=32768, overlap=0):
    """Contiguous time-chunks of whole feed blocks (the last chunk takes the ragged tail).

    Returns [(start, stop, history_start)] per rank; history_start..start is the overlap
    (clamped at 0) a rank must load in front of its chunk."""
    n_blocks = total_samples // block
    per = [n_blocks // world + (1 if r < n_blocks % world else 0) for r in range(world)]
    plan, pos = [], 0
    for r in range(world):
        start = pos
        stop = start + per[r] * block
        if r == world - 1:
            stop = total_samples
        plan.append((start, stop, max(0, start - overlap)))
        pos = stop
    return plan


def required_overlap(sample_rate, fft_size):
    """Samples a rank must see before its chunk: the reference's ring (stale-slot reads reach one
    ring length back, burst_detect.c:292-296, :401-422) plus the longest burst window."""
    max_len = int(sample_rate * 0.09)
    post = int(sample_rate * 16e-3)
    pre = 2 * fft_size
    ring = max(2 * sample_rate, max_len + pre + post + 4 * fft_size)
    return ring + max_len + post + pre + 2 * fft_size


def gather_records(dist, records, rec_size, cap, device=None):
    """Gather fixed-size records (uint8 [n, rec_size]) from every rank to rank 0.

    One padded dist.gather of cap*rec_size bytes plus the counts: burst records are ~0.5-4.5 KB
    each, so this is latency-, not bandwidth-bound.  Returns list-per-rank on rank 0, else None."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    recs = np.ascontiguousarray(records, np.uint8).reshape(-1, rec_size)
    k = min(len(recs), cap)
    buf = torch.zeros(cap * rec_size + 8, dtype=torch.uint8, device=device)
    hdr = np.array([k], np.int64).view(np.uint8)
    buf[:8] = torch.from_numpy(hdr.copy()).to(buf.device)
    if k:
        buf[8:8 + k * rec_size] = torch.from_numpy(recs[:k].reshape(-1).copy()).to(buf.device)
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    res = []
    for t in out:
        a = t.cpu().numpy()
        n = int(a[:8].view(np.int64)[0])
        res.append(a[8:8 + n * rec_size].reshape(n, rec_size).copy())
    return res


class RecordGather:
    """The streams-mode record gather of bench.py (BASELINE config 5 / the headline metric at N > 1): every rank sends the
    compact frame records of a step -- fixed-size messages of a count word + `cap` records of `rec_bytes` -- to rank 0 with an
    asynchronous dist.gather, double-buffered so that the collective of step i runs beside the detector scan of step i + 1.
    Nothing is ever dropped silently: more records than the message holds is an error, and rank 0 counts what arrived from
    the messages' count words (`gathered`), which the caller compares with what the ranks produced and sent."""

    def __init__(self, dist, torch, rank, world, cap, rec_bytes, device, pin=True):
        self.dist, self.torch, self.rank, self.world, self.cap, self.rec = dist, torch, rank, world, cap, rec_bytes
        self.msg = 8 + cap * rec_bytes
        mk = lambda dev=None: torch.zeros((self.msg,), dtype=torch.uint8, device=dev)          # noqa: E731
        self.host = [mk().pin_memory() if pin else mk() for _ in range(2)]
        self.bufs = [mk(device) for _ in range(2)]
        self.lists = [[mk(device) for _ in range(world)] for _ in range(2)] if rank == 0 else [None, None]
        self.work = [None, None]
        self.gathered = torch.zeros((1,), dtype=torch.int64, device=device)      # rank 0: records received
        self.step = 0
        self.sent = 0

    def wait(self, slot):
        if self.work[slot] is None:
            return
        self.work[slot].wait()
        self.work[slot] = None
        if self.rank == 0:
            self.gathered.add_(self.torch.stack([l[:8] for l in self.lists[slot]]).view(self.torch.int64).sum())

    def send(self, records):
        """records: uint8 [k, rec_bytes] (numpy).  Returns k."""
        slot = self.step & 1
        self.step += 1
        self.wait(slot)
        k = len(records)
        if k > self.cap:
            raise SystemExit("record gather: %d records in one step exceed the message (%d)" % (k, self.cap))
        hb = self.host[slot].numpy()
        hb[:8] = np.array([k], dtype=np.int64).view(np.uint8)
        if k:
            hb[8:8 + k * self.rec] = np.ascontiguousarray(records, np.uint8).reshape(-1)
        self.bufs[slot].copy_(self.host[slot], non_blocking=True)
        self.work[slot] = self.dist.gather(self.bufs[slot], self.lists[slot], dst=0, async_op=True)
        self.sent += k
        return k

    def finish(self):
        for slot in (0, 1):
            self.wait(slot)

    def reset_counts(self):
        self.finish()
        self.gathered.zero_()
        self.sent = 0


def max_over_ranks(dist, value, device=None):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def handoff_recv(dist, nbytes, device=None):
    """Rank k>0: receive the detector state blob from rank k-1."""
    import torch
    t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.recv(t, src=dist.get_rank() - 1)
    return t.cpu().numpy()


def handoff_send(dist, blob, device=None):
    """Rank k<world-1: send the detector state blob to rank k+1."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(blob, np.uint8))
    if device is not None:
        t = t.to(device)
    dist.send(t, dst=dist.get_rank() + 1)


def run_time_sharded(dist, make_pipeline, iq, fmt_is_cf32, sample_rate, fft_size, block=32768, device=None):
    """Process ONE stream `iq` (host array, every rank holds or can read it) across all ranks.

    make_pipeline() -> object with feed_host/seed_history/import_state/export_state/
    poll_demods_raw (irdm.Pipeline).  Ranks scan in order (state hand-off); everything after the
    scan of a chunk (downmix, demod) overlaps with the next rank's scan.  Returns the gathered
    record arrays on rank 0."""
    rank, world = dist.get_rank(), dist.get_world_size()
    per = 1 if fmt_is_cf32 else 2
    n = len(iq) // per
    plan = chunk_plan(n, world, block, required_overlap(sample_rate, fft_size))
    start, stop, hist0 = plan[rank]
    pipe = make_pipeline(stop - start)
    if rank > 0:
        pipe.seed_history(iq[hist0 * per:start * per], start)
        blob = handoff_recv(dist, pipe.L.irdm_state_bytes(pipe.h), device)
        pipe.import_state(blob)
    if rank < world - 1:
        # detector pass of this chunk must finish before the state can move on
        pipe.feed_host(iq[start * per:stop * per])
        handoff_send(dist, pipe.export_state(), device)
    else:
        pipe.feed_host(iq[start * per:stop * per])
    if hasattr(pipe, "flush"):
        pipe.flush()                     # pipeline_depth >= 1: the chunk's per-burst work is still in flight
    recs = pipe.poll_demods_raw()
    return gather_records(dist, recs, recs.shape[1] if recs.size else 4544, 4096, device)


class TimeShard:
    """ONE stream, processed in super-steps of `world` consecutive chunks, chunk k of a super-step on rank k
    (BASELINE config 4).  Per rank and super-step:

        seed_history_device   the `overlap` samples in front of the chunk (arrive with the chunk)
        feed_begin            K1 + history-ring copy: no detector state needed, runs while the state is on its way
        recv + import head    of the previous chunk's detector state (rank k-1; rank 0: rank world-1's of the previous
                              super-step): header + DetState + baseline sums, 65 KB at 12 MHz, GPU to GPU
        feed_end              scan enqueued (+ per-burst chain of the chunk before); round 0 reads no history
        recv + import history the 512-frame history (16-32 MiB) arrives while K1 and round 0 run; round 1's sums pass
                              waits for it on the device (irdm_expect_history)
        export + send         head, then history, as soon as the scan has settled; only recv head -> scan -> send head
                              is sequential across the ranks (burst_detect.c:438-454, :594-631)
        advance               the chunk's per-burst stages ENQUEUED (irdm_advance): they overlap the next ranks' scans and
                              this rank's next super-step (scatter, K1); records one step late, drain() at the end

    With "nccl" the blob is a device tensor (RCCL send/recv over xGMI); with "gloo" (CPU tests, or several ranks
    sharing one GPU) it bounces through a host tensor.  Expected throughput: world * chunk / max(world * t_hop,
    t_rank), t_hop = recv + import + scan + export + send, t_rank = t_hop + K1 + per-burst chain -- the state chain
    caps the speed-up at t_rank / t_hop however many GPUs there are."""

    def __init__(self, dist, pipe, torch, device, chunk_samples, bps, overlap, overlap_chain=True):
        self.dist, self.pipe, self.torch, self.device = dist, pipe, torch, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.chunk, self.bps, self.overlap = chunk_samples, bps, overlap
        self.nccl = dist.get_backend() == "nccl"
        self.nbytes = pipe.state_bytes()
        # the blob travels in two messages: the head (header, detector state, baseline sums: what round 0 of the scan
        # reads; 65 KB at 12 MHz) and the 512-frame history behind it (first read by round 1's sums pass)
        self.head = pipe.state_head_bytes()
        self.state = torch.empty(self.nbytes, dtype=torch.uint8, device=device)
        self.host_state = None if self.nccl else torch.empty(self.nbytes, dtype=torch.uint8)
        self.step_no = 0
        # overlap_chain: a step ends with irdm_advance (the chunk's per-burst chain enqueued, not waited for): the chain runs
        # beside the next super-step's scatter, K1 and scan, and its records come out one step later (drain() brings the
        # last ones).  False: irdm_flush at the end of every step (records of a chunk with its own step).
        self.overlap_chain = overlap_chain and hasattr(pipe.L, "irdm_advance")
        self.pending = None               # rank 0: (work_head, work_history) of the last rank's state, receives posted ahead
        # host time per part of a step, seconds, summed (bench.py reports them per super-step: what a rank waits for where)
        self.t = {k: 0.0 for k in ("seed_k1", "recv_head", "scan_enqueue", "recv_history", "scan_settle", "send", "chain")}
        self.steps_timed = 0

    # ---- the two parts of the state blob: receives may be posted ahead (rank 0), sends block ----
    def _post_recv(self, src, lo, hi):
        t = self.state if self.nccl else self.host_state
        return self.dist.irecv(t[lo:hi], src=src)

    def _finish_recv(self, work, lo, hi):
        work.wait()
        if not self.nccl:
            self.state[lo:hi].copy_(self.host_state[lo:hi])
        if self.device.type == "cuda":          # (a CPU device only in the emulated tests: nothing is in flight there)
            # the stream the part arrived on, NOT the device: this rank's scan may be on the GPU waiting for exactly this
            # part (irdm_expect_history), and a device-wide wait would wait for the scan
            self.torch.cuda.current_stream(self.device).synchronize()

    def _recv_part(self, src, lo, hi):
        self._finish_recv(self._post_recv(src, lo, hi), lo, hi)

    def _send_part(self, dst, lo, hi):
        if self.nccl:
            self.dist.send(self.state[lo:hi], dst=dst)
        else:
            self.host_state[lo:hi].copy_(self.state[lo:hi])
            self.dist.send(self.host_state[lo:hi], dst=dst)

    def step(self, buf, first_of_stream):
        """buf: device uint8 tensor holding [overlap samples | chunk samples] for this rank's chunk of the current
        super-step (the overlap part is ignored for the very first chunk of the stream).  Returns when this rank's scan has
        settled and its state is on its way; with overlap_chain the chunk's per-burst chain is still running (its records
        reach the pipeline's queues during the next step, the last ones at drain()).  Rank 0 has POSTED the receives of the
        last rank's state and completes them at the start of its next step: the only point-to-point traffic that crosses a
        step boundary, matched by the last rank's sends of this step."""
        pipe, rank, world = self.pipe, self.rank, self.world
        abs_start = (self.step_no * world + rank) * self.chunk
        base = buf.data_ptr()
        if world == 1:                   # nothing to hand over: the context carries its own state
            pipe.feed_begin(base + self.overlap * self.bps, self.chunk, None)
            pipe.feed_end()
            pipe.flush()
            self.step_no += 1
            return
        import time
        t = self.t
        tp = [time.perf_counter()]

        def lap(key):
            now = time.perf_counter()
            t[key] += now - tp[0]
            tp[0] = now
        first = first_of_stream and self.step_no == 0 and rank == 0
        head, nbytes, sp = self.head, self.nbytes, self.state.data_ptr()
        if not first:
            pipe.seed_history_device(base, self.overlap, abs_start)
        pipe.feed_begin(base + self.overlap * self.bps, self.chunk, None)
        lap("seed_k1")
        late_history = False
        w_hist = None
        if not first:
            prev = (rank - 1) % world
            if self.pending is not None:
                w_head, w_hist = self.pending            # (rank 0: posted at the end of the previous step)
                self.pending = None
            else:
                w_head = self._post_recv(prev, 0, head)
            self._finish_recv(w_head, 0, head)
            pipe.import_state_head_device(sp, head)
            # the history follows while K1 and round 0 of the scan run -- if this scan can wait for it on the device
            # (a primed detector, the band scan, pipeline_depth >= 1, a real GPU: the emulated device runs a launch to
            # its end when it is enqueued)
            late_history = self.device.type == "cuda" and pipe.expect_history(sp + head)
            if w_hist is None:
                w_hist = self._post_recv(prev, head, nbytes)
            lap("recv_head")
            if not late_history:
                self._finish_recv(w_hist, head, nbytes)
                pipe.import_state_history_device(sp + head, nbytes - head)
                lap("recv_history")
        pipe.feed_end()
        lap("scan_enqueue")
        if late_history:
            self._finish_recv(w_hist, head, nbytes)
            pipe.import_state_history_device(sp + head, nbytes - head)
            lap("recv_history")
        pipe.export_state_device(sp, nbytes)             # (waits for this chunk's scan)
        lap("scan_settle")
        nxt = (rank + 1) % world
        self._send_part(nxt, 0, head)                    # the next rank's scan can start: head first ...
        self._send_part(nxt, head, nbytes)               # ... the history behind it
        lap("send")
        if self.overlap_chain:
            pipe.advance()                               # this chunk's per-burst chain: enqueued, not waited for
        else:
            pipe.flush()
        lap("chain")
        self.steps_timed += 1
        if rank == 0:
            # the last rank's state of this super-step is what rank 0's next chunk starts from: the receives are posted
            # now and completed when the next step needs them -- rank 0 does not idle until the last rank's scan settles
            self.pending = (self._post_recv(world - 1, 0, head), self._post_recv(world - 1, head, nbytes))
        self.step_no += 1

    def drain(self):
        """End of the stream: the chains still in flight finish (their records become pollable) and rank 0 takes delivery of
        the last rank's final state, so that no point-to-point message is left unmatched."""
        if self.world > 1:
            self.pipe.flush()
            if self.pending is not None:
                self._finish_recv(self.pending[0], 0, self.head)
                self._finish_recv(self.pending[1], self.head, self.nbytes)
                self.pending = None
        return
