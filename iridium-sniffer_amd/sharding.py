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
        flush                 the chunk's per-burst stages, overlapping the next ranks' scans

    With "nccl" the blob is a device tensor (RCCL send/recv over xGMI); with "gloo" (CPU tests, or several ranks
    sharing one GPU) it bounces through a host tensor.  Expected throughput: world * chunk / max(world * t_hop,
    t_rank), t_hop = recv + import + scan + export + send, t_rank = t_hop + K1 + per-burst chain -- the state chain
    caps the speed-up at t_rank / t_hop however many GPUs there are."""

    def __init__(self, dist, pipe, torch, device, chunk_samples, bps, overlap):
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
        self.have_head = self.have_hist = False       # rank 0: parts of the last rank's state already received

    def _recv_part(self, src, lo, hi):
        if self.nccl:
            self.dist.recv(self.state[lo:hi], src=src)
        else:
            self.dist.recv(self.host_state[lo:hi], src=src)
            self.state[lo:hi].copy_(self.host_state[lo:hi])
        if self.device.type == "cuda":          # (a CPU device only in the emulated tests: nothing is in flight there)
            # the stream the part arrived on, NOT the device: this rank's scan may be on the GPU waiting for exactly this
            # part (irdm_expect_history), and a device-wide wait would wait for the scan
            self.torch.cuda.current_stream(self.device).synchronize()

    def _send_part(self, dst, lo, hi):
        if self.nccl:
            self.dist.send(self.state[lo:hi], dst=dst)
        else:
            self.host_state[lo:hi].copy_(self.state[lo:hi])
            self.dist.send(self.host_state[lo:hi], dst=dst)

    def step(self, buf, first_of_stream):
        """buf: device uint8 tensor holding [overlap samples | chunk samples] for this rank's chunk of the current
        super-step (the overlap part is ignored for the very first chunk of the stream).  Returns after the chunk's
        records are in the pipeline's queues.  Every rank's point-to-point traffic of a super-step is complete when
        step() returns on all ranks (rank 0 takes the last rank's state at the END of its step and keeps it for the
        next one), so collectives may follow."""
        pipe, rank, world = self.pipe, self.rank, self.world
        abs_start = (self.step_no * world + rank) * self.chunk
        base = buf.data_ptr()
        if world == 1:                   # nothing to hand over: the context carries its own state
            pipe.feed_begin(base + self.overlap * self.bps, self.chunk, None)
            pipe.feed_end()
            pipe.flush()
            self.step_no += 1
            return
        first = first_of_stream and self.step_no == 0 and rank == 0
        head, nbytes, sp = self.head, self.nbytes, self.state.data_ptr()
        if not first:
            pipe.seed_history_device(base, self.overlap, abs_start)
        pipe.feed_begin(base + self.overlap * self.bps, self.chunk, None)
        late_history = False
        if not first:
            prev = (rank - 1) % world
            if not self.have_head:
                self._recv_part(prev, 0, head)
            # (rank 0 received what had arrived of the last rank's state at the end of its previous step)
            pipe.import_state_head_device(sp, head)
            # the history follows while K1 and round 0 of the scan run -- if this scan can wait for it on the device
            # (a primed detector, the band scan, pipeline_depth >= 1, a real GPU: the emulated device runs a launch to
            # its end when it is enqueued)
            late_history = (not self.have_hist) and self.device.type == "cuda" and pipe.expect_history(sp + head)
            if not late_history:
                if not self.have_hist:
                    self._recv_part(prev, head, nbytes)
                pipe.import_state_history_device(sp + head, nbytes - head)
        pipe.feed_end()
        if late_history:
            self._recv_part((rank - 1) % world, head, nbytes)
            pipe.import_state_history_device(sp + head, nbytes - head)
        self.have_head = self.have_hist = False
        pipe.export_state_device(sp, nbytes)             # (waits for this chunk's scan)
        nxt = (rank + 1) % world
        self._send_part(nxt, 0, head)                    # the next rank's scan can start: head first ...
        self._send_part(nxt, head, nbytes)               # ... the history behind it
        pipe.flush()
        if rank == 0:
            self._recv_part(world - 1, 0, head)
            self._recv_part(world - 1, head, nbytes)
            self.have_head = self.have_hist = True
        self.step_no += 1

    def drain(self):
        """Nothing left in flight (kept for callers of the earlier protocol)."""
        return
