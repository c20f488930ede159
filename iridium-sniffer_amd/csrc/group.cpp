// group.cpp -- ONE stream across several GPUs of one process (include/irdm_hip.h "a group").
//
// The reference's host runs one detector thread, four downmix threads and a demodulator thread over queues
// (main.c:667-694); the detector is sequential across frames (noise-floor ring, active bursts, ids:
// burst_detect.c:438-454, :594-631), everything behind it is per burst.  A group keeps that shape across GPUs: the stream
// is cut into chunks, chunk k lives on member k mod N, and the only thing that travels in stream order is the detector's
// state -- from the member that scanned chunk k to the member that scans chunk k + 1 -- while K1 (the FFT magnitudes of a
// chunk need no state) and the per-burst chains of all members run side by side.
//
// Built on the single-context C-ABI (irdm_seed_history_device, irdm_feed_begin / irdm_feed_end, the two-part state
// export / import, irdm_advance, irdm_poll_chunk_marks): this file drives N contexts from one thread and moves bytes
// between their devices with RCCL (grouped ncclSend / ncclRecv pairs on two communicator sets: IQ samples, detector state).
// Nothing here computes on samples.
//
// What a super-step (N consecutive chunks, one per member) does, in the order the calls are made:
//   stage     per member: the chunk from the host (one hipMemcpyAsync per member, each over its own PCIe link) or from
//             member 0's memory (RCCL scatter), into the member's landing buffer [overlap | chunk]; the overlap -- the
//             samples in front of the chunk that burst windows and the reference's stale ring reads reach back to -- is
//             the tail of the PREVIOUS member's landing buffer, GPU to GPU, behind that member's own upload on its
//             transfer stream.  Two landing buffers per member: the next super-step is staged while this one computes.
//   K1        per member: irdm_seed_history_device (overlap into the ring), irdm_feed_begin (K1 + ring copy): all
//             members' K1s run at once.
//   hops      member by member: wait for the previous member's head (header, DetState, baseline sums: 65 KB at 12 MHz),
//             import it, irdm_expect_history + irdm_feed_end (the scan starts; its round 0 reads no history), import the
//             512-frame history (16-32 MiB) when it has arrived, irdm_export_state_device (returns when the scan has
//             settled), ncclSend head then history to the next member, irdm_advance (this chunk's per-burst chain
//             enqueued, not waited for).  recv head -> scan -> send head is the one sequential path across the members.
//   records   each member's queues are drained by its chunk marks into a store keyed by the stream's chunk number; a
//             chunk's records leave the group when every earlier chunk is complete (irdm_chunks_complete).
// The host stays in the hop (it reads the burst count of a scan and carries the stream position, the burst count and the
// history index in the blob's header): ~50 us of the ~0.8 ms a hop takes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <vector>

#ifndef IRDM_RCCL_EMULATED
#include <dlfcn.h>
#endif

#include <unistd.h>

#include "../../include/irdm_hip.h"
#include "kernels.hpp"

namespace {

#define GRP_HIP(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            fprintf(stderr, "irdm_hip group: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                                       \
        }                                                                                                    \
    } while (0)

// ---- RCCL: loaded on first use (a process that never makes a group never maps librccl) ----
struct Rccl {
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    bool ok;
};

Rccl *rccl()
{
    static Rccl r = {};
    static bool tried = false;
    if (tried) return r.ok ? &r : nullptr;
    tried = true;
#ifdef IRDM_RCCL_EMULATED
    r.CommInitAll = ncclCommInitAll;
    r.CommDestroy = ncclCommDestroy;
    r.GroupStart = ncclGroupStart;
    r.GroupEnd = ncclGroupEnd;
    r.Send = ncclSend;
    r.Recv = ncclRecv;
    r.GetErrorString = ncclGetErrorString;
    r.ok = true;
#else
    const char *names[] = { getenv("IRDM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *h = nullptr;
    for (const char *n : names)
        if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!h) {
        fprintf(stderr, "irdm_hip group: librccl not found (%s); set IRDM_RCCL_LIB\n", dlerror());
        return nullptr;
    }
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetErrorString;
    if (!r.ok) fprintf(stderr, "irdm_hip group: librccl lacks an entry point this library needs\n");
#endif
    return r.ok ? &r : nullptr;
}

#define GRP_NCCL(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t e_ = (expr);                                                                            \
        if (e_ != ncclSuccess) {                                                                             \
            fprintf(stderr, "irdm_hip group: %s failed: %s (%s:%d)\n", #expr, rccl()->GetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                                       \
        }                                                                                                    \
    } while (0)

// the records of one chunk of the stream, in the order the member queued them
struct ChunkRecords {
    std::vector<irdm_burst_t> bursts;
    std::vector<irdm_frame_info_t> frames;
    std::vector<std::vector<float>> frame_samples;
    std::vector<irdm_demod_t> demods;
    std::vector<irdm_demod_packed_t> packed;
    std::vector<irdm_decoded_t> decoded;
    std::vector<irdm_ida_t> ida;
};

struct Member {
    irdm_pipeline_t *p = nullptr;
    int dev = 0;
    hipStream_t xs = nullptr;            // IQ samples: uploads, scatter and overlap transfers
    hipStream_t hs = nullptr;            // detector state blobs
    void *land[2] = { nullptr, nullptr };
    void *chunk_at[2] = { nullptr, nullptr };   // where the staged chunk's samples are: behind the overlap in land[], or in the ring
    void *blob_out = nullptr, *blob_in = nullptr;
    hipEvent_t ev_head = nullptr, ev_hist = nullptr;      // on hs: the previous member's head / history has arrived in blob_in
    hipEvent_t ev_land[2] = { nullptr, nullptr };         // on xs: the landing buffer holds its chunk and overlap
    uint64_t fed = 0;                    // chunks this member has been fed
    uint64_t staged = 0;                 // chunks staged into its landing buffers (fed <= staged <= fed + 2)
    bool state_posted = false;           // a head + history are on their way into blob_in (or there)
};

struct Staged {                          // one staged super-step
    const void *src = nullptr;
    size_t n = 0;
    bool host = false;
    std::vector<size_t> lens;            // its chunks
};

} // namespace

struct irdm_group {
    int n = 0;
    std::vector<Member> m;
    std::vector<ncclComm_t> c_iq, c_state;
    bool comms = false;
    size_t chunk = 0, ov = 0, bps = 0, state_bytes = 0, head_bytes = 0;
    int feed_block = 0;
    uint64_t next_chunk = 0;             // chunks fed so far (the stream's chunk number of the next one)
    uint64_t next_stage = 0;             // chunks staged so far
    uint64_t abs_fed = 0, abs_staged = 0;     // samples
    size_t prev_len = 0;                 // length of the last staged chunk (where its tail is)
    std::deque<Staged> staged;           // at most two: the super-step being fed and the one behind it
    bool closed = false;
    bool broken = false;                 // a transfer or a staging step failed midway: the counters of the chunks staged so far
                                         // may be ahead of what was fed -- every later stage / feed / flush call fails cleanly
    int loopback = 0, keep_samples = 0;
    // merged output
    std::map<uint64_t, ChunkRecords> store;
    uint64_t next_emit = 0;
    std::deque<irdm_burst_t> q_bursts;
    std::deque<irdm_frame_info_t> q_frames;
    std::deque<std::vector<float>> q_frame_samples;
    std::deque<irdm_demod_t> q_demods;
    std::deque<irdm_demod_packed_t> q_packed;
    std::deque<irdm_decoded_t> q_decoded;
    std::deque<irdm_ida_t> q_ida;
    uint64_t st_hops = 0, st_hop_bytes = 0, st_scatter_bytes = 0, st_overlap_bytes = 0, st_late = 0;
    std::vector<float> tmp_samples;
};

namespace {

bool full_protocol(const irdm_group *g) { return g->n > 1 || g->loopback; }

// one point-to-point transfer between two members' streams (src == dst: a plain device copy)
int transfer(irdm_group *g, std::vector<ncclComm_t> &comms, int src, const void *from, hipStream_t s_src, int dst, void *to,
             hipStream_t s_dst, size_t bytes)
{
    if (bytes == 0) return 0;
    if (src == dst && !g->loopback) {
        // (hipMemcpyAsync, i.e. the DMA engines, beside the kernels)
        GRP_HIP(hipSetDevice(g->m[dst].dev));
        GRP_HIP(hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, s_dst));
        return 0;
    }
    Rccl *r = rccl();
    GRP_NCCL(r->GroupStart());
    // (whatever fails between GroupStart and GroupEnd, the group is closed again: an open one would swallow every later call)
    ncclResult_t e1 = ncclSuccess, e2 = ncclSuccess;
    const bool d1 = hipSetDevice(g->m[src].dev) == hipSuccess;
    if (d1) e1 = r->Send(from, bytes, ncclUint8, dst, comms[src], s_src);
    const bool d2 = d1 && hipSetDevice(g->m[dst].dev) == hipSuccess;
    if (d2) e2 = r->Recv(to, bytes, ncclUint8, src, comms[dst], s_dst);
    const ncclResult_t e3 = r->GroupEnd();
    if (!d1 || !d2 || e1 != ncclSuccess || e2 != ncclSuccess || e3 != ncclSuccess) {
        fprintf(stderr, "irdm_hip group: ncclSend / ncclRecv %d -> %d failed: %s\n", src, dst,
                !d1 || !d2 ? "hipSetDevice" : r->GetErrorString(e1 != ncclSuccess ? e1 : e2 != ncclSuccess ? e2 : e3));
        g->broken = true;
        return -1;
    }
    return 0;
}

// a member's queues -> the store, mark by mark
int drain_member(irdm_group *g, int r)
{
    Member &mb = g->m[r];
    irdm_chunk_mark_t marks[16];
    int k;
    while ((k = irdm_poll_chunk_marks(mb.p, marks, 16)) > 0) {
        for (int i = 0; i < k; i++) {
            const irdm_chunk_mark_t &mk = marks[i];
            // member-local chunk c of member r is chunk c * N + r of the stream
            ChunkRecords &cr = g->store[mk.chunk * (uint64_t)g->n + (uint64_t)r];
            size_t at;
            at = cr.bursts.size();
            cr.bursts.resize(at + mk.n_bursts);
            if (mk.n_bursts && irdm_poll_bursts(mb.p, cr.bursts.data() + at, (int)mk.n_bursts) != (int)mk.n_bursts) return -1;
            for (uint32_t j = 0; j < mk.n_frames; j++) {
                irdm_frame_info_t fi;
                if (g->keep_samples) {
                    g->tmp_samples.resize((size_t)2 * IRDM_MAX_FRAME_SAMPLES);
                    if (irdm_poll_frames(mb.p, &fi, g->tmp_samples.data(), 1) != 1) return -1;
                    const size_t ns = fi.drop_reason == 0 ? (size_t)2 * (size_t)fi.num_samples : 0;
                    cr.frame_samples.emplace_back(g->tmp_samples.begin(), g->tmp_samples.begin() + ns);
                } else {
                    if (irdm_poll_frames(mb.p, &fi, nullptr, 1) != 1) return -1;
                    cr.frame_samples.emplace_back();
                }
                cr.frames.push_back(fi);
            }
            at = cr.demods.size();
            cr.demods.resize(at + mk.n_demods);
            if (mk.n_demods && irdm_poll_demods(mb.p, cr.demods.data() + at, (int)mk.n_demods) != (int)mk.n_demods) return -1;
            at = cr.packed.size();
            cr.packed.resize(at + mk.n_packed);
            if (mk.n_packed && irdm_poll_demods_packed(mb.p, cr.packed.data() + at, (int)mk.n_packed) != (int)mk.n_packed) return -1;
            at = cr.decoded.size();
            cr.decoded.resize(at + mk.n_decoded);
            if (mk.n_decoded && irdm_poll_decoded(mb.p, cr.decoded.data() + at, (int)mk.n_decoded) != (int)mk.n_decoded) return -1;
            at = cr.ida.size();
            cr.ida.resize(at + mk.n_ida);
            if (mk.n_ida && irdm_poll_ida(mb.p, cr.ida.data() + at, (int)mk.n_ida) != (int)mk.n_ida) return -1;
        }
    }
    return k < 0 ? -1 : 0;
}

// chunks of the stream whose records are complete leave the store, in order
void emit_ready(irdm_group *g)
{
    for (;;) {
        if (g->next_emit >= g->next_chunk) return;
        const int r = (int)(g->next_emit % (uint64_t)g->n);
        const uint64_t local = g->next_emit / (uint64_t)g->n;
        if (local >= irdm_chunks_complete(g->m[r].p)) return;
        auto it = g->store.find(g->next_emit);
        if (it != g->store.end()) {
            ChunkRecords &cr = it->second;
            g->q_bursts.insert(g->q_bursts.end(), cr.bursts.begin(), cr.bursts.end());
            g->q_frames.insert(g->q_frames.end(), cr.frames.begin(), cr.frames.end());
            for (auto &v : cr.frame_samples) g->q_frame_samples.push_back(std::move(v));
            g->q_demods.insert(g->q_demods.end(), cr.demods.begin(), cr.demods.end());
            g->q_packed.insert(g->q_packed.end(), cr.packed.begin(), cr.packed.end());
            g->q_decoded.insert(g->q_decoded.end(), cr.decoded.begin(), cr.decoded.end());
            g->q_ida.insert(g->q_ida.end(), cr.ida.begin(), cr.ida.end());
            g->store.erase(it);
        }
        g->next_emit++;
    }
}

int collect(irdm_group *g)
{
    for (int r = 0; r < g->n; r++)
        if (drain_member(g, r) != 0) return -1;
    emit_ready(g);
    return 0;
}

// the chunks of a feed: lengths, or -1 if the call breaks the rules of the stream
int cut(const irdm_group *g, size_t n_samples, std::vector<size_t> &lens)
{
    lens.clear();
    if (g->closed && n_samples) {
        fprintf(stderr, "irdm_hip group: stream already ended by a chunk shorter than max_chunk_samples\n");
        return -1;
    }
    if (n_samples > (size_t)g->n * g->chunk) {
        fprintf(stderr, "irdm_hip group: %zu samples exceed one super-step (%d x %zu)\n", n_samples, g->n, g->chunk);
        return -1;
    }
    for (size_t off = 0; off < n_samples; off += g->chunk) lens.push_back(std::min(g->chunk, n_samples - off));
    return 0;
}

static int stage_chunks(irdm_group *g, const void *src, size_t n_samples, bool host);

// (a staging step that fails after some of its chunks were counted leaves the per-member counters ahead of g->staged: the
// group is latched broken instead of mis-numbering the chunks of the next feed)
int stage(irdm_group *g, const void *src, size_t n_samples, bool host)
{
    if (g->broken) {
        fprintf(stderr, "irdm_hip group: an earlier transfer failed; the group cannot go on\n");
        return -1;
    }
    const uint64_t before = g->next_stage;
    const int rc = stage_chunks(g, src, n_samples, host);
    if (rc != 0 && g->next_stage != before) g->broken = true;
    return rc;
}

static int stage_chunks(irdm_group *g, const void *src, size_t n_samples, bool host)
{
    if (g->staged.size() >= 2) {
        fprintf(stderr, "irdm_hip group: two super-steps are already staged; feed one first\n");
        return -1;
    }
    std::vector<size_t> lens;
    if (cut(g, n_samples, lens) != 0) return -1;
    if (lens.empty()) return 0;
    // (two staged super-steps at most, so a member's two landing buffers are the chunk it works on -- or will work on next
    // -- and the one behind it; a landing buffer is written again two chunks of its member later, behind everything its
    // transfer stream was asked to read from it)
    const char *base = static_cast<const char *>(src);
    size_t off = 0;
    for (size_t i = 0; i < lens.size(); i++) {
        const uint64_t gno = g->next_stage;
        const int r = (int)(gno % (uint64_t)g->n);
        Member &mb = g->m[r];
        if (mb.staged - mb.fed >= 2) {
            fprintf(stderr, "irdm_hip group: member %d already holds two staged chunks\n", r);
            return -1;
        }
        char *land = static_cast<char *>(mb.land[mb.staged & 1]);
        char *dst = land + g->ov * g->bps;
        const size_t bytes = lens[i] * g->bps;
        if (!full_protocol(g)) {
            // one member, no hand-off: the stream is continuous in its context, so the chunk goes straight to its place in
            // the history ring (irdm_ingest_ptr's arithmetic for a chunk that is up to two ahead of the one being fed: the
            // ring is sized for exactly that) and irdm_feed_begin finds it there -- no landing buffer, no ring copy
            uint64_t ring_len = 0;
            char *ring = static_cast<char *>(irdm_ring_ptr(mb.p, &ring_len));
            const uint64_t pos = ring_len ? g->abs_staged % ring_len : 0;
            if (ring && pos + lens[i] <= ring_len) dst = ring + pos * g->bps;
        }
        mb.chunk_at[mb.staged & 1] = dst;
        GRP_HIP(hipSetDevice(mb.dev));
        // the buffer held the member's chunk before last: K1 and the ring copy that read it are long over, but nothing
        // orders this stream behind them (returns at once: that chunk's scan has settled since)
        if (mb.staged >= 2 && irdm_wait_ingest(mb.p) != 0) return -1;
        if (host) {
            GRP_HIP(hipMemcpyAsync(dst, base + off * g->bps, bytes, hipMemcpyHostToDevice, mb.xs));
        } else if (transfer(g, g->c_iq, 0, base + off * g->bps, g->m[0].xs, r, dst, mb.xs, bytes) != 0) {
            return -1;
        }
        g->st_scatter_bytes += bytes;
        // the overlap: the last ov samples in front of this chunk = the tail of the previous chunk's landing buffer (the
        // previous member's; chunks are at least ov long, only the stream's last chunk may be shorter and it has no successor)
        if (gno > 0 && full_protocol(g)) {
            const int rp = (int)((gno - 1) % (uint64_t)g->n);
            Member &pm = g->m[rp];
            const char *pland = static_cast<const char *>(pm.land[(pm.staged - 1) & 1]);
            // [ov + prev_len - ov, ov + prev_len) of the previous landing buffer, in samples
            if (transfer(g, g->c_iq, rp, pland + g->prev_len * g->bps, pm.xs, r, land, mb.xs, g->ov * g->bps) != 0) return -1;
            g->st_overlap_bytes += g->ov * g->bps;
        }
        GRP_HIP(hipSetDevice(mb.dev));
        GRP_HIP(hipEventRecord(mb.ev_land[mb.staged & 1], mb.xs));
        mb.staged++;
        g->next_stage++;
        g->abs_staged += lens[i];
        g->prev_len = lens[i];
        off += lens[i];
        if (lens[i] < g->chunk) g->closed = true;
    }
    Staged st;
    st.src = src;
    st.n = n_samples;
    st.host = host;
    st.lens = lens;
    g->staged.push_back(st);
    return 0;
}

int run_super_step(irdm_group *g)
{
    const Staged st = g->staged.front();
    const int k = (int)st.lens.size();
    const uint64_t g0 = g->next_chunk;
    const bool proto = full_protocol(g);
    // ---- K1 of every chunk ----
    uint64_t abs = g->abs_fed;
    for (int i = 0; i < k; i++) {
        const uint64_t gno = g0 + (uint64_t)i;
        const int r = (int)(gno % (uint64_t)g->n);
        Member &mb = g->m[r];
        const size_t len = st.lens[(size_t)i];
        char *land = static_cast<char *>(mb.land[mb.fed & 1]);
        GRP_HIP(hipSetDevice(mb.dev));
        GRP_HIP(hipEventSynchronize(mb.ev_land[mb.fed & 1]));   // the slice and its overlap are in the landing buffer
        if (proto && gno > 0 && irdm_seed_history_device(mb.p, land, g->ov, abs) != 0) {
            fprintf(stderr, "irdm_hip group: irdm_seed_history_device failed (member %d, chunk %llu)\n", r, (unsigned long long)gno);
            return -1;
        }
        if (irdm_feed_begin(mb.p, mb.chunk_at[mb.fed & 1], len, nullptr) != 0) return -1;
        abs += len;
    }
    // ---- the detector's chain, member by member ----
    for (int i = 0; i < k; i++) {
        const uint64_t gno = g0 + (uint64_t)i;
        const int r = (int)(gno % (uint64_t)g->n);
        Member &mb = g->m[r];
        char *blob_in = static_cast<char *>(mb.blob_in);
        bool late = false;
        GRP_HIP(hipSetDevice(mb.dev));
        if (proto && gno > 0) {
            if (!mb.state_posted) {
                fprintf(stderr, "irdm_hip group: no detector state on its way to member %d\n", r);
                return -1;
            }
            GRP_HIP(hipEventSynchronize(mb.ev_head));
            if (irdm_import_state_head_device(mb.p, blob_in, g->head_bytes) != 0) return -1;
#ifndef IRDM_RCCL_EMULATED
            // (the emulated device runs a launch to its end when it is enqueued: a scan cannot wait there)
            late = irdm_expect_history(mb.p, blob_in + g->head_bytes) == 1;
#endif
            if (!late) {
                GRP_HIP(hipEventSynchronize(mb.ev_hist));
                if (irdm_import_state_history_device(mb.p, blob_in + g->head_bytes, g->state_bytes - g->head_bytes) != 0) return -1;
            }
        }
        if (irdm_feed_end(mb.p) < 0) return -1;
        if (late) {
            GRP_HIP(hipEventSynchronize(mb.ev_hist));
            if (irdm_import_state_history_device(mb.p, blob_in + g->head_bytes, g->state_bytes - g->head_bytes) != 0) return -1;
            g->st_late++;
        }
        mb.state_posted = false;
        mb.fed++;
        if (proto) {
            // the scan settles; head first (the next member's scan can start), the history behind it
            if (irdm_export_state_device(mb.p, mb.blob_out, g->state_bytes) < 0) return -1;
            const int nx = (int)((gno + 1) % (uint64_t)g->n);
            Member &nm = g->m[nx];
            const char *out = static_cast<const char *>(mb.blob_out);
            char *in = static_cast<char *>(nm.blob_in);
            if (transfer(g, g->c_state, r, out, mb.hs, nx, in, nm.hs, g->head_bytes) != 0) return -1;
            GRP_HIP(hipSetDevice(nm.dev));
            GRP_HIP(hipEventRecord(nm.ev_head, nm.hs));
            if (transfer(g, g->c_state, r, out + g->head_bytes, mb.hs, nx, in + g->head_bytes, nm.hs,
                         g->state_bytes - g->head_bytes) != 0)
                return -1;
            GRP_HIP(hipSetDevice(nm.dev));
            GRP_HIP(hipEventRecord(nm.ev_hist, nm.hs));
            nm.state_posted = true;
            g->st_hops++;
            g->st_hop_bytes += g->state_bytes;
            // blob_out is read by the send until the receiver has it: the next export of this member comes a super-step
            // later, behind this hop's receiver having imported (its own export waits for its scan, which waited for it)
            GRP_HIP(hipSetDevice(mb.dev));
        }
        if (irdm_advance(mb.p) < 0) return -1;
        g->next_chunk++;
        if (collect(g) != 0) return -1;
    }
    g->abs_fed = abs;
    g->staged.pop_front();
    return k;
}

int feed(irdm_group *g, const void *src, size_t n_samples, bool host)
{
    if (!g || (!src && n_samples) || g->broken) return -1;
    if (!g->staged.empty()) {
        const Staged &st = g->staged.front();
        if (st.src != src || st.n != n_samples || st.host != host) {
            fprintf(stderr, "irdm_hip group: the super-step staged first is another buffer than the one fed\n");
            return -1;
        }
    } else if (stage(g, src, n_samples, host) != 0) {
        return -1;
    }
    if (g->staged.empty()) return 0;
    return run_super_step(g);
}

template <typename T>
int drain(std::deque<T> &q, T *out, int max)
{
    int n = 0;
    while (n < max && !q.empty()) {
        out[n++] = q.front();
        q.pop_front();
    }
    return n;
}

} // namespace

extern "C" void irdm_group_destroy(irdm_group_t *g)
{
    if (!g) return;
    for (Member &mb : g->m) {
        (void)hipSetDevice(mb.dev);
        if (mb.p) irdm_flush(mb.p);
        if (mb.xs) (void)hipStreamSynchronize(mb.xs);
        if (mb.hs) (void)hipStreamSynchronize(mb.hs);
    }
    if (g->comms) {
        Rccl *r = rccl();
        for (ncclComm_t c : g->c_iq) if (c) r->CommDestroy(c);
        for (ncclComm_t c : g->c_state) if (c) r->CommDestroy(c);
    }
    for (Member &mb : g->m) {
        (void)hipSetDevice(mb.dev);
        if (mb.p) irdm_destroy(mb.p);
        for (void *l : mb.land) if (l) (void)hipFree(l);
        if (mb.blob_out) (void)hipFree(mb.blob_out);
        if (mb.blob_in) (void)hipFree(mb.blob_in);
        if (mb.ev_head) (void)hipEventDestroy(mb.ev_head);
        if (mb.ev_hist) (void)hipEventDestroy(mb.ev_hist);
        for (hipEvent_t e : mb.ev_land) if (e) (void)hipEventDestroy(e);
        if (mb.xs) (void)hipStreamDestroy(mb.xs);
        if (mb.hs) (void)hipStreamDestroy(mb.hs);
    }
    delete g;
}

static int group_build(irdm_group *g, const irdm_config_t *cfg, const int *devices)
{
    int n_dev = 0;
    GRP_HIP(hipGetDeviceCount(&n_dev));
    for (int r = 0; r < g->n; r++) {
        Member &mb = g->m[r];
        mb.dev = devices ? devices[r] : r;
#ifndef IRDM_RCCL_EMULATED
        if (mb.dev < 0 || mb.dev >= n_dev) {
            fprintf(stderr, "irdm_hip group: device %d of %d asked for, %d present\n", mb.dev, g->n, n_dev);
            return -1;
        }
#endif
        irdm_config_t c = *cfg;
        c.device = mb.dev;
        if (c.pipeline_depth < 1) c.pipeline_depth = 1;
        // (every member must stamp the same wall-clock origin: the header of the state blob carries member 0's onwards,
        // but records of chunk 0 are built before any hop)
        if (r > 0) c.start_time_ns = irdm_start_time_ns(g->m[0].p);
        mb.p = irdm_create(&c);
        if (!mb.p) return -1;
        if (irdm_set_option(mb.p, "chunk_marks", 1) != 0) return -1;
        GRP_HIP(hipSetDevice(mb.dev));
        GRP_HIP(hipStreamCreateWithFlags(&mb.xs, hipStreamNonBlocking));
        GRP_HIP(hipStreamCreateWithFlags(&mb.hs, hipStreamNonBlocking));
        GRP_HIP(hipEventCreateWithFlags(&mb.ev_head, hipEventDisableTiming));
        GRP_HIP(hipEventCreateWithFlags(&mb.ev_hist, hipEventDisableTiming));
        for (int b = 0; b < 2; b++) GRP_HIP(hipEventCreateWithFlags(&mb.ev_land[b], hipEventDisableTiming));
    }
    irdm_pipeline_t *p0 = g->m[0].p;
    g->chunk = irdm_max_chunk_samples(p0);
    g->bps = irdm_bytes_per_sample(p0);
    g->ov = (irdm_required_overlap(p0) + 15) / 16 * 16;
    g->state_bytes = irdm_state_bytes(p0);
    g->head_bytes = irdm_state_head_bytes(p0);
    if (g->n > 1 && g->chunk < g->ov) {
        fprintf(stderr, "irdm_hip group: max_chunk_samples %zu is smaller than the chunk overlap %zu\n", g->chunk, g->ov);
        return -1;
    }
    for (int r = 0; r < g->n; r++) {
        Member &mb = g->m[r];
        GRP_HIP(hipSetDevice(mb.dev));
        for (int b = 0; b < 2; b++) GRP_HIP(hipMalloc(&mb.land[b], (g->ov + g->chunk) * g->bps));
        GRP_HIP(hipMalloc(&mb.blob_out, g->state_bytes));
        GRP_HIP(hipMalloc(&mb.blob_in, g->state_bytes));
    }
    return 0;
}

static int group_comms(irdm_group *g)
{
    if (g->comms) return 0;
    Rccl *r = rccl();
    if (!r) return -1;
    std::vector<int> devs;
    for (Member &mb : g->m) devs.push_back(mb.dev);
    g->c_iq.assign((size_t)g->n, nullptr);
    g->c_state.assign((size_t)g->n, nullptr);
    // librccl prints a version banner on STDOUT when its first communicator is made -- the stream a host like
    // iridium-sniffer-hip prints its RAW lines to.  While the communicators are made, file descriptor 1 is the process's
    // stderr: whatever the library writes or leaves in stdout's buffer goes there.
    // (documented at irdm_group_create; IRDM_GROUP_KEEP_STDOUT=1 leaves the descriptor alone)
    fflush(stdout);
    const char *ks = getenv("IRDM_GROUP_KEEP_STDOUT");
    const int keep = (ks && ks[0] == '1') ? -1 : dup(1);
    if (keep >= 0) dup2(2, 1);
    const ncclResult_t e1 = r->CommInitAll(g->c_iq.data(), g->n, devs.data());
    const ncclResult_t e2 = e1 == ncclSuccess ? r->CommInitAll(g->c_state.data(), g->n, devs.data()) : e1;
    fflush(stdout);
    if (keep >= 0) {
        dup2(keep, 1);
        close(keep);
    }
    GRP_NCCL(e1);
    GRP_NCCL(e2);
    g->comms = true;
    return 0;
}

extern "C" int irdm_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" irdm_group_t *irdm_group_create(const irdm_config_t *cfg, int n_gpus, const int *devices)
{
    if (!cfg || n_gpus < 1 || n_gpus > 64) return nullptr;
#ifndef IRDM_RCCL_EMULATED
    if (n_gpus > 1) {
        // (honesty clause: groups of two and more members have run on the CPU emulation of the devices and of RCCL and as two
        // ranks sharing one GPU, never yet on two real GPUs -- tests/test_gpu_group.py skips there)
        static bool warned = false;
        if (!warned) fprintf(stderr, "irdm_hip group: %d members -- EXPERIMENTAL: the multi-GPU group protocol has not run on multi-GPU hardware yet\n", n_gpus);
        warned = true;
    }
#endif
    irdm_group *g = new (std::nothrow) irdm_group();
    if (!g) return nullptr;
    g->n = n_gpus;
    g->m.resize((size_t)n_gpus);
    if (group_build(g, cfg, devices) != 0 || (n_gpus > 1 && group_comms(g) != 0)) {
        irdm_group_destroy(g);
        return nullptr;
    }
    return g;
}

extern "C" int irdm_group_size(const irdm_group_t *g) { return g ? g->n : -1; }

extern "C" irdm_pipeline_t *irdm_group_member(irdm_group_t *g, int i) { return g && i >= 0 && i < g->n ? g->m[(size_t)i].p : nullptr; }

extern "C" int irdm_group_set_option(irdm_group_t *g, const char *key, int value)
{
    if (!g || !key) return -1;
    if (!strcmp(key, "group_loopback")) {
        if (g->next_stage) return -1;            // before the stream starts
        if (value && g->chunk < g->ov) {
            fprintf(stderr, "irdm_hip group: max_chunk_samples %zu is smaller than the chunk overlap %zu\n", g->chunk, g->ov);
            return -1;
        }
        if (value && group_comms(g) != 0) return -1;
        g->loopback = value ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "chunk_marks") || !strcmp(key, "pipeline_depth")) return -1;
    int rc = 0;
    for (Member &mb : g->m) rc |= irdm_set_option(mb.p, key, value);
    if (rc == 0 && !strcmp(key, "keep_frame_samples")) g->keep_samples = value;
    return rc ? -1 : 0;
}

extern "C" int64_t irdm_group_get_stat(const irdm_group_t *g, const char *key)
{
    if (!g || !key) return -1;
    if (!strcmp(key, "hops")) return (int64_t)g->st_hops;
    if (!strcmp(key, "hop_bytes")) return (int64_t)g->st_hop_bytes;
    if (!strcmp(key, "scatter_bytes")) return (int64_t)g->st_scatter_bytes;
    if (!strcmp(key, "overlap_bytes")) return (int64_t)g->st_overlap_bytes;
    if (!strcmp(key, "late_history")) return (int64_t)g->st_late;
    if (!strcmp(key, "chunks")) return (int64_t)g->next_chunk;
    if (!strcmp(key, "overlap_samples")) return (int64_t)g->ov;
    if (!strcmp(key, "tagged")) {
        // the count travels with the detector state: the member that scanned the last chunk holds the stream's
        if (!full_protocol(g) || g->next_chunk == 0) return (int64_t)irdm_tagged_bursts(g->m[0].p);
        return (int64_t)irdm_tagged_bursts(g->m[(size_t)((g->next_chunk - 1) % (uint64_t)g->n)].p);
    }
    int64_t sum = 0;
    for (const Member &mb : g->m) {
        const int64_t v = irdm_get_stat(mb.p, key);
        if (v < 0) return v;
        sum += v;
    }
    return sum;
}

extern "C" int irdm_group_stage_host(irdm_group_t *g, const void *h_iq, size_t n_samples)
{
    if (!g || (!h_iq && n_samples)) return -1;
    return stage(g, h_iq, n_samples, true);
}

extern "C" int irdm_group_stage_device(irdm_group_t *g, const void *d_iq, size_t n_samples)
{
    if (!g || (!d_iq && n_samples)) return -1;
    return stage(g, d_iq, n_samples, false);
}

extern "C" int irdm_group_feed_host(irdm_group_t *g, const void *h_iq, size_t n_samples) { return feed(g, h_iq, n_samples, true); }

extern "C" int irdm_group_feed_device(irdm_group_t *g, const void *d_iq, size_t n_samples) { return feed(g, d_iq, n_samples, false); }

extern "C" int irdm_group_flush(irdm_group_t *g)
{
    if (!g || g->broken) return -1;
    if (!g->staged.empty()) {
        fprintf(stderr, "irdm_hip group: a staged super-step was never fed\n");
        return -1;
    }
    int rc = 0;
    for (Member &mb : g->m) {
        (void)hipSetDevice(mb.dev);
        if (irdm_flush(mb.p) < 0) rc = -1;
    }
    if (rc == 0) rc = collect(g);
    return rc;
}

extern "C" int irdm_group_poll_bursts(irdm_group_t *g, irdm_burst_t *out, int max)
{
    if (!g || !out || max < 0) return -1;
    return drain(g->q_bursts, out, max);
}

extern "C" int irdm_group_poll_frames(irdm_group_t *g, irdm_frame_info_t *out, float *samples_out, int max)
{
    if (!g || !out || max < 0) return -1;
    int n = 0;
    while (n < max && !g->q_frames.empty()) {
        out[n] = g->q_frames.front();
        g->q_frames.pop_front();
        std::vector<float> &sv = g->q_frame_samples.front();
        if (samples_out) {
            float *dst = samples_out + (size_t)n * 2 * IRDM_MAX_FRAME_SAMPLES;
            memset(dst, 0, sizeof(float) * 2 * IRDM_MAX_FRAME_SAMPLES);
            if (!sv.empty()) memcpy(dst, sv.data(), sizeof(float) * sv.size());
        }
        g->q_frame_samples.pop_front();
        n++;
    }
    return n;
}

extern "C" int irdm_group_poll_demods(irdm_group_t *g, irdm_demod_t *out, int max)
{
    if (!g || !out || max < 0) return -1;
    return drain(g->q_demods, out, max);
}

extern "C" int irdm_group_poll_demods_packed(irdm_group_t *g, irdm_demod_packed_t *out, int max)
{
    if (!g || !out || max < 0) return -1;
    return drain(g->q_packed, out, max);
}

extern "C" int irdm_group_poll_decoded(irdm_group_t *g, irdm_decoded_t *out, int max)
{
    if (!g || !out || max < 0) return -1;
    return drain(g->q_decoded, out, max);
}

extern "C" int irdm_group_poll_ida(irdm_group_t *g, irdm_ida_t *out, int max)
{
    if (!g || !out || max < 0) return -1;
    return drain(g->q_ida, out, max);
}
