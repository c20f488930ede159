// wave_emul.hpp -- TEST INFRASTRUCTURE: a 64-lane wavefront on the CPU, so that csrc/band_wave.hpp (the walk pass as the
// gfx950 kernel compiles it: a wavefront per band and segment, lane = burst slot, ballots / readlanes / DPP shifts) can be
// run against the oracle without a GPU (tests/band_host.cpp, tests/test_band_host.py).
//
// The 64 lanes are 64 user-space contexts (ucontext) that run the same code; every cross-lane operation deposits the
// lane's value in a box, waits until all lanes have arrived (the lanes are resumed round robin), reads what it needs and
// waits again before the box is reused.  Control flow must be uniform at the cross-lane operations -- which it is in
// band_wave.hpp by construction (ballot results, readlane results and wave-uniform loads decide every branch around them).
//
// Included BEFORE band_wave.hpp: it maps the device keywords and builtins that header uses onto this emulation.
#pragma once
#include <stdint.h>
#include <string.h>
#include <ucontext.h>
#include <functional>
#include <vector>

namespace wave_emul {

constexpr int kLanes = 64;

struct Team {
    ucontext_t main_ctx, ctx[kLanes];
    std::vector<char> stack[kLanes];
    bool done[kLanes];
    int cur = 0, count = 0, live = 0;
    unsigned gen = 0;
    uint64_t box[kLanes];
    std::function<void(int)> body;
};

inline Team &team()
{
    static Team t;
    return t;
}

inline int lane() { return team().cur; }

inline void yield()
{
    Team &t = team();
    const int me = t.cur;
    int nxt = me;
    do nxt = (nxt + 1) % kLanes;
    while (t.done[nxt] && nxt != me);
    if (nxt == me) return;
    t.cur = nxt;
    swapcontext(&t.ctx[me], &t.ctx[nxt]);
}

inline void barrier()
{
    Team &t = team();
    const unsigned g = t.gen;
    if (++t.count == t.live) {
        t.count = 0;
        t.gen++;
    }
    while (t.gen == g) yield();
}

inline void lane_entry(int l)
{
    Team &t = team();
    t.body(l);
    t.done[l] = true;
    t.live--;
    // (every lane runs the same cross-lane operations, so they all end in the same round)
    for (int k = 1; k <= kLanes; k++) {
        const int nxt = (l + k) % kLanes;
        if (!t.done[nxt]) {
            t.cur = nxt;
            setcontext(&t.ctx[nxt]);
        }
    }
    setcontext(&t.main_ctx);
}

// run body(lane) on all 64 lanes in lock step
inline void run(const std::function<void(int)> &body)
{
    Team &t = team();
    t.body = body;
    t.count = 0;
    t.live = kLanes;
    for (int l = 0; l < kLanes; l++) {
        t.done[l] = false;
        if (t.stack[l].empty()) t.stack[l].resize(256 * 1024);
        getcontext(&t.ctx[l]);
        t.ctx[l].uc_stack.ss_sp = t.stack[l].data();
        t.ctx[l].uc_stack.ss_size = t.stack[l].size();
        t.ctx[l].uc_link = nullptr;
        makecontext(&t.ctx[l], reinterpret_cast<void (*)()>(lane_entry), 1, l);
    }
    t.cur = 0;
    swapcontext(&t.main_ctx, &t.ctx[0]);
}

inline uint64_t exchange(uint64_t v, int from_lane)          // every lane gives v, gets lane from_lane's
{
    Team &t = team();
    t.box[lane()] = v;
    barrier();
    const uint64_t r = t.box[from_lane & (kLanes - 1)];
    barrier();
    return r;
}

inline uint64_t ballot(bool p)
{
    Team &t = team();
    t.box[lane()] = p ? 1 : 0;
    barrier();
    uint64_t m = 0;
    for (int i = 0; i < kLanes; i++) m |= (t.box[i] & 1) << i;
    barrier();
    return m;
}

inline int readlane(int v, int l) { return (int)(uint32_t)exchange((uint32_t)v, l); }
inline int readfirstlane(int v) { return readlane(v, 0); }      // (all lanes are active wherever band_wave.hpp uses it)

template <typename X>
inline X shfl_xor(X v, int d)
{
    static_assert(sizeof(X) == 4, "32-bit values");
    uint32_t u;
    memcpy(&u, &v, 4);
    u = (uint32_t)exchange(u, lane() ^ d);
    X r;
    memcpy(&r, &u, 4);
    return r;
}

// v_mov_b32_dpp row_shr:n (ctrl 0x110 + n), bound_ctrl off: lanes without a source inside their row of 16 keep `old`
inline int update_dpp(int old, int src, int ctrl, int, int, bool)
{
    Team &t = team();
    const int n = ctrl - 0x110, l = lane();
    t.box[l] = (uint32_t)src;
    barrier();
    const int r = (l & 15) >= n ? (int)(uint32_t)t.box[l - n] : old;
    barrier();
    return r;
}

}  // namespace wave_emul

#define __device__
#define __forceinline__ inline
#define __builtin_amdgcn_readfirstlane(v) wave_emul::readfirstlane((v))
#define __builtin_amdgcn_readlane(v, l) wave_emul::readlane((v), (l))
#define __builtin_amdgcn_ballot_w64(p) wave_emul::ballot((p))
#define __builtin_amdgcn_update_dpp(o, s, c, rm, bm, bc) wave_emul::update_dpp((o), (s), (c), (rm), (bm), (bc))
#define __shfl_xor(v, d) wave_emul::shfl_xor((v), (d))

inline float __uint_as_float(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline uint32_t __float_as_uint(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
namespace irdm {
inline int min(int a, int b) { return a < b ? a : b; }
}
