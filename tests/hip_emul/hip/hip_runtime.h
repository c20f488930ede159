// hip/hip_runtime.h -- TEST INFRASTRUCTURE, not HIP: just enough of the HIP device language, emulated on the CPU, to run
// iridium-sniffer_amd/csrc/scan_band.hip (the band scan's kernels exactly as the gfx950 build compiles them) inside a host
// test without a GPU (tests/scan_emul.cpp, tests/test_kernels_emul.py).
//
// A workgroup is a set of user-space contexts (ucontext), one per thread, resumed round robin.  __syncthreads() and the
// wavefront-level operations (ballot, readlane, shuffles, DPP row shifts, wave barrier) are rendezvous points: a thread
// deposits its value, waits until every live thread of the workgroup / wavefront has arrived, reads what it needs and
// waits again before the slot is reused.  Threads that have returned no longer count (as exited wavefronts do not on
// the hardware).  Workgroups of a launch run one after the other; a launch returns when its last workgroup is done, so
// streams, events and memory fences are no-ops and atomics are plain operations.  __shared__ variables are statics
// (one workgroup at a time); dynamic LDS is one 160 KB buffer (hip_emul::dyn_lds()).
#pragma once
#define IRDM_HIP_EMULATED 1   /* the product compiled against this header: the CPU emulation (tests/emul_build.py) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <algorithm>
#include <functional>
#include <vector>

#define __HIPCC__ 1
#define __HIP_DEVICE_COMPILE__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_SYSTEM 1

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct short2 { short x, y; };
struct char2 { signed char x, y; };
inline float2 make_float2(float x, float y) { return float2{ x, y }; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{ x, y, z, w }; }
inline int2 make_int2(int x, int y) { return int2{ x, y }; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{ x, y, z, w }; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorUnknown = 1 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
template <typename... A> inline hipError_t hipFuncSetAttribute(A...) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
using std::max;
using std::min;

namespace hip_emul {

struct Idx3 { unsigned x, y, z; };

// Switching between the threads of a workgroup: on x86-64 a dozen instructions (callee-saved registers and the stack
// pointer; ucontext's swapcontext makes two signal-mask system calls per switch, which made the emulated tests ten times
// slower), elsewhere ucontext.
#if defined(__x86_64__) && !defined(HIP_EMUL_UCONTEXT)
#define HIP_EMUL_FAST_SWITCH 1
static __attribute__((naked, noinline)) void fiber_switch(void **save_sp, void *load_sp)
{
    asm volatile("pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
                 "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
                 "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret");
}
#endif

struct Machine {
    ucontext_t main_ctx;
    std::vector<ucontext_t> ctx;
    std::vector<void *> sp;
    void *main_sp = nullptr;
    std::vector<std::vector<char>> stacks;
    std::vector<char> done;
    std::vector<uint64_t> box;
    int n = 0, cur = 0, live = 0;
    int wg_count = 0;
    unsigned wg_gen = 0;
    std::vector<int> wave_live, wave_count;
    std::vector<unsigned> wave_gen;
    Idx3 block{ 0, 0, 0 }, bdim{ 1, 1, 1 }, gdim{ 1, 1, 1 };
    std::function<void()> body;
    unsigned long long clock = 0;
    std::vector<unsigned char> lds;
};

inline Machine &M()
{
    static Machine m;
    return m;
}

inline unsigned char *dyn_lds()
{
    Machine &m = M();
    if (m.lds.empty()) m.lds.resize(160 * 1024 + 64);
    return reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(m.lds.data()) + 63) & ~(uintptr_t)63);
}

inline Idx3 thread_idx() { return Idx3{ (unsigned)M().cur, 0, 0 }; }
inline Idx3 block_idx() { return M().block; }
inline Idx3 block_dim() { return M().bdim; }
inline Idx3 grid_dim() { return M().gdim; }

inline void yield()
{
    Machine &m = M();
    const int me = m.cur;
    int nxt = me;
    do nxt = nxt + 1 == m.n ? 0 : nxt + 1;
    while (m.done[nxt] && nxt != me);
    if (nxt == me) return;
    m.cur = nxt;
#if HIP_EMUL_FAST_SWITCH
    fiber_switch(&m.sp[me], m.sp[nxt]);
#else
    swapcontext(&m.ctx[me], &m.ctx[nxt]);
#endif
}

inline void wg_barrier()
{
    Machine &m = M();
    const unsigned g = m.wg_gen;
    if (++m.wg_count >= m.live) {
        m.wg_count = 0;
        m.wg_gen++;
    }
    while (m.wg_gen == g) yield();
}

inline void wave_barrier()
{
    Machine &m = M();
    const int w = m.cur >> 6;
    const unsigned g = m.wave_gen[w];
    if (++m.wave_count[w] >= m.wave_live[w]) {
        m.wave_count[w] = 0;
        m.wave_gen[w]++;
    }
    while (m.wave_gen[w] == g) yield();
}

inline void thread_exit()
{
    Machine &m = M();
    const int me = m.cur, w = me >> 6;
    m.done[me] = 1;
    m.live--;
    m.wave_live[w]--;
    // threads waiting at a rendezvous that only this one had not reached may go on
    if (m.live > 0 && m.wg_count >= m.live) {
        m.wg_count = 0;
        m.wg_gen++;
    }
    if (m.wave_live[w] > 0 && m.wave_count[w] >= m.wave_live[w]) {
        m.wave_count[w] = 0;
        m.wave_gen[w]++;
    }
    for (int k = 1; k <= m.n; k++) {
        const int nxt = (me + k) % m.n;
        if (!m.done[nxt]) {
            m.cur = nxt;
#if HIP_EMUL_FAST_SWITCH
            void *dead;
            fiber_switch(&dead, m.sp[nxt]);
#else
            setcontext(&m.ctx[nxt]);
#endif
        }
    }
#if HIP_EMUL_FAST_SWITCH
    void *dead;
    fiber_switch(&dead, m.main_sp);
#else
    setcontext(&m.main_ctx);
#endif
}

inline void thread_entry()
{
    M().body();
    thread_exit();
}

inline void run_workgroup(int n_threads, const std::function<void()> &body)
{
    Machine &m = M();
    m.n = n_threads;
    m.live = n_threads;
    m.wg_count = 0;
    m.body = body;
    if ((int)m.stacks.size() < n_threads) {
        m.ctx.resize(n_threads);
        m.sp.resize(n_threads);
        m.stacks.resize(n_threads);
    }
    m.done.assign(n_threads, 0);
    m.box.assign(n_threads, 0);
    const int n_waves = (n_threads + 63) / 64;
    m.wave_live.assign(n_waves, 0);
    m.wave_count.assign(n_waves, 0);
    m.wave_gen.assign(n_waves, 0);
    for (int t = 0; t < n_threads; t++) {
        m.wave_live[t >> 6]++;
        if (m.stacks[t].empty()) m.stacks[t].resize(192 * 1024);
#if HIP_EMUL_FAST_SWITCH
        // a fresh stack as fiber_switch expects one: six register slots, then the address it returns to; the slot above
        // keeps the stack pointer 8 modulo 16 at thread_entry's first instruction, as after a call
        uintptr_t top = (reinterpret_cast<uintptr_t>(m.stacks[t].data()) + m.stacks[t].size()) & ~(uintptr_t)15;
        void **q = reinterpret_cast<void **>(top);
        *--q = nullptr;
        *--q = reinterpret_cast<void *>(thread_entry);
        for (int r = 0; r < 6; r++) *--q = nullptr;
        m.sp[t] = q;
#else
        getcontext(&m.ctx[t]);
        m.ctx[t].uc_stack.ss_sp = m.stacks[t].data();
        m.ctx[t].uc_stack.ss_size = m.stacks[t].size();
        m.ctx[t].uc_link = nullptr;
        makecontext(&m.ctx[t], reinterpret_cast<void (*)()>(thread_entry), 0);
#endif
    }
    m.cur = 0;
#if HIP_EMUL_FAST_SWITCH
    fiber_switch(&m.main_sp, m.sp[0]);
#else
    swapcontext(&m.main_ctx, &m.ctx[0]);
#endif
}

template <typename K, typename... A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args)
{
    Machine &m = M();
    m.gdim = Idx3{ grid.x, grid.y, grid.z };
    m.bdim = Idx3{ block.x, block.y, block.z };
    // The hardware promises no order among the workgroups of a launch: HIP_EMUL_ORDER=reverse runs them last to first,
    // HIP_EMUL_ORDER=shuffle in a pseudo-random order that changes from launch to launch (results must not depend on it)
    static const char *order = getenv("HIP_EMUL_ORDER");
    static unsigned long long lcg = 88172645463325252ull;
    std::vector<unsigned> seq(grid.x);
    for (unsigned i = 0; i < grid.x; i++) seq[i] = i;
    if (order && order[0] == 'r') std::reverse(seq.begin(), seq.end());
    if (order && order[0] == 's')
        for (unsigned i = grid.x; i > 1; i--) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(seq[i - 1], seq[(unsigned)((lcg >> 33) % i)]);
        }
    for (unsigned ky = 0; ky < grid.y; ky++)
        for (unsigned k = 0; k < grid.x; k++) {
            m.block = Idx3{ seq[k], (order && order[0] == 'r') ? grid.y - 1 - ky : ky, 0 };
            run_workgroup((int)block.x, [&]() { kernel(args...); });
        }
}

// ---- wavefront-level exchanges (lanes of the calling thread's wavefront) ----
inline uint64_t wave_exchange(uint64_t v, int from_lane)
{
    Machine &m = M();
    const int base = m.cur & ~63;
    m.box[m.cur] = v;
    wave_barrier();
    const uint64_t r = m.box[base + (from_lane & 63)];
    wave_barrier();
    return r;
}

inline uint64_t ballot(bool p)
{
    Machine &m = M();
    const int base = m.cur & ~63;
    m.box[m.cur] = p ? 1 : 0;
    wave_barrier();
    uint64_t r = 0;
    for (int i = 0; i < 64 && base + i < m.n; i++)
        if (!m.done[base + i]) r |= (m.box[base + i] & 1) << i;
    wave_barrier();
    return r;
}

inline int first_live_lane()
{
    Machine &m = M();
    const int base = m.cur & ~63;
    for (int i = 0; i < 64 && base + i < m.n; i++)
        if (!m.done[base + i]) return i;
    return 0;
}

template <typename X>
inline X shfl(X v, int src_lane)
{
    static_assert(sizeof(X) == 4, "32-bit values");
    uint32_t u;
    memcpy(&u, &v, 4);
    u = (uint32_t)wave_exchange(u, src_lane);
    X r;
    memcpy(&r, &u, 4);
    return r;
}

inline int update_dpp(int old, int src, int ctrl, int, int, bool)
{
    Machine &m = M();
    const int l = m.cur & 63, base = m.cur & ~63;
    m.box[m.cur] = (uint32_t)src;
    wave_barrier();
    int r = old;
    if (ctrl >= 0x111 && ctrl <= 0x11f) {                 // row_shr:n
        const int n = ctrl - 0x110;
        if ((l & 15) >= n) r = (int)(uint32_t)m.box[base + l - n];
    } else if (ctrl == 0x138) {                           // wave_shr:1
        if (l >= 1) r = (int)(uint32_t)m.box[base + l - 1];
    } else {
        fprintf(stderr, "hip_emul: DPP control 0x%x is not emulated\n", ctrl);
        abort();
    }
    wave_barrier();
    return r;
}

struct BufferRsrc {
    char *base;
    uint32_t num;
};

}  // namespace hip_emul

#define threadIdx (hip_emul::thread_idx())
#define blockIdx (hip_emul::block_idx())
#define blockDim (hip_emul::block_dim())
#define gridDim (hip_emul::grid_dim())
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) hip_emul::launch(kernel, grid, block, lds, stream, __VA_ARGS__)

inline void __syncthreads() { hip_emul::wg_barrier(); }
inline void __threadfence() {}
inline void __threadfence_system() {}
inline unsigned long long wall_clock64() { return hip_emul::M().clock += 7; }

#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() hip_emul::wave_barrier()
#define __builtin_amdgcn_ballot_w64(p) hip_emul::ballot((p))
#define __builtin_amdgcn_readlane(v, l) hip_emul::shfl<int>((v), (l))
#define __builtin_amdgcn_readfirstlane(v) hip_emul::shfl<int>((v), hip_emul::first_live_lane())
#define __builtin_amdgcn_update_dpp(o, s, c, rm, bm, bc) hip_emul::update_dpp((o), (s), (c), (rm), (bm), (bc))
template <typename X> inline X __shfl_xor(X v, int d) { return hip_emul::shfl<X>(v, (hip_emul::M().cur & 63) ^ d); }
template <typename X> inline X __shfl_up(X v, int d)
{
    const int l = hip_emul::M().cur & 63;
    return hip_emul::shfl<X>(v, l >= d ? l - d : l);
}

template <typename T, typename V> inline T atomicAdd(T *p, V v) { const T o = *p; *p = (T)(o + (T)v); return o; }
template <typename T, typename V> inline T atomicOr(T *p, V v) { const T o = *p; *p = (T)(o | (T)v); return o; }
template <typename T, typename V> inline T atomicMin(T *p, V v) { const T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <typename T, typename V> inline T atomicMax(T *p, V v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
template <typename T, typename V> inline T hip_emul_fetch_add(T *p, V v) { const T o = *p; *p = (T)(o + (T)v); return o; }
#define __hip_atomic_fetch_add(p, v, order, scope) hip_emul_fetch_add((p), (v))

typedef hip_emul::BufferRsrc __amdgpu_buffer_rsrc_t;
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void *p, short, int num, int)
{
    return __amdgpu_buffer_rsrc_t{ static_cast<char *>(p), (uint32_t)num };
}
// raw buffer access: the range check is on the vector offset (the scalar offset is not part of it on gfx9)
inline int __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
    if ((uint32_t)voff + 4u > r.num) return 0;
    int v;
    memcpy(&v, r.base + (size_t)(uint32_t)voff + (size_t)(uint32_t)soff, 4);
    return v;
}
inline void __builtin_amdgcn_raw_buffer_store_b32(int v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
    if ((uint32_t)voff + 4u > r.num) return;
    memcpy(r.base + (size_t)(uint32_t)voff + (size_t)(uint32_t)soff, &v, 4);
}
namespace hip_emul {
struct u32x2 { uint32_t v[2]; };
struct u32x4 { uint32_t v[4]; };
}
inline hip_emul::u32x2 __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
    hip_emul::u32x2 v{};
    if ((uint32_t)voff + 8u > r.num) return v;
    memcpy(&v, r.base + (size_t)(uint32_t)voff + (size_t)(uint32_t)soff, 8);
    return v;
}
inline hip_emul::u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
    hip_emul::u32x4 v{};
    if ((uint32_t)voff + 16u > r.num) return v;
    memcpy(&v, r.base + (size_t)(uint32_t)voff + (size_t)(uint32_t)soff, 16);
    return v;
}
// (V: any 16-byte value -- the emulation's u32x4 or a GCC vector of four words)
template <typename V> inline void __builtin_amdgcn_raw_buffer_store_b128(V v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
    static_assert(sizeof(V) == 16, "a 128-bit store");
    if ((uint32_t)voff + 16u > r.num) return;
    memcpy(r.base + (size_t)(uint32_t)voff + (size_t)(uint32_t)soff, &v, 16);
}

inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __brev(unsigned v)
{
    unsigned r = 0;
    for (int b = 0; b < 32; b++)
        if (v & (1u << b)) r |= 1u << (31 - b);
    return r;
}

// (detect.hip)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) ((void)(*(p) = (v)))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
inline int __any(int p) { return hip_emul::ballot(p != 0) != 0; }
inline int __all(int p) { return hip_emul::ballot(p == 0) == 0; }
// (bitlayer.hip)
#define __constant__ static
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __ffs(int v) { return __builtin_ffs(v); }
// (downmix.hip, libm_port.hpp)
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{ x, y, z, w }; }
inline double __fma_rn(double a, double b, double c) { return __builtin_fma(a, b, c); }
inline double __dmul_rn(double a, double b) { volatile double p = a * b; return p; }
inline double __dadd_rn(double a, double b) { volatile double s = a + b; return s; }
inline unsigned long long __builtin_amdgcn_s_memtime_emul() { return hip_emul::M().clock += 3; }
#define __builtin_amdgcn_s_memtime() __builtin_amdgcn_s_memtime_emul()
#define HIP_SYMBOL(x) (&(x))
struct hipDeviceProp_t { int multiProcessorCount; };
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 1; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(0x20); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
template <typename S> inline hipError_t hipMemcpyFromSymbol(void *dst, S *sym, size_t n, size_t off = 0, int = 0)
{
    memcpy(dst, reinterpret_cast<const char *>(sym) + off, n);
    return hipSuccess;
}
template <typename S> inline hipError_t hipMemcpyToSymbol(S *sym, const void *src, size_t n, size_t off = 0, int = 0)
{
    memcpy(reinterpret_cast<char *>(sym) + off, src, n);
    return hipSuccess;
}
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
template <typename X> inline X __shfl_down(X v, int d)
{
    const int l = hip_emul::M().cur & 63;
    return hip_emul::shfl<X>(v, l + d < 64 ? l + d : l);
}
template <typename X> inline X __shfl(X v, int src) { return hip_emul::shfl<X>(v, src); }
// v_mfma_f32_16x16x4_f32 as the guide and tools/ubench/mfma_fir.hip say the instruction computes: D = A B + C with
// A[i][k] in lane 16 k + i, B[k][n] in lane 16 k + n, lane l register r holding row 4 (l >> 4) + r, column l & 15, and every
// element an fmaf chain over k = 0..3 in order (the ubench checks exactly this against the hardware, bit for bit).
template <typename V> inline V __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, V c, int, int, int)
{
    const int l = hip_emul::M().cur & 63, col = l & 15;
    float bb[4];
    for (int k = 0; k < 4; k++) bb[k] = hip_emul::shfl<float>(b, 16 * k + col);
    V d = c;
    for (int r = 0; r < 4; r++) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; k++) acc = fmaf(hip_emul::shfl<float>(a, 16 * k + row), bb[k], acc);
        d[r] = acc;
    }
    return d;
}

// ---- the host runtime API as far as csrc/pipeline.cpp and csrc/compat.cpp use it: device memory is host memory, streams
// and events are tokens, everything has completed when the call returns ----
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2,
       hipHostMallocCoherent = 0x40000000 };
inline hipError_t hipMalloc(void **p, size_t n)
{
    *p = nullptr;
    if (posix_memalign(p, 256, n ? n : 256) != 0) return hipErrorUnknown;
    return hipSuccess;
}
template <typename T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc(reinterpret_cast<void **>(p), n); }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <typename T> inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc(reinterpret_cast<void **>(p), n, f); }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = reinterpret_cast<hipStream_t>(0x10); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) { return hipStreamCreate(s); }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(0x20); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 1; *hi = -1; return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = (size_t)64 << 30; *tot = (size_t)64 << 30; return hipSuccess; }

namespace hip_emul {
// v_writelane_b32 v, s, 0 (fir_reg.hip: lane 0 of v <- the wavefront-uniform s)
inline float writelane0(float v, float s) { return (M().cur & 63) == 0 ? s : v; }
}
template <typename T, typename V> inline T atomicAnd(T *p, V v) { const T o = *p; *p = (T)(o & (T)v); return o; }
template <typename T, typename V> inline T atomicExch(T *p, V v) { const T o = *p; *p = (T)v; return o; }
template <typename T> inline T atomicCAS(T *p, T cmp, T v) { const T o = *p; if (o == cmp) *p = v; return o; }
inline unsigned long long __ballot(int p) { return hip_emul::ballot(p != 0); }
