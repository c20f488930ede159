// TEST INFRASTRUCTURE: the handful of RCCL entry points csrc/group.cpp uses, for the CPU emulation of the HIP device
// language (tests/hip_emul/hip/hip_runtime.h).  Every "device" of the emulation is host memory and every stream runs a
// call to its end when it is enqueued, so a point-to-point pair is a memcpy made when the ncclGroupEnd that closes the
// group finds both halves: a receive on communicator d from peer s takes the oldest unmatched send on communicator s to
// peer d.  A send or receive outside a group, or a group that ends with an unmatched half, is an error -- the product
// always groups the two halves of a transfer (one thread drives all members), and the emulation holds it to that.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>

#define IRDM_RCCL_EMULATED 1

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
struct ncclCommEmul { int rank, n; void *world; };
typedef ncclCommEmul *ncclComm_t;

namespace rccl_emul {
struct Op { bool send; int self, peer; void *buf; size_t bytes; };
struct State { int depth = 0; std::vector<Op> ops; uint64_t pairs = 0, bytes = 0; };
inline State &S() { static thread_local State s; return s; }
}

inline const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "emulated RCCL: invalid usage"; }

inline ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *)
{
    for (int i = 0; i < n; i++) comms[i] = new ncclCommEmul{ i, n, comms };
    return ncclSuccess;
}
inline ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
inline ncclResult_t ncclGroupStart() { rccl_emul::S().depth++; return ncclSuccess; }
inline ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t)
{
    if (rccl_emul::S().depth == 0 || peer < 0 || peer >= c->n) return ncclInvalidUsage;
    rccl_emul::S().ops.push_back({ true, c->rank, peer, const_cast<void *>(buf), count });
    return ncclSuccess;
}
inline ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t)
{
    if (rccl_emul::S().depth == 0 || peer < 0 || peer >= c->n) return ncclInvalidUsage;
    rccl_emul::S().ops.push_back({ false, c->rank, peer, buf, count });
    return ncclSuccess;
}
inline ncclResult_t ncclGroupEnd()
{
    rccl_emul::State &s = rccl_emul::S();
    if (s.depth == 0) return ncclInvalidUsage;
    if (--s.depth > 0) return ncclSuccess;
    std::vector<rccl_emul::Op> ops;
    ops.swap(s.ops);
    std::vector<char> used(ops.size(), 0);
    // (every send is read before any receive is written only where the buffers do not overlap: the product's do not)
    for (size_t i = 0; i < ops.size(); i++) {
        if (ops[i].send) continue;
        size_t j = 0;
        for (; j < ops.size(); j++)
            if (!used[j] && ops[j].send && ops[j].self == ops[i].peer && ops[j].peer == ops[i].self) break;
        if (j == ops.size() || ops[j].bytes != ops[i].bytes) return ncclInvalidUsage;
        used[j] = used[i] = 1;
        memmove(ops[i].buf, ops[j].buf, ops[i].bytes);
        s.pairs++;
        s.bytes += ops[i].bytes;
    }
    for (size_t i = 0; i < ops.size(); i++)
        if (!used[i]) return ncclInvalidUsage;
    return ncclSuccess;
}
