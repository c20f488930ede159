// bitlayer_emul.cpp -- TEST INFRASTRUCTURE: csrc/bitlayer.hip's frame_decode kernel (frame_decode.c:414-598: access-code
// check, de-interleave, BCH by syndrome tables, Chase decoding on the LLRs, IRA / IBC field extraction) on the CPU
// emulation of tests/hip_emul/hip/hip_runtime.h, with the syndrome tables built as csrc/create.cpp builds them
// (frame_decode.c:95-129), against the oracle's frame_decode (itself pinned to the reference's object code).
#include <hip/hip_runtime.h>
#include <vector>

#include "bitlayer_emul.inc"

using namespace irdm;

namespace {

unsigned poly_rem(unsigned poly, unsigned v)
{
    if (!v) return 0u;
    const int pb = 32 - __builtin_clz(poly);
    for (int i = 31; i >= pb - 1; i--)
        if (v & (1u << i)) v ^= poly << (i - pb + 1);
    return v;
}

// remainder of every 1- and 2-bit error pattern -> (number of errors, pattern); csrc/create.cpp irdm_create
std::vector<int2> syndrome_table(unsigned poly, int nbits, int max_err, int size)
{
    std::vector<int2> t((size_t)size, make_int2(-1, 0));
    for (int b1 = 0; b1 < nbits; b1++) {
        const unsigned v = 1u << b1, r = poly_rem(poly, v);
        if (r < (unsigned)size) t[r] = make_int2(1, (int)v);
    }
    if (max_err >= 2)
        for (int b1 = 0; b1 < nbits; b1++)
            for (int b2 = b1 + 1; b2 < nbits; b2++) {
                const unsigned v = (1u << b1) | (1u << b2), r = poly_rem(poly, v);
                if (r < (unsigned)size && t[r].x < 0) t[r] = make_int2(2, (int)v);
            }
    return t;
}

}  // namespace

extern "C" {

// bits: [n][kMaxBits] hard bits, llr: [n][kMaxBits] (ignored unless use_llr), n_bits[n]; out: n DecodedOut
int bitlayer_emul_frame_decode(const uint8_t *bits, const float *llr, const int *n_bits, int n, int use_llr, DecodedOut *out)
{
    std::vector<DemodOut> frames(n);
    for (int i = 0; i < n; i++) {
        memset(&frames[i], 0, sizeof(DemodOut));
        frames[i].ok = 1;
        frames[i].n_symbols = n_bits[i] / 2;
        memcpy(frames[i].bits, bits + (size_t)i * kMaxBits, kMaxBits);
        if (llr) memcpy(frames[i].llr, llr + (size_t)i * kMaxBits, sizeof(float) * kMaxBits);
    }
    std::vector<int2> ra = syndrome_table(1207u, 31, 2, 1024), hdr = syndrome_table(29u, 7, 1, 16);
    memset(out, 0, sizeof(DecodedOut) * n);
    return launch_frame_decode(frames.data(), n, ra.data(), hdr.data(), use_llr, n_bits, out, nullptr);
}

int bitlayer_emul_sizes(int *decoded_bytes, int *max_bits)
{
    *decoded_bytes = (int)sizeof(DecodedOut);
    *max_bits = kMaxBits;
    return 0;
}

}
