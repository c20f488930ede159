// bitlayer.hip -- post-demod bit layer on gfx950: frame_decode() (frame_decode.c:414-598), SURVEY 8f row 3.
//
//   access code check (:428-431) -> IBC: BCH(7,3) header (:441-452), 2-way de-interleave (:156-176), BCH(31,21)
//   with Chase decoding on the LLRs (:224-295), parity (:399-407), field extraction (:368-393)
//   -> IRA: 3-way de-interleave (:178-199), three header blocks, paging blocks (:317-366).
//
// Integer / bitwise work, independent per frame: one lane per frame.  Codewords live in 32-bit registers (bit 30 =
// first bit, as bits_to_uint builds them); the de-interleavers are address arithmetic on the frame's bit array.  The
// only floats are the LLR comparisons of the Chase decoder's selection of the five least reliable positions (first
// minimum wins, :256-266) -- ordering only, so the result is exact.  lat / lon / alt (double atan2 / sqrt, :336-342) are
// finished on the host with the host libm from the integer position this kernel returns.
#include "common.hpp"
#include "types.hpp"
#include "kernels.hpp"

namespace irdm {

namespace {

constexpr unsigned kPolyRa = 1207u;   // BCH(31,21), frame_decode.c:36
constexpr unsigned kPolyHdr = 29u;    // BCH(7,3),   frame_decode.c:37
constexpr int kChase = 5;             // frame_decode.c:48

__device__ __forceinline__ unsigned gf2_rem(unsigned poly, int poly_bits, unsigned val)      // :82-91
{
    for (int i = 31; i >= poly_bits - 1; i--)
        if (val & (1u << i)) val ^= poly << (i - poly_bits + 1);
    return val;
}

// one de-interleaved 32-bit block: symbols first, first-stride, ... (16 of them), two bits each
struct Block {
    unsigned cw;        // bits 0..30 of the block, first bit at position 30 (bits_to_uint(block32, 31))
    unsigned parity;    // bit 31 of the block
};

__device__ __forceinline__ Block gather_block(const uint8_t *__restrict__ in, int first_sym, int stride)
{
    unsigned w = 0;
#pragma unroll
    for (int p = 0; p < 16; p++) {
        const int s = first_sym - stride * p;
        w = (w << 2) | ((unsigned)(in[2 * s] & 1) << 1) | (unsigned)(in[2 * s + 1] & 1);
    }
    Block b;
    b.cw = w >> 1;
    b.parity = w & 1u;
    return b;
}

// LLR of block position k (0..30) of the same gather
__device__ __forceinline__ float block_llr(const float *__restrict__ llr, int first_sym, int stride, int k)
{
    const int s = first_sym - stride * (k >> 1);
    return llr[2 * s + (k & 1)];
}

// chase_bch_decode_p (:224-295): corrected codeword in *out (data = out >> 10, check = out & 0x3ff); returns the error
// count or -1
__device__ int chase_bch(const Block &b, const float *__restrict__ llr, int first_sym, int stride,
                         const int2 *__restrict__ syn_ra, unsigned *out)
{
    unsigned val = b.cw;
    unsigned syn = gf2_rem(kPolyRa, 11, val);
    if (syn == 0) { *out = val; return 0; }
    if (syn < 1024 && syn_ra[syn].x >= 0) { *out = val ^ (unsigned)syn_ra[syn].y; return syn_ra[syn].x; }
    if (!llr) return -1;
    // the five least reliable positions, partial selection sort with "first minimum wins" (:256-266): equivalent to
    // five passes of strict-less arg-min over the positions not yet taken, in the permuted order the swaps produce.
    // The permutation matters for ties only through the order of comparison; it is reproduced literally.
    int pos[31];
    for (int i = 0; i < 31; i++) pos[i] = i;
    for (int i = 0; i < kChase; i++) {
        int mi = i;
        float mv = block_llr(llr, first_sym, stride, pos[i]);
        for (int j = i + 1; j < 31; j++) {
            const float v = block_llr(llr, first_sym, stride, pos[j]);
            if (v < mv) { mv = v; mi = j; }
        }
        const int t = pos[i]; pos[i] = pos[mi]; pos[mi] = t;
    }
    unsigned fm[kChase];
    for (int i = 0; i < kChase; i++) fm[i] = 1u << (30 - pos[i]);
    for (int mask = 1; mask < (1 << kChase); mask++) {
        unsigned f = b.cw;
        for (int k = 0; k < kChase; k++)
            if (mask & (1 << k)) f ^= fm[k];
        syn = gf2_rem(kPolyRa, 11, f);
        if (syn == 0) { *out = f; return 0; }
        if (syn < 1024 && syn_ra[syn].x >= 0) { *out = f ^ (unsigned)syn_ra[syn].y; return syn_ra[syn].x; }
    }
    return -1;
}

// check_parity32 (:399-407): data + check + parity bit have even weight
__device__ __forceinline__ bool parity_ok(unsigned corrected, unsigned parity_bit)
{
    return ((__popc(corrected & 0x7fffffffu) + (int)parity_bit) & 1) == 0;
}

// the decoded data bits, 21 per block, packed MSB-first into 32-bit words of `stream`
__device__ __forceinline__ void append21(unsigned *stream, int &len, unsigned corrected)
{
    const unsigned d = (corrected >> 10) & 0x1fffffu;
    for (int i = 0; i < 21; i++) {
        const unsigned bit = (d >> (20 - i)) & 1u;
        const int k = len + i;
        stream[k >> 5] |= bit << (31 - (k & 31));
    }
    len += 21;
}

__device__ __forceinline__ unsigned sbit(const unsigned *stream, int k) { return (stream[k >> 5] >> (31 - (k & 31))) & 1u; }

__device__ __forceinline__ unsigned sfield(const unsigned *stream, int k, int n)             // extract_uint, :309-315
{
    unsigned v = 0;
    for (int i = 0; i < n; i++) v = (v << 1) | sbit(stream, k + i);
    return v;
}

__device__ __forceinline__ int ssigned12(const unsigned *stream, int k)                      // extract_signed12, :299-307
{
    const int mag = (int)sfield(stream, k + 1, 11);
    return sbit(stream, k) ? mag - (1 << 11) : mag;
}

// remaining 64-bit blocks (:478-497, :569-588)
__device__ void more_blocks(const uint8_t *data, const float *llr, int offset, int limit, const int2 *syn_ra,
                            unsigned *stream, int cap_bits, int &len)
{
    while (offset + 64 <= limit && len + 42 <= cap_bits) {
        const Block b1 = gather_block(data + offset, 31, 2), b2 = gather_block(data + offset, 30, 2);
        unsigned c1, c2;
        const int ea = chase_bch(b1, llr ? llr + offset : nullptr, 31, 2, syn_ra, &c1);
        const int eb = chase_bch(b2, llr ? llr + offset : nullptr, 30, 2, syn_ra, &c2);
        if (ea < 0 || eb < 0) break;
        if (!parity_ok(c1, b1.parity)) break;
        if (!parity_ok(c2, b2.parity)) break;
        append21(stream, len, c1);
        append21(stream, len, c2);
        offset += 64;
    }
}

}  // namespace

__global__ __launch_bounds__(64) void frame_decode_kernel(const DemodOut *__restrict__ frames, int n_frames,
                                                          const int2 *__restrict__ syn_ra,
                                                          const int2 *__restrict__ syn_hdr, int use_llr,
                                                          const int *__restrict__ n_bits_in,
                                                          DecodedOut *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_frames) return;
    const DemodOut &f = frames[i];
    DecodedOut o;
    memset(&o, 0, sizeof(o));
    // demod_frame_t.n_bits is 2 * n_symbols (qpsk_demod.c:478); the stage-alone entry passes arbitrary lengths
    const int n_bits = n_bits_in ? n_bits_in[i] : (f.ok ? 2 * f.n_symbols : 0);
    const uint8_t *bits = f.bits;
    // access codes (:51-56) as 24-bit words, first bit most significant
    unsigned acc = 0;
    if (n_bits >= 24)
        for (int k = 0; k < 24; k++) acc = (acc << 1) | (unsigned)(bits[k] & 1);
    const bool is_dl = acc == 0x3030F3u, is_ul = acc == 0xCC3CFCu;
    if (n_bits < 24 || (!is_dl && !is_ul)) { out[i] = o; return; }
    const uint8_t *data = bits + 24;
    const float *llr = use_llr ? f.llr + 24 : nullptr;
    const int data_len = n_bits - 24;

    if (data_len >= 6 + 64) {                                           // ---- IBC (:441-505)
        unsigned hv = 0;
        for (int k = 0; k < 6; k++) hv = (hv << 1) | (unsigned)(data[k] & 1);
        const unsigned hs = gf2_rem(kPolyHdr, 5, hv);
        bool hdr_ok = false;
        if (hs == 0) hdr_ok = true;
        else if (hs < 16 && syn_hdr[hs].x >= 0) { hv ^= (unsigned)syn_hdr[hs].y; hdr_ok = true; }
        if (hdr_ok) {
            const Block b1 = gather_block(data + 6, 31, 2), b2 = gather_block(data + 6, 30, 2);
            unsigned c1, c2;
            const int e1 = chase_bch(b1, llr ? llr + 6 : nullptr, 31, 2, syn_ra, &c1);
            const int e2 = chase_bch(b2, llr ? llr + 6 : nullptr, 30, 2, syn_ra, &c2);
            if (e1 >= 0 && e2 >= 0 && parity_ok(c1, b1.parity) && parity_ok(c2, b2.parity)) {
                unsigned stream[8];                                     // 256 bits (:466)
                for (int k = 0; k < 8; k++) stream[k] = 0;
                int len = 0;
                append21(stream, len, c1);
                append21(stream, len, c2);
                const int ibc_max = data_len < 262 ? data_len : 262;
                more_blocks(data, llr, 6 + 64, ibc_max, syn_ra, stream, 256, len);
                o.type = 2;
                o.bch_len = len;
                o.bc_type = (int)((hv >> 4) & 7u);
                if (len >= 42) {                                        // parse_ibc (:368-393)
                    o.sat_id = (int)sfield(stream, 0, 7);
                    o.beam_id = (int)sfield(stream, 7, 6);
                    o.timeslot = (int)sbit(stream, 14);
                    o.sv_blocking = (int)sbit(stream, 15);
                    if (len >= 84 && sfield(stream, 42, 6) == 1u) o.iri_time = sfield(stream, 52, 32);
                }
                out[i] = o;
                return;
            }
        }
    }

    if (data_len >= 96) {                                               // ---- IRA (:514-595)
        const Block b1 = gather_block(data, 47, 3), b2 = gather_block(data, 46, 3), b3 = gather_block(data, 45, 3);
        unsigned c1, c2, c3;
        const int e1 = chase_bch(b1, llr, 47, 3, syn_ra, &c1);
        const int e2 = chase_bch(b2, llr, 46, 3, syn_ra, &c2);
        const int e3 = chase_bch(b3, llr, 45, 3, syn_ra, &c3);
        if (e1 >= 0 && e2 >= 0 && e3 >= 0 && parity_ok(c1, b1.parity) && parity_ok(c2, b2.parity) &&
            parity_ok(c3, b3.parity)) {
            unsigned stream[16];                                        // 512 bits (:545)
            for (int k = 0; k < 16; k++) stream[k] = 0;
            int len = 0;
            append21(stream, len, c1);
            append21(stream, len, c2);
            append21(stream, len, c3);
            more_blocks(data, llr, 96, data_len, syn_ra, stream, 512, len);
            o.type = 1;
            o.bch_len = len;
            if (len >= 63) {                                            // parse_ira (:317-366)
                o.sat_id = (int)sfield(stream, 0, 7);
                o.beam_id = (int)sfield(stream, 7, 6);
                o.pos_xyz[0] = ssigned12(stream, 13);
                o.pos_xyz[1] = ssigned12(stream, 25);
                o.pos_xyz[2] = ssigned12(stream, 37);
                int off = 63;
                while (off + 42 <= len && o.n_pages < 12) {
                    bool all1 = true;
                    for (int k = 0; k < 42; k++)
                        if (!sbit(stream, off + k)) { all1 = false; break; }
                    if (all1) break;
                    o.page_tmsi[o.n_pages] = sfield(stream, off, 32);
                    o.page_msc[o.n_pages] = (int)sfield(stream, off + 34, 5);
                    o.n_pages++;
                    off += 42;
                }
            }
        }
    }
    out[i] = o;
}

int launch_frame_decode(const DemodOut *frames, int n_frames, const int2 *syn_ra, const int2 *syn_hdr, int use_llr,
                        const int *n_bits, DecodedOut *out, hipStream_t stream)
{
    if (n_frames <= 0) return 0;
    hipLaunchKernelGGL(frame_decode_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, stream, frames, n_frames,
                       syn_ra, syn_hdr, use_llr, n_bits, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace irdm
