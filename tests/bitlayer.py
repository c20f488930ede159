"""Test-side ENCODER for the frames frame_decode() (frame_decode.c) recognises: BCH(31,21) + parity blocks,
the Iridium 2-way / 3-way symbol interleavers, IRA (ring alert) and IBC (broadcast) layouts, access codes.
Independent of the decoder under test: it is the inverse written from the bit layout, used to build inputs with known
answers, with controlled bit errors and reliabilities for the Chase decoder."""
import numpy as np

ACCESS_DL = [0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 1, 1]
ACCESS_UL = [1, 1, 0, 0, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0]
POLY_RA, POLY_HDR = 1207, 29


def gf2_rem(poly, val):
    pb = poly.bit_length()
    for i in range(31, pb - 2, -1):
        if val & (1 << i):
            val ^= poly << (i - pb + 1)
    return val


def to_bits(v, n):
    return [(v >> (n - 1 - i)) & 1 for i in range(n)]


def bch_block32(data21):
    """21 data bits -> 32-bit block: data | 10 check bits | overall parity (even weight)"""
    v = 0
    for b in data21:
        v = (v << 1) | int(b)
    cw = (v << 10) | gf2_rem(POLY_RA, v << 10)
    bits = to_bits(cw, 31)
    return bits + [sum(bits) & 1]


def interleave2(b1, b2):
    """inverse of de_interleave: 2 x 32 bits -> 64 bits"""
    out = [0] * 64
    for p in range(16):
        s = 31 - 2 * p
        out[2 * s], out[2 * s + 1] = b1[2 * p], b1[2 * p + 1]
        s = 30 - 2 * p
        out[2 * s], out[2 * s + 1] = b2[2 * p], b2[2 * p + 1]
    return out


def interleave3(b1, b2, b3):
    out = [0] * 96
    for p in range(16):
        for blk, base in ((b1, 47), (b2, 46), (b3, 45)):
            s = base - 3 * p
            out[2 * s], out[2 * s + 1] = blk[2 * p], blk[2 * p + 1]
    return out


def s12(v):
    """12-bit field of extract_signed12: sign bit + 11 bits, value = mag - 2048 when the sign bit is set"""
    return [1] + to_bits(v + 2048, 11) if v < 0 else [0] + to_bits(v, 11)


def ira_stream(sat, beam, x, y, z, pages, rng, terminate=True):
    hdr = to_bits(sat, 7) + to_bits(beam, 6) + s12(x) + s12(y) + s12(z) + [int(b) for b in rng.integers(0, 2, 14)]
    assert len(hdr) == 63
    st = list(hdr)
    for tmsi, msc in pages:
        st += to_bits(tmsi, 32) + [int(b) for b in rng.integers(0, 2, 2)] + to_bits(msc, 5) + \
              [int(b) for b in rng.integers(0, 2, 3)]
    if terminate:
        st += [1] * 42
    return st


def ira_frame(stream, uplink=False):
    """data stream (multiple of 21 bits; 63 header + 42 per page) -> frame bits incl. access code"""
    assert len(stream) % 21 == 0 and len(stream) >= 63
    blocks = [bch_block32(stream[i:i + 21]) for i in range(0, len(stream), 21)]
    bits = list(ACCESS_UL if uplink else ACCESS_DL) + interleave3(*blocks[:3])
    rest = blocks[3:]
    for i in range(0, len(rest) - 1, 2):
        bits += interleave2(rest[i], rest[i + 1])
    return bits


def ibc_frame(bc_type, stream, uplink=False):
    """bc_type 0..3 (what 6 header bits can carry), stream = multiple of 42 data bits"""
    assert len(stream) % 42 == 0 and len(stream) >= 42
    hv = (bc_type << 4) | gf2_rem(POLY_HDR, bc_type << 4)
    bits = list(ACCESS_UL if uplink else ACCESS_DL) + to_bits(hv, 6)
    blocks = [bch_block32(stream[i:i + 21]) for i in range(0, len(stream), 21)]
    for i in range(0, len(blocks), 2):
        bits += interleave2(blocks[i], blocks[i + 1])
    return bits


def ibc_stream(sat, beam, timeslot, sv_blocking, iri_time, rng, n_blocks=2):
    b1 = to_bits(sat, 7) + to_bits(beam, 6) + [int(rng.integers(0, 2))] + [timeslot, sv_blocking] + \
         [int(b) for b in rng.integers(0, 2, 26)]
    st = list(b1)
    if n_blocks >= 2:
        typ = 1 if iri_time is not None else int(rng.integers(2, 64))
        st += to_bits(typ, 6) + [int(b) for b in rng.integers(0, 2, 4)] + to_bits(iri_time or 0, 32)
    for _ in range(n_blocks - 2):
        st += [int(b) for b in rng.integers(0, 2, 42)]
    return st


def corrupt(bits, rng, n_errors, first=24, reliable=1.0, weak=0.05, mark=True, extra_weak=0):
    """flip n_errors random bits after the access code; LLRs: `reliable` everywhere, `weak` on the flipped bits when
    mark (so the Chase decoder can find them), plus extra_weak decoys"""
    bits = list(bits)
    llr = np.full(len(bits), reliable, np.float32) + rng.random(len(bits)).astype(np.float32) * 0.1
    idx = rng.choice(np.arange(first, len(bits)), size=min(n_errors, len(bits) - first), replace=False)
    for i in idx:
        bits[i] ^= 1
        if mark:
            llr[i] = weak * float(rng.random())
    for i in rng.choice(np.arange(first, len(bits)), size=min(extra_weak, len(bits) - first), replace=False):
        llr[i] = weak * float(rng.random())
    return bits, llr


# ---------------------------------------------------------------- IDA (ida_decode.c) ----
POLY_DA, POLY_LCW2, POLY_LCW3 = 3545, 465, 41
LCW_PERM = [40, 39, 36, 35, 32, 31, 28, 27, 24, 23, 20, 19, 16, 15, 12, 11, 8, 7, 4, 3,
            41, 38, 37, 34, 33, 30, 29, 26, 25, 22, 21, 18, 17, 14, 13, 10, 9, 6, 5, 2,
            1, 46, 45, 44, 43, 42]


def lcw_bits(ft, d5, lcw3):
    """46 on-air LCW bits: lcw1 = BCH(7,3) of ft, lcw2 = 13-bit codeword of 5 data bits (generator 465; the decoder reads
    its top 6 bits as lcw_ft (2) + lcw_code (4)), lcw3 = BCH(26,21) of lcw3; permuted and pair-swapped as the air
    interface does (inverse of decode_lcw's two steps)"""
    v1 = (ft << 4) | gf2_rem(POLY_HDR, ft << 4)
    v2 = (d5 << 8) | gf2_rem(POLY_LCW2, d5 << 8)
    v3 = (lcw3 << 5) | gf2_rem(POLY_LCW3, lcw3 << 5)
    lb = to_bits(v1, 7) + to_bits(v2, 13) + to_bits(v3, 26)
    swapped = [0] * 46
    for i in range(46):
        swapped[LCW_PERM[i] - 1] = lb[i]
    data = [0] * 46
    for i in range(0, 46, 2):
        data[i + 1], data[i] = swapped[i], swapped[i + 1]
    return data


def crc_ccitt(data):
    crc = 0xFFFF
    for b in data:
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc


def ida_stream(da_ctr, da_len, cont, payload20, rng, good_crc=True):
    """200 decoded bits: header (cont @3, ctr @5-7, len @11-15, zero @17-19), 20 payload bytes, CRC-CCITT, 4 spare"""
    hdr = [int(b) for b in rng.integers(0, 2, 20)]
    hdr[3] = cont
    hdr[5:8] = to_bits(da_ctr, 3)
    hdr[11:16] = to_bits(da_len, 5)
    hdr[17:20] = [0, 0, 0]
    pl = []
    for by in payload20:
        pl += to_bits(by, 8)
    msg_bits = hdr + [0] * 12 + pl
    msg = [int("".join(map(str, msg_bits[i:i + 8])), 2) for i in range(0, len(msg_bits), 8)]
    crc = crc_ccitt(msg)
    if not good_crc:
        crc ^= 0x0101
    return hdr + pl + to_bits(crc, 16) + [int(b) for b in rng.integers(0, 2, 4)]


def da_block31(data20):
    v = int("".join(map(str, data20)), 2)
    return to_bits((v << 11) | gf2_rem(POLY_DA, v << 11), 31)


def interleave_n(h1, h2, n_sym):
    """inverse of de_interleave_n: h1 <- symbols n-1, n-3, ..; h2 <- symbols n-2, n-4, .."""
    out = [0] * (2 * n_sym)
    p = 0
    for s in range(n_sym - 1, 0, -2):
        out[2 * s], out[2 * s + 1] = h1[p], h1[p + 1]
        p += 2
    p = 0
    for s in range(n_sym - 2, -1, -2):
        out[2 * s], out[2 * s + 1] = h2[p], h2[p + 1]
        p += 2
    return out


def ida_frame(lcw, stream200, rng, uplink=False):
    """access code + LCW + two 124-bit blocks (chunks in air order 3,1,2,0) + the 64-bit tail block = 382 bits"""
    assert len(stream200) == 200
    ch = [da_block31(stream200[i:i + 20]) for i in range(0, 200, 20)]
    bits = list(ACCESS_UL if uplink else ACCESS_DL) + list(lcw)
    for blk in range(2):
        s = ch[4 * blk:4 * blk + 4]
        comb = s[3] + s[1] + s[2] + s[0]
        bits += interleave_n(comb[:62], comb[62:], 62)
    h2 = [int(rng.integers(0, 2))] + ch[8]
    h1 = [int(rng.integers(0, 2))] + ch[9]
    bits += interleave_n(h1, h2, 32)
    assert len(bits) == 382
    return bits
