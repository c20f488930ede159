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
