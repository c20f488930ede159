"""Post-demod bit layer: the oracle's orc_frame_decode against the reference's frame_decode.c compiled
in place (oracle/_ref), on encoded IRA / IBC frames with known answers, bit errors inside and beyond the BCH and Chase
correction radius, truncated frames, and garbage."""
import ctypes as C

import numpy as np
import pytest

import bitlayer as bl
import orc


class Decoded(C.Structure):
    _fields_ = [("type", C.c_int32), ("sat_id", C.c_int32), ("beam_id", C.c_int32), ("pos_xyz", C.c_int32 * 3),
                ("alt", C.c_int32), ("n_pages", C.c_int32), ("lat", C.c_double), ("lon", C.c_double),
                ("page_tmsi", C.c_uint32 * 12), ("page_msc", C.c_int32 * 12), ("timeslot", C.c_int32),
                ("sv_blocking", C.c_int32), ("bc_type", C.c_int32), ("iri_time", C.c_uint32),
                ("bch_len", C.c_int32), ("pad", C.c_int32)]


FIELDS = [f for f, _ in Decoded._fields_ if f not in ("bch_len", "pad")]


def as_tuple(d):
    out = []
    for f in FIELDS:
        v = getattr(d, f)
        if f in ("lat", "lon"):
            v = np.float64(v).view(np.uint64)            # bit-identical doubles
        elif hasattr(v, "__len__"):
            v = tuple(v)
        out.append(v)
    return tuple(out)


def decode_with(fn, bits, llr):
    b = np.ascontiguousarray(bits, np.uint8)
    d = Decoded()
    lp = None if llr is None else np.ascontiguousarray(llr, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    r = fn(b.ctypes.data_as(C.POINTER(C.c_uint8)), lp, len(b), C.byref(d))
    return r, d


def make_cases(seed, n=120):
    rng = np.random.default_rng(seed)
    cases = []
    for k in range(n):
        kind = k % 6
        if kind in (0, 1, 2):                            # IRA
            pages = [(int(rng.integers(0, 2**32)), int(rng.integers(0, 32))) for _ in range(int(rng.integers(0, 5)))]
            st = bl.ira_stream(int(rng.integers(0, 128)), int(rng.integers(0, 64)), int(rng.integers(-2047, 2048)),
                               int(rng.integers(-2047, 2048)), int(rng.integers(-2047, 2048)), pages, rng,
                               terminate=bool(rng.integers(0, 2)))
            if len(st) % 42 != 21:                       # 63 + 42k always is; keep the pairing of the tail blocks
                st = st[:63 + (len(st) - 63) // 42 * 42]
            bits = bl.ira_frame(st, uplink=bool(kind == 2))
        elif kind in (3, 4):                             # IBC
            st = bl.ibc_stream(int(rng.integers(0, 128)), int(rng.integers(0, 64)), int(rng.integers(0, 2)),
                               int(rng.integers(0, 2)), int(rng.integers(0, 2**32)) if kind == 3 else None, rng,
                               n_blocks=int(rng.integers(1, 5)))
            bits = bl.ibc_frame(int(rng.integers(0, 4)), st)
        else:                                            # garbage behind a valid access code / no access code
            bits = list(bl.ACCESS_DL if rng.integers(0, 2) else rng.integers(0, 2, 24)) + \
                   [int(b) for b in rng.integers(0, 2, int(rng.integers(0, 400)))]
        bits += [int(b) for b in rng.integers(0, 2, int(rng.integers(0, 70)))]      # trailing payload noise
        n_err = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 14, 30]))
        bits, llr = bl.corrupt(bits, rng, n_err, mark=bool(rng.integers(0, 4)), extra_weak=int(rng.integers(0, 6)))
        if rng.integers(0, 8) == 0:
            cut = int(rng.integers(0, len(bits)))
            bits, llr = bits[:cut], llr[:cut]
        cases.append((bits, None if rng.integers(0, 6) == 0 else llr))
    return cases


def test_encoder_round_trip_through_the_oracle(oracle):
    """clean frames decode to exactly the fields that were encoded"""
    oracle.orc_frame_decode.restype = C.c_int
    rng = np.random.default_rng(5)
    st = bl.ira_stream(77, 33, -1234, 987, 2000, [(0xDEADBEEF, 17), (12345, 3)], rng)
    r, d = decode_with(oracle.orc_frame_decode, bl.ira_frame(st), None)
    assert r == 1 and d.type == 1 and (d.sat_id, d.beam_id, tuple(d.pos_xyz)) == (77, 33, (-1234, 987, 2000))
    assert d.n_pages == 2 and d.page_tmsi[0] == 0xDEADBEEF and d.page_msc[0] == 17 and d.page_msc[1] == 3
    assert d.alt == int(np.sqrt(1234.0**2 + 987.0**2 + 2000.0**2) * 4.0) - 6378 + 23
    st = bl.ibc_stream(99, 12, 1, 0, 0xCAFEF00D, rng, n_blocks=3)
    r, d = decode_with(oracle.orc_frame_decode, bl.ibc_frame(2, st), None)
    assert r == 1 and d.type == 2 and (d.sat_id, d.beam_id, d.timeslot, d.sv_blocking, d.bc_type) == (99, 12, 1, 0, 2)
    assert d.iri_time == 0xCAFEF00D and d.bch_len == 3 * 42


@pytest.mark.parametrize("seed", range(4))
def test_oracle_matches_frame_decode_c(oracle, reflib, seed):
    oracle.orc_frame_decode.restype = C.c_int
    reflib.ref_frame_decode.restype = C.c_int
    kinds = {0: 0, 1: 0, 2: 0}
    for bits, llr in make_cases(seed):
        ro, do = decode_with(oracle.orc_frame_decode, bits, llr)
        rr, dr = decode_with(reflib.ref_frame_decode, bits, llr)
        assert ro == rr and as_tuple(do) == as_tuple(dr), (ro, rr, as_tuple(do), as_tuple(dr))
        kinds[do.type] += 1
    assert kinds[1] >= 20 and kinds[2] >= 10 and kinds[0] >= 20, kinds      # IRA, IBC and rejected frames all occur


class Ida(C.Structure):
    _fields_ = [("ok", C.c_int32), ("ft", C.c_int32), ("lcw_ft", C.c_int32), ("lcw_code", C.c_int32),
                ("ec_lcw", C.c_int32), ("lcw3_val", C.c_uint32), ("da_ctr", C.c_int32), ("da_len", C.c_int32),
                ("cont", C.c_int32), ("crc_ok", C.c_int32), ("stored_crc", C.c_uint32), ("computed_crc", C.c_uint32),
                ("fixederrs", C.c_int32), ("payload_len", C.c_int32), ("bch_len", C.c_int32), ("pad", C.c_int32),
                ("payload", C.c_uint8 * 32), ("bch_stream", C.c_uint8 * 256), ("lcw_header", C.c_char * 128)]


def ida_decode_with(fn, bits, llr, direction):
    b = np.ascontiguousarray(bits, np.uint8)
    d = Ida()
    lp = None if llr is None else np.ascontiguousarray(llr, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    r = fn(b.ctypes.data_as(C.POINTER(C.c_uint8)), lp, len(b), direction, C.byref(d))
    return r, d


def make_ida_cases(seed, n=120):
    rng = np.random.default_rng(1000 + seed)
    cases = []
    for k in range(n):
        ft = 2 if k % 8 else int(rng.integers(0, 8))                    # mostly IDA (ft == 2)
        lcw = bl.lcw_bits(ft, int(rng.integers(0, 32)), int(rng.integers(0, 1 << 21)))
        da_len = int(rng.integers(0, 21)) if k % 9 else int(rng.integers(21, 32))
        st = bl.ida_stream(int(rng.integers(0, 8)), da_len, int(rng.integers(0, 2)),
                           [int(b) for b in rng.integers(0, 256, 20)], rng, good_crc=bool(k % 5))
        if k % 11 == 0:
            st[17 + int(rng.integers(0, 3))] = 1                          # the "zero" field set -> rejected
        bits = bl.ida_frame(lcw, st, rng, uplink=bool(k % 3 == 0))
        n_err = int(rng.choice([0, 0, 1, 2, 4, 7, 12, 25]))
        bits, llr = bl.corrupt(bits, rng, n_err, mark=bool(rng.integers(0, 4)), extra_weak=int(rng.integers(0, 6)))
        if rng.integers(0, 10) == 0:
            cut = int(rng.integers(150, len(bits)))
            bits, llr = bits[:cut], llr[:cut]
        direction = 2 if k % 3 == 0 else (1 if k % 13 else 0)            # direction 0 = undefined -> rejected
        cases.append((bits, None if rng.integers(0, 6) == 0 else llr, direction))
    return cases


def ida_tuple(d):
    return (d.ok, d.ft, d.lcw_ft, d.lcw_code, d.ec_lcw, d.lcw3_val, d.da_ctr, d.da_len, d.cont, d.crc_ok, d.stored_crc,
            d.computed_crc, d.fixederrs, d.payload_len, d.bch_len, bytes(d.payload), bytes(d.bch_stream), d.lcw_header)


def test_ida_encoder_round_trip_through_the_oracle(oracle):
    oracle.orc_ida_decode.restype = C.c_int
    rng = np.random.default_rng(9)
    payload = list(range(100, 120))
    st = bl.ida_stream(5, 17, 1, payload, rng)
    r, d = ida_decode_with(oracle.orc_ida_decode, bl.ida_frame(bl.lcw_bits(2, 0b00011, 0x12345), st, rng), None, 1)
    assert r == 1 and (d.ft, d.da_ctr, d.da_len, d.cont, d.crc_ok, d.bch_len, d.fixederrs) == (2, 5, 17, 1, 1, 200, 0)
    assert list(d.payload[:17]) == payload[:17] and d.lcw3_val == 0x12345 and d.ec_lcw == 0
    assert d.lcw_header.decode().startswith("LCW(2,T:maint,C:") and len(d.lcw_header.decode()) == 111


@pytest.mark.parametrize("seed", range(4))
def test_oracle_matches_ida_decode_c(oracle, reflib, seed):
    oracle.orc_ida_decode.restype = C.c_int
    reflib.ref_ida_decode.restype = C.c_int
    n_ok = n_crc = n_rej = 0
    for bits, llr, direction in make_ida_cases(seed):
        ro, do = ida_decode_with(oracle.orc_ida_decode, bits, llr, direction)
        rr, dr = ida_decode_with(reflib.ref_ida_decode, bits, llr, direction)
        assert ro == rr and ida_tuple(do) == ida_tuple(dr), (ida_tuple(do)[:16], ida_tuple(dr)[:16])
        n_ok += ro
        n_crc += do.crc_ok
        n_rej += 1 - ro
    assert n_ok >= 30 and n_crc >= 10 and n_rej >= 20, (n_ok, n_crc, n_rej)


def test_lcw_header_text_for_every_type_and_code(oracle, reflib):
    """format_lcw_header's switch (ida_decode.c:405-539): every (lcw_ft, lcw_code) pair with random lcw3 values, through
    whole frames so the reference's own formatter runs"""
    oracle.orc_ida_decode.restype = C.c_int
    reflib.ref_ida_decode.restype = C.c_int
    rng = np.random.default_rng(77)
    seen = set()
    for d5 in range(32):
        for rep in range(6):
            lcw3 = int(rng.integers(0, 1 << 21))
            st = bl.ida_stream(1, 20, 0, [0] * 20, rng)
            bits = bl.ida_frame(bl.lcw_bits(2, d5, lcw3), st, rng)
            ro, do = ida_decode_with(oracle.orc_ida_decode, bits, None, 1)
            rr, dr = ida_decode_with(reflib.ref_ida_decode, bits, None, 1)
            assert ro == rr == 1 and do.lcw_header == dr.lcw_header, (d5, do.lcw_header, dr.lcw_header)
            seen.add((do.lcw_ft, do.lcw_code))
    assert len(seen) == 32          # the 5 encodable data bits reach 32 of the 64 (type, code) pairs
