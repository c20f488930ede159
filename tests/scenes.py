"""Edge-case scenes for the detector state machine (SURVEY.md 8c "negative cases"): squelch + history reset,
bursts longer than max_burst_len, carriers on DC / in the guard bands, simultaneous strong bursts, more concurrent
bursts than the sparse scan keeps in its lanes.  Seeded, generated with siggen; the oracle decides what is right."""
import numpy as np

import siggen


def _payload(rng, n=150):
    return rng.integers(0, 4, n).tolist()


def squelch(fs=2_000_000, seed=21):
    """44 carriers start within 2 ms: more than max_bursts (40 @ 2 MHz) -> squelch dumps them, squelch_count reaches
    10 -> noise-floor history reset and re-priming (burst_detect.c:594-631); a second wave after re-priming decodes."""
    n = int(2.7 * fs) // 32768 * 32768
    rng = np.random.default_rng(seed)
    first = 530 * 2048
    chans = [c for c in range(-22, 23) if c != 0]
    bursts = [dict(start=first + 4000 + 37 * i, freq_hz=siggen.channel_freq(ch), payload=_payload(rng), amp=0.03)
              for i, ch in enumerate(chans)]
    for i, ch in enumerate(chans[:6]):
        bursts.append(dict(start=first + 2_400_000 + 70_000 * i, freq_hz=siggen.channel_freq(ch), payload=_payload(rng)))
    return fs, siggen.make_stream(fs, n, bursts, seed=seed)[0]


def too_long(fs=2_000_000, seed=22):
    """A 0.36 s carrier: ended by max_burst_len (burst_detect.c:499-502) with the forced baseline update, re-created,
    while normal bursts come and go beside it."""
    n = int(1.5 * fs) // 32768 * 32768
    rng = np.random.default_rng(seed)
    first = 530 * 2048
    bursts = [dict(start=first + 1000, freq_hz=siggen.channel_freq(5), quads=_payload(rng, 9000)),
              dict(start=first + 150_000, freq_hz=siggen.channel_freq(-7), payload=_payload(rng, 160)),
              dict(start=first + 420_000, freq_hz=siggen.channel_freq(12), payload=_payload(rng, 160)),
              dict(start=first + 1_200_000, freq_hz=siggen.channel_freq(-3), payload=_payload(rng, 160))]
    return fs, siggen.make_stream(fs, n, bursts, seed=seed)[0]


def dc_and_edges(fs=2_000_000, seed=23):
    """Carriers on DC (bins dc+-3 never produce peaks, burst_detect.c:537-542), beside the notch, inside the guard
    bands (peaks only in [w/2, N-w/2))."""
    n = int(1.3 * fs) // 32768 * 32768
    rng = np.random.default_rng(seed)
    first = 530 * 2048
    bursts = [dict(start=first + 1000, freq_hz=300.0, payload=_payload(rng)),
              dict(start=first + 200_000, freq_hz=-2500.0, payload=_payload(rng)),
              dict(start=first + 400_000, freq_hz=fs / 2 - 15_000.0, payload=_payload(rng)),
              dict(start=first + 600_000, freq_hz=-fs / 2 + 30_000.0, payload=_payload(rng)),
              dict(start=first + 800_000, freq_hz=siggen.channel_freq(9), payload=_payload(rng))]
    return fs, siggen.make_stream(fs, n, bursts, seed=seed)[0]


def strong_simultaneous(fs=2_000_000, seed=24):
    """Five strong bursts with the same start and length (created and deleted in the same frames, long prefilter
    lists), an adjacent-channel pile-up with a weak neighbour in a strong burst's skirt, two near-threshold bursts."""
    n = int(1.3 * fs) // 32768 * 32768
    rng = np.random.default_rng(seed)
    first = 530 * 2048
    bursts = [dict(start=first + 5000, freq_hz=siggen.channel_freq(ch),
                   quads=siggen.frame_quadrants(_payload(rng, 170)), amp=0.4) for ch in (-15, -6, 4, 11, 18)]
    for i, ch in enumerate((-12, 2, 14)):
        bursts.append(dict(start=first + 300_000 + 3000 * i, freq_hz=siggen.channel_freq(ch), payload=_payload(rng, 170), amp=0.3))
    bursts.append(dict(start=first + 301_500, freq_hz=siggen.channel_freq(3), payload=_payload(rng, 170), amp=0.02))
    bursts.append(dict(start=first + 700_000, freq_hz=siggen.channel_freq(-20), payload=_payload(rng, 170), amp=0.004))
    bursts.append(dict(start=first + 900_000, freq_hz=siggen.channel_freq(20), payload=_payload(rng, 170), amp=0.006))
    return fs, siggen.make_stream(fs, n, bursts, seed=seed)[0]


def many_active_10m(fs=10_000_000, seed=25, n_sim=70):
    """70 concurrent bursts at 10 MHz: below max_bursts (200) so no squelch, above the 64 lane slots of the sparse
    scan -> it must abort and the dense scan must take over for that chunk."""
    n = int(0.75 * fs) // 32768 * 32768
    rng = np.random.default_rng(seed)
    first = 530 * 8192
    chans = [c for c in range(-110, 111, 3) if c != 0][:n_sim]
    bursts = [dict(start=first + 3000 + 211 * i, freq_hz=siggen.channel_freq(ch), payload=_payload(rng, 160), amp=0.03)
              for i, ch in enumerate(chans)]
    for i in range(6):
        bursts.append(dict(start=first + 1_500_000 + 150_000 * i,
                           freq_hz=siggen.channel_freq(int(rng.integers(-100, 100)) or 1), payload=_payload(rng, 160)))
    return fs, siggen.make_stream(fs, n, bursts, seed=seed)[0]


ALL = dict(squelch=squelch, too_long=too_long, dc_and_edges=dc_and_edges,
           strong_simultaneous=strong_simultaneous, many_active_10m=many_active_10m)


def frame_lengths(fs=2_000_000, seed=26, simplex=False):
    """Payload lengths around the frame-length rules (burst_downmix.c:763-777): shorter than the minimum (dropped),
    exactly minimum / maximum, longer than the maximum (cut).  simplex=True is meant to be run with a capture centre
    above 1626 MHz (80..444 symbols instead of 131..191)."""
    n = int(1.9 * fs) // 32768 * 32768
    rng = np.random.default_rng(seed)
    first = 530 * 2048
    lens = (40, 67, 68, 69, 300, 431, 432, 433, 640) if simplex else (60, 118, 119, 120, 150, 178, 179, 180, 320)
    bursts = [dict(start=first + 2000 + 290_000 * i, freq_hz=siggen.channel_freq(3 + 2 * i), payload=_payload(rng, p))
              for i, p in enumerate(lens)]
    return fs, siggen.make_stream(fs, n, bursts, seed=seed)[0]


def junk(fs=2_000_000, seed=27):
    """Things that are detected but are not Iridium frames: an unmodulated carrier, a 1 ms blip, QPSK without preamble
    or unique word, a burst whose unique word is damaged in 1 / 2 / 3 symbols (hard check tolerance, soft rescue,
    rejection: qpsk_demod.c:277-325), two bursts on the same channel 3 ms apart."""
    n = int(1.9 * fs) // 32768 * 32768
    rng = np.random.default_rng(seed)
    first = 530 * 2048
    bursts = [dict(start=first + 2000, freq_hz=siggen.channel_freq(-9), quads=[0] * 400),
              dict(start=first + 150_000, freq_hz=siggen.channel_freq(6), quads=_payload(rng, 25)),
              dict(start=first + 300_000, freq_hz=siggen.channel_freq(-4), quads=_payload(rng, 200))]
    for k, nbad in enumerate((1, 2, 3, 5)):
        q = siggen.frame_quadrants(_payload(rng, 160))
        for j in range(nbad):
            q[16 + 2 * j + 1] = (q[16 + 2 * j + 1] + 1 + (j & 1)) % 4
        bursts.append(dict(start=first + 450_000 + 150_000 * k, freq_hz=siggen.channel_freq(10 - 3 * k), quads=q))
    bursts.append(dict(start=first + 1_100_000, freq_hz=siggen.channel_freq(15), payload=_payload(rng, 150)))
    bursts.append(dict(start=first + 1_100_000 + 6000 + 16_000, freq_hz=siggen.channel_freq(15), payload=_payload(rng, 150)))
    return fs, siggen.make_stream(fs, n, bursts, seed=seed)[0]


def cfo_spread(fs=2_000_000, seed=28):
    """Carriers off the channel grid by up to +-12 kHz (coarse bin rounding, the 4096-point CFO estimate and its
    parabolic refinement, burst_downmix.c:482-535), at amplitudes from near-threshold to strong."""
    n = int(1.9 * fs) // 32768 * 32768
    rng = np.random.default_rng(seed)
    first = 530 * 2048
    offs = (-12_000.0, -7_300.0, -488.3, 0.0, 244.1, 488.28125, 3_111.0, 9_765.0, 11_999.0)
    amps = (0.05, 0.007, 0.2, 0.05, 0.011, 0.05, 0.6, 0.05, 0.02)
    bursts = [dict(start=first + 2000 + 200_000 * i, freq_hz=siggen.channel_freq(int(rng.integers(-20, 21)) or 2, extra=o),
                   payload=_payload(rng, 140 + 4 * i), amp=a, uplink=bool(i % 3 == 2))
              for i, (o, a) in enumerate(zip(offs, amps))]
    return fs, siggen.make_stream(fs, n, bursts, seed=seed)[0]


ALL.update(frame_lengths=frame_lengths, junk=junk, cfo_spread=cfo_spread)


def random_scene(seed, fs=2_000_000, secs=1.6):
    """Randomised differential-test scene: 10-60 emitters with log-uniform amplitudes (near-threshold .. very strong),
    random lengths (a few symbols .. longer than max_burst_len), random off-grid carriers (adjacent-channel pile-ups
    included), starts clustered in a few groups so that many bursts are born and die in the same frames."""
    rng = np.random.default_rng(10_000 + seed)
    nfft = 1 << int(round(np.log2(fs / 1000.0)))
    n = int(secs * fs) // 32768 * 32768
    first = 530 * nfft
    n_em = int(rng.integers(10, 61))
    groups = np.sort(rng.integers(first, n - int(0.25 * fs), size=int(rng.integers(2, 7))))
    bursts = []
    for _ in range(n_em):
        g = int(rng.choice(groups))
        start = g + int(rng.integers(0, int(0.02 * fs))) if rng.random() < 0.7 else int(rng.integers(first, n - int(0.05 * fs)))
        kind = rng.random()
        if kind < 0.6:
            quads = siggen.frame_quadrants(_payload(rng, int(rng.integers(100, 200))), uplink=bool(rng.random() < 0.3))
        elif kind < 0.8:
            quads = _payload(rng, int(rng.integers(5, 400)))
        elif kind < 0.92:
            quads = [0] * int(rng.integers(20, 1500))
        else:
            quads = _payload(rng, int(rng.integers(2300, 4000)))          # longer than max_burst_len (2250 symbols)
        f = float(rng.uniform(-fs / 2 + 25e3, fs / 2 - 25e3))
        amp = float(np.exp(rng.uniform(np.log(0.0035), np.log(0.5))))
        bursts.append(dict(start=start, freq_hz=f, quads=quads, amp=amp))
    return fs, siggen.make_stream(fs, n, bursts, seed=seed)[0]


UW_QUADS = {1: [0, 2, 2, 2, 2, 0, 0, 0, 2, 0, 0, 2], 2: [2, 2, 0, 0, 0, 2, 0, 0, 2, 0, 2, 2]}        # qpsk_demod.c:40-44


def stage_c_frames(n, sps, max_frame=4440, seed=77):
    """Downmixed frames for stage-C tests (qpsk_demod.c:393-535): symbols at quadrant centres (the unique word first), `sps`
    samples apart, straight lines between them, a constant phase within the slicer's margin, noise -- the timing loop and
    the PLL lock at any sps.  Lengths from 40 samples to the 4440 of a simplex frame, the first six fixed (maximum, maximum,
    one short of it, the normal maximum 1910, one more, 43).  Returns (buf [n, 2 * max_frame] float32, lens, directions)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lens = [int(v) for v in rng.integers(40, max_frame + 1, n)]
    lens[:6] = [max_frame, max_frame, max_frame - 1, 1910, 1911, 43][:min(6, n)]
    dirs = np.ascontiguousarray(rng.integers(1, 3, n), np.int32)
    buf = np.zeros((n, 2 * max_frame), np.float32)
    r = np.random.default_rng(seed + int(sps * 100))
    for i in range(n):
        nsym = int(lens[i] / sps) + 3
        q = np.concatenate([UW_QUADS[int(dirs[i])], r.integers(0, 4, nsym)])
        sym = 0.4 * np.exp(1j * (np.pi / 4 + q * np.pi / 2 + r.uniform(-0.2, 0.2)))
        t = np.arange(lens[i]) / sps
        k = t.astype(int)
        x = sym[k] * (1 - (t - k)) + sym[k + 1] * (t - k)
        x = x + (r.standard_normal(lens[i]) + 1j * r.standard_normal(lens[i])) * 0.4 * float(r.choice([0.01, 0.05, 0.15]))
        buf[i, 0:2 * lens[i]:2] = x.real
        buf[i, 1:2 * lens[i]:2] = x.imag
    return buf, lens, dirs
