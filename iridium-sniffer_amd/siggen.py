"""Seeded synthetic Iridium-like IQ generator.

Host-side test/bench input only: complex AWGN plus DQPSK bursts
[16 preamble | 12 unique-word | P payload] symbols at 25 ksym/s, RRC alpha=0.4,
placed on an arbitrary carrier.  Nothing here is on the product path.

Symbol model (what the reference demodulator expects, qpsk_demod.c:217-225,
:264-273, iridium.h:30-31): quadrant q -> exp(j(pi/4 + q*pi/2)).
"""
import numpy as np

SYMBOL_RATE = 25000
UW_DL = [0, 2, 2, 2, 2, 0, 0, 0, 2, 0, 0, 2]
UW_UL = [2, 2, 0, 0, 0, 2, 0, 0, 2, 0, 2, 2]
DQPSK_MAP = [0, 2, 3, 1]           # decoded = MAP[(new-old)%4]   (qpsk_demod.c:46)
DQPSK_INV = {0: 0, 2: 1, 3: 2, 1: 3}

# RAW line documented in the reference (ARCHITECTURE.md:264 / :270): PRBS15 burst,
# 12 UW + 179 payload symbols -> 382 bits.
KNOWN_ANSWER_BITS = (
    "0011000000110000111100111000000000000011000000000000101000000000001111000000"
    "0000100010000000001100110000000010101010000000111111110000001000000010000011"
    "0000001100001010000010100011110000111100100010001000101100110011001110101010"
    "1010100111111111111101000000000000111000000000001001000000000011011000000000"
    "1011010000000011101110000000100110010000001101010110000010111111010000111000"
    "00")


def rrc_pulse(sps, alpha=0.4, span=5):
    """Unit-energy root-raised-cosine, +-span symbols, times sqrt(sps)."""
    n = np.arange(-span * sps, span * sps + 1, dtype=np.float64)
    t = n / sps
    h = np.empty_like(t)
    for i, ti in enumerate(t):
        if abs(ti) < 1e-12:
            h[i] = 1.0 - alpha + 4 * alpha / np.pi
        elif abs(abs(ti) - 1 / (4 * alpha)) < 1e-9:
            h[i] = alpha / np.sqrt(2) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha))
                                         + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
        else:
            h[i] = (np.sin(np.pi * ti * (1 - alpha)) + 4 * alpha * ti * np.cos(np.pi * ti * (1 + alpha))) \
                   / (np.pi * ti * (1 - (4 * alpha * ti) ** 2))
    h /= np.sqrt(np.sum(h * h))
    return h * np.sqrt(sps)


def bits_to_quadrants(bits):
    """Inverse of decode_dqpsk + map_symbols_to_bits for a whole frame (UW included)."""
    q, old = [], 0
    for i in range(0, len(bits), 2):
        v = (int(bits[i]) << 1) | int(bits[i + 1])
        old = (old + DQPSK_INV[v]) % 4
        q.append(old)
    return q


def quadrants_to_bits(quads):
    out, old = [], 0
    for s in quads:
        v = DQPSK_MAP[(s - old) % 4]
        old = s
        out += [(v >> 1) & 1, v & 1]
    return out


def frame_quadrants(payload_quads, uplink=False):
    """preamble + UW + payload as raw quadrant symbols."""
    if uplink:
        pre = [2 if (i % 2 == 0) else 0 for i in range(16)]
        uw = UW_UL
    else:
        pre = [0] * 16
        uw = UW_DL
    return pre + list(uw) + list(payload_quads)


def make_burst(fs, quads, freq_hz, phase, amp=0.05, alpha=0.4, span=5):
    """Complex baseband burst at sample rate fs (must be a multiple of 25 kHz).

    Pulse shaping is evaluated in polyphase form (each output sample is the sum of the
    2*span+1 symbol pulses that overlap it), float64, deterministic."""
    sps = int(round(fs / SYMBOL_RATE))
    sym = np.exp(1j * (np.pi / 4 + np.asarray(quads, dtype=np.float64) * np.pi / 2))
    nsym = len(sym)
    h = rrc_pulse(sps, alpha, span) / np.sqrt(sps)   # peak ~ 1 per symbol before amp
    nj = 2 * span + 1
    hp = np.zeros((nj, sps))
    hp.reshape(-1)[:len(h)] = h                       # hp[j, r] = h[j*sps + r]
    out = np.zeros((nsym + nj - 1, sps), dtype=np.complex128)
    for j in range(nj):
        out[j:j + nsym, :] += sym[:, None] * hp[j][None, :]
    sig = out.reshape(-1)[:(nsym + 2 * span - 1) * sps + 1]
    n = np.arange(len(sig), dtype=np.float64)
    sig = amp * sig * np.exp(1j * (2 * np.pi * freq_hz / fs * n + phase))
    return sig.astype(np.complex64)


def make_stream(fs, n_samples, bursts, noise_sigma=0.002, seed=0):
    """bursts: list of dict(start, freq_hz, quads | payload, uplink, amp, phase).

    Returns (iq complex64 [n_samples], truth list)."""
    rng = np.random.default_rng(seed)
    iq = (rng.standard_normal(n_samples, dtype=np.float32)
          + 1j * rng.standard_normal(n_samples, dtype=np.float32)).astype(np.complex64)
    iq *= np.float32(noise_sigma)
    truth = []
    for b in bursts:
        quads = b.get("quads")
        if quads is None:
            quads = frame_quadrants(b["payload"], b.get("uplink", False))
        ph = b.get("phase", rng.uniform(0, 2 * np.pi))
        sig = make_burst(fs, quads, b["freq_hz"], ph, b.get("amp", 0.05))
        s = int(b["start"])
        e = min(n_samples, s + len(sig))
        if e > s:
            iq[s:e] += sig[:e - s]
        truth.append(dict(start=s, freq_hz=b["freq_hz"], quads=list(quads), n=len(sig)))
    return iq, truth


def channel_freq(ch, extra=1234.0):
    """Carrier on the 41.667 kHz Iridium grid (relative to capture centre)."""
    return ch * (1e6 / 24.0) + extra


def standard_scene(fs, n_samples, n_bursts, seed, payload_range=(119, 179), first_start=None,
                   fft_size=None, uplink_every=0, amp=0.05, simplex=False):
    """SURVEY 8d scene: first burst after >= 512 FFT frames of noise, bursts spread out."""
    rng = np.random.default_rng(seed + 1000)
    if fft_size is None:
        fft_size = 1 << int(round(np.log2(fs / 1000.0)))
    first = first_start if first_start is not None else 520 * fft_size
    span = n_samples - first - int(0.05 * fs)
    starts = np.sort(rng.integers(0, max(span, 1), size=n_bursts)) + first
    half_ch = int((fs / 2 - 60e3) // (1e6 / 24.0))
    bursts = []
    for i, s in enumerate(starts):
        p = int(rng.integers(payload_range[0], payload_range[1] + 1))
        payload = rng.integers(0, 4, size=p).tolist()
        ch = int(rng.integers(-half_ch, half_ch + 1))
        if ch == 0:
            ch = 1
        bursts.append(dict(start=int(s), freq_hz=channel_freq(ch), payload=payload,
                           uplink=bool(uplink_every and (i % uplink_every == uplink_every - 1)),
                           amp=amp))
    return make_stream(fs, n_samples, bursts, seed=seed)


def to_ci16(iq, scale=131072.0):
    x = np.empty(2 * len(iq), dtype=np.float32)
    x[0::2] = iq.real
    x[1::2] = iq.imag
    return np.clip(np.round(x * scale), -32768, 32767).astype(np.int16)


def to_ci8(iq, scale=512.0):
    x = np.empty(2 * len(iq), dtype=np.float32)
    x[0::2] = iq.real
    x[1::2] = iq.imag
    return np.clip(np.round(x * scale), -128, 127).astype(np.int8)
