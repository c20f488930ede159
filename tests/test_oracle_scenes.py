"""CPU: the edge-case scenes really exercise what they claim (the oracle's view), so the GPU parity tests on the
same scenes (test_gpu_scenes.py) cover those branches of the reference's state machine."""
import numpy as np

import orc
import scenes


def test_squelch_scene_dumps_resets_and_recovers():
    fs, iq = scenes.squelch()
    r = orc.run_stream(iq, fs)
    early = [b for b in r.bursts if b.start < 530 * 2048 + 1_000_000]
    late = [b for b in r.bursts if b.start >= 530 * 2048 + 2_000_000]
    assert len(early) >= 30                                   # the dumped wave
    assert max(b.stop - b.start for b in early) < 20_000      # squelched long before a normal burst would end
    assert len(late) == 6 and len(r.demods) == 6              # decoded again after re-priming


def test_too_long_scene_forces_burst_ends():
    fs, iq = scenes.too_long()
    r = orc.run_stream(iq, fs)
    max_len = int(0.09 * fs)
    assert sum(1 for b in r.bursts if b.stop - b.start > max_len) >= 2
    assert len(r.demods) >= 3


def test_dc_scene_never_centres_a_burst_on_the_notch():
    fs, iq = scenes.dc_and_edges()
    r = orc.run_stream(iq, fs)
    n = 2048
    assert r.bursts and all(abs(b.center_bin - n // 2) > 3 for b in r.bursts)
    assert all(40 // 2 <= b.center_bin < n - 40 // 2 for b in r.bursts)


def test_many_active_scene_exceeds_the_sparse_scan_slots():
    fs, iq = scenes.many_active_10m()
    r = orc.run_stream(iq, fs)
    ev = sorted([(b.start, 1) for b in r.bursts] + [(b.stop, -1) for b in r.bursts])
    cur = peak = 0
    for _, d in ev:
        cur += d
        peak = max(peak, cur)
    assert peak > 64 and len(r.demods) >= 70
