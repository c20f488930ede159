"""-m gpu: device memory of a context.  The rotator checkpoint table (rotator.h:36-46 restated as checkpoints of the
phase recurrence, one row per FFT bin) is by far the largest allocation and depends on the sample rate only: contexts of
one rate on one device share it, and it lives until the last of them is closed."""
import numpy as np
import pytest
import torch

import irdm
import orc
import parity
import siggen

pytestmark = pytest.mark.gpu


def _used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return total - free


def test_contexts_of_one_rate_share_the_rotator_table():
    fs = 2_000_000
    n = int(0.6 * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, 5, seed=5)
    ref = orc.run_stream(iq, fs)
    kw = dict(max_chunk_samples=n, max_bursts_per_chunk=256)
    u0 = _used()
    a = irdm.Pipeline(fs, **kw)
    u1 = _used()
    b = irdm.Pipeline(fs, **kw)
    u2 = _used()
    first, second = u1 - u0, u2 - u1
    # 2048 bins x (longest burst window / 16 + 2) checkpoints x 8 bytes > 100 MB at 2 MHz
    assert second < first - 100e6, (first, second)

    def run(p):
        p.set_option("keep_frame_samples", 1)
        p.feed_host(iq)
        bursts = p.poll_bursts()
        infos, samples = p.poll_frames()
        return dict(bursts=bursts, infos=infos, samples=samples, demods=p.poll_demods(), tagged=p.tagged)

    parity.compare(run(a), ref)
    a.close()                         # the table stays: b still holds it
    parity.compare(run(b), ref)
    b.close()
    c = irdm.Pipeline(fs, **kw)       # and is rebuilt after the last user is gone
    parity.compare(run(c), ref)
    c.close()
