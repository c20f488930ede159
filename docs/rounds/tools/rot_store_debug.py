"""Measurement aid: the same 2 MHz scene through the pipeline with rot_store 0 and 1; where do the downmixed frames differ?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "iridium-sniffer_amd"))
import numpy as np
import irdm
import parity
import siggen

fs = 2_000_000
iq = siggen.standard_scene(fs, int(2.4 * fs), 8, seed=11, uplink_every=4)[0]
res = {}
for v in (0, 1, 3):
    res[v] = parity.run_gpu(iq, fs, options={"rot_store": v})
p = irdm.Pipeline(fs, max_chunk_samples=65536, max_bursts_per_chunk=64)
p.set_option("rot_store", 1)
p.close()
for v in (1, 3):
  a, b = res[0], res[v]
  print("rot_store", v, "frames", len(a["infos"]), len(b["infos"]))
  for fa, sa, fb, sb in zip(a["infos"], a["samples"], b["infos"], b["samples"]):
      same = fa.num_samples == fb.num_samples and np.array_equal(sa.view(np.uint32), sb.view(np.uint32))
      d = np.nonzero(sa.view(np.uint32).reshape(-1, 2) != sb.view(np.uint32).reshape(-1, 2))[0] if fa.num_samples == fb.num_samples else []
      print("id", fa.id, "drop", fa.drop_reason, fb.drop_reason, "start", fa.start, "dec_len", fa.dec_len, "L", fa.dec_len - fa.start, "n", fa.num_samples,
            "uw_start_idx", fa.uw_start_idx, fb.uw_start_idx, "corr", fa.corr_re, fb.corr_re, "same" if same else "DIFF first %s last %s count %d" % (d[:6], d[-3:], len(d)))
