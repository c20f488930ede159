"""Condenses the bench lines of one experiment run (gpurun_out/<dir>/*.json, written by tools/r5_experiments.sh) into one
committed file: per run the rate, the step, the stage brackets, the kernel clocks, the feeding thread's time per step, the
scan's counters and, where the run had it on, the scan's device timeline.
Usage: python profiles/condense_runs.py gpurun_out/r5_j profiles/r5_sum_restart.json "what the runs are" [name-prefix ...]"""
import glob
import json
import os
import sys


def find(d, key):
    if isinstance(d, dict):
        if key in d:
            return d[key]
        for v in d.values():
            r = find(v, key)
            if r is not None:
                return r
    return None


src, out, what = sys.argv[1], sys.argv[2], sys.argv[3]
prefixes = sys.argv[4:]
names = ("k1_ring_enqueue", "settle", "chain_enqueue", "scan_enqueue", "wait_older_chain", "final_sync", "settle_wait_scan",
         "settle_counters", "settle_records", "build_records")
runs = {}
for f in sorted(glob.glob(os.path.join(src, "*.json"))):
    name = os.path.basename(f)[:-5]
    if prefixes and not any(name.startswith(p) for p in prefixes):
        continue
    try:
        d = json.load(open(f))
    except Exception:
        continue
    n = d["steps"] + d["warmup"]
    h = find(d, "host_us_total") or {}
    r = {"Msamples_per_s": d["value"], "ms_per_step": d["ms_per_step"], "stage_ms": find(d, "stage_ms"),
         "kernel_clock_ms": find(d, "kernel_clock_ms"),
         "host_us_per_step": {k: round(h.get(k.replace("k1_ring_enqueue", "k1_ring").replace("wait_older_chain", "wait_older_chain"), h.get(k, 0)) / n) for k in names} if h else None,
         "scan": find(d, "scan")}
    alone = find(d, "stage_ms_alone")
    if alone:
        r["stage_ms_alone"] = alone
    tl = find(d, "scan_timeline_us")
    if tl:
        r["scan_timeline_us_pass_idle"] = {k: v[:2] for k, v in tl.items()}
    runs[name] = r
json.dump({"what": what, "runs": runs}, open(out, "w"), indent=1)
print(out, len(runs), "runs")
