#!/usr/bin/env python3
"""A rocprofv3 --kernel-trace CSV of a bench.py run as text: per-kernel totals over the traced steady state, how much of the
time a chip-filling kernel (decimator, K1, post1, post2) was on the chip, how long the scan's stream was occupied, and a
few periods as start / end / duration / queue / kernel lines (what profiles/r*_kernel_trace_gantt.txt holds).
Usage: python tools/trace_gantt.py <..._kernel_trace.csv> [periods=4] > profiles/rN_kernel_trace_gantt.txt"""
import collections
import csv
import sys

WIDE = ("fir_decimate", "fft_mag", "downmix_post1", "downmix_post2", "post_tiles", "post_cfo")
SCAN = ("band_", "prefilter", "detect_scan", "gone_export")


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("irdm::", "")
    return name.split("<")[0].split("(")[0]


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    path = sys.argv[1]
    periods = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows = []
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Name")
        if "irdm::" not in name:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(name)))
    rows.sort()
    k1 = [r for r in rows if r[3].startswith("fft_mag")]
    if len(k1) < periods + 3:
        print("# too few K1 launches in the trace:", len(k1))
        return
    # steady state: from the K1 launch `periods + 2` before the last one to the last one
    t_lo, t_hi = k1[-(periods + 2)][0], k1[-2][0]
    n_per = periods
    win = [r for r in rows if r[0] >= t_lo and r[0] < t_hi]
    span = (t_hi - t_lo) / 1e3
    print("# %s" % path)
    print("# window: %d periods between K1 launches, %.1f us per period" % (n_per, span / n_per))
    wide = [(s, e) for s, e, q, n in win if n.startswith(WIDE)]
    scan = [(s, e) for s, e, q, n in win if n.startswith(SCAN)]
    print("# a chip-filling kernel on the chip: %.1f us per period (sum of their spans %.1f)" %
          (union(wide) / 1e3 / n_per, sum(e - s for s, e in wide) / 1e3 / n_per))
    print("# the scan's passes on the chip: %.1f us per period (sum %.1f)" %
          (union(scan) / 1e3 / n_per, sum(e - s for s, e in scan) / 1e3 / n_per))
    agg = collections.OrderedDict()
    for s, e, q, n in win:
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    print("#\n# per kernel: launches, average us, us per period")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-32s %4d %9.1f %9.1f" % (n, c, t / c, t / n_per))
    queues = {}
    print("#\n# start us, end us, duration us, queue, kernel")
    for s, e, q, n in win:
        qn = queues.setdefault(q, "q%d" % len(queues))
        print("%9.1f %9.1f %7.1f  %-3s %s" % ((s - t_lo) / 1e3, (e - t_lo) / 1e3, (e - s) / 1e3, qn, n))


if __name__ == "__main__":
    main()
