#!/usr/bin/env python3
"""Condense gpurun_out/<dir> (rocprofv3 --kernel-trace --stats and the two --pmc passes) into
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json.  FETCH_SIZE is doubled for the wide coalesced
readers as MI355X_MICROARCH.md (HBM section) prescribes; WRITE_SIZE is taken as reported.  Units: bytes."""
import collections
import csv
import json
import sys

src, tag = sys.argv[1], sys.argv[2]


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:70]


rows = list(csv.DictReader(open(f"{src}/r1_kernel_stats.csv")))
import os
if os.path.exists(f"{src}/r1d0_kernel_stats.csv"):
    with open(f"profiles/{tag}_kernel_stats_depth0.csv", "w") as f:
        f.write("kernel,calls,total_ms,avg_us,pct\n")
        for r in csv.DictReader(open(f"{src}/r1d0_kernel_stats.csv")):
            if "irdm::" not in r["Name"] and "rocclr" not in r["Name"]:
                continue
            f.write("%s,%s,%.3f,%.2f,%s\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                             float(r["AverageNs"]) / 1e3, r["Percentage"]))
with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
    f.write("kernel,calls,total_ms,avg_us,pct\n")
    for r in rows:
        if "irdm::" not in r["Name"] and "rocclr" not in r["Name"]:
            continue
        f.write("%s,%s,%.3f,%.2f,%s\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                         float(r["AverageNs"]) / 1e3, r["Percentage"]))
pmc = collections.defaultdict(dict)
for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{src}/{name}_counter_collection.csv")):
        if "irdm::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k][key + "_KB_mean"] = sum(v) / len(v)
        pmc[k]["launches_" + key] = len(v)
out = {}
for k, d in pmc.items():
    fetch = d.get("FETCH_SIZE_KB_mean", 0.0) * 1024 * 2      # gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads
    write = d.get("WRITE_SIZE_KB_mean", 0.0) * 1024
    out[k] = dict(d, hbm_read_bytes=fetch, hbm_write_bytes=write, traffic_bytes=fetch + write)
json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1, sort_keys=True)
print(open(f"profiles/{tag}_kernel_stats.csv").read())
for k, d in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes"])[:8]:
    print("%-50s traffic %.1f MB" % (k, d["traffic_bytes"] / 1e6))
