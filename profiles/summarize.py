#!/usr/bin/env python3
"""Condense gpurun_out/<dir> (tools/measure_round.sh: rocprofv3 --kernel-trace --stats runs, the FETCH_SIZE / WRITE_SIZE
passes and the decimator's SQ-counter passes, 10 MHz cfg3 and 12 MHz / 40 bursts per Msample) into
  profiles/<tag>_kernel_stats[_depth0|_cfg5|_cfg5_depth0].csv   per-kernel launches / total / average
  profiles/<tag>_pmc.json, <tag>_pmc_cfg5.json                  HBM bytes per launch (FETCH_SIZE doubled for the wide
                                                                coalesced readers as MI355X_MICROARCH.md prescribes)
  profiles/<tag>_fir_pmc.json                                   the decimator's SQ counters per launch, both scenes
  profiles/<tag>_bench_*.json                                   the bench lines of the run
Usage: python profiles/summarize.py gpurun_out/<dir> <tag>"""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:70]


def stats(infile, outfile):
    if not os.path.exists(infile):
        return
    with open(outfile, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,pct\n")
        for r in csv.DictReader(open(infile)):
            if "irdm::" not in r["Name"] and "rocclr" not in r["Name"]:
                continue
            f.write("%s,%s,%.3f,%.2f,%s\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                             float(r["AverageNs"]) / 1e3, r["Percentage"]))


stats(f"{src}/r1_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
stats(f"{src}/r1d0_kernel_stats.csv", f"profiles/{tag}_kernel_stats_depth0.csv")
stats(f"{src}/c5_kernel_stats.csv", f"profiles/{tag}_kernel_stats_cfg5_12mhz_d40.csv")
stats(f"{src}/c5d0_kernel_stats.csv", f"profiles/{tag}_kernel_stats_cfg5_12mhz_d40_depth0.csv")


def pmc(prefix, outfile):
    acc = collections.defaultdict(dict)
    for name, key in ((prefix + "pmc_fetch", "FETCH_SIZE"), (prefix + "pmc_write", "WRITE_SIZE")):
        path = f"{src}/{name}_counter_collection.csv"
        if not os.path.exists(path):
            return
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if "irdm::" in r["Kernel_Name"]:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            acc[k][key + "_KB_mean"] = sum(v) / len(v)
            acc[k]["launches_" + key] = len(v)
    out = {}
    for k, d in acc.items():
        fetch = d.get("FETCH_SIZE_KB_mean", 0.0) * 1024 * 2      # gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads
        write = d.get("WRITE_SIZE_KB_mean", 0.0) * 1024
        out[k] = dict(d, hbm_read_bytes=fetch, hbm_write_bytes=write, traffic_bytes=fetch + write)
    json.dump(out, open(outfile, "w"), indent=1, sort_keys=True)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes"])[:6]:
        print("%-22s %-50s traffic %.1f MB" % (os.path.basename(outfile), k, d["traffic_bytes"] / 1e6))


pmc("", f"profiles/{tag}_pmc.json")
pmc("c5_", f"profiles/{tag}_pmc_cfg5.json")

sq = {}
for scene, prefix in (("cfg3_10mhz_d10", "sq"), ("cfg5_12mhz_d40", "c5_sq")):
    vals = collections.defaultdict(list)
    dur = []
    for i in (1, 2):
        path = f"{src}/{prefix}{i}/pmc_counter_collection.csv"
        if not os.path.exists(path):
            continue
        seen = set()
        for r in csv.DictReader(open(path)):
            if "fir_decimate" not in r["Kernel_Name"]:
                continue
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            kernel, grid, vg = short(r["Kernel_Name"]), int(r["Grid_Size"]), r["VGPR_Count"]
    if vals:
        sq[scene] = {"kernel": kernel, "grid_threads_last_launch": grid, "launch_us_mean_with_counters": sum(dur) / len(dur),
                     "per_launch_mean": {k: sum(v) / len(v) for k, v in sorted(vals.items())}}
if sq:
    json.dump(sq, open(f"profiles/{tag}_fir_pmc.json", "w"), indent=1, sort_keys=True)
    print(json.dumps(sq, indent=1))

for name in ("b", "b0", "d2", "d40", "cfg5_12mhz_d40", "lds_fir", "scalar_fir", "cfg4_n1", "cfg4_n1_first8chunks"):
    if os.path.exists(f"{src}/{name}.json") and os.path.getsize(f"{src}/{name}.json") > 10:
        shutil.copy(f"{src}/{name}.json", f"profiles/{tag}_bench_{name}.json")
for name in ("k1_bench.txt", "hop_timing.txt"):
    if os.path.exists(f"{src}/{name}") and os.path.getsize(f"{src}/{name}") > 10:
        shutil.copy(f"{src}/{name}", f"profiles/{tag}_{name}")
# K1 alone: SQ counters per launch (tools/ubench/k1_bench under rocprofv3 --pmc)
k1 = {}
for i in (1, 2):
    path = f"{src}/k1sq{i}/pmc_counter_collection.csv"
    if not os.path.exists(path):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        k1.setdefault(k, {})[c] = sum(v) / len(v)
for k, d in k1.items():
    if all(c in d for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")):
        tot = d["SQ_WAIT_ANY"] + d["SQ_WAIT_INST_ANY"] + d["SQ_ACTIVE_INST_ANY"]
        d["wave_time_split"] = {"issuing": round(d["SQ_ACTIVE_INST_ANY"] / tot, 3), "waiting_for_issue": round(d["SQ_WAIT_INST_ANY"] / tot, 3),
                                "parked_waitcnt_barrier": round(d["SQ_WAIT_ANY"] / tot, 3)}
    if "SQ_WAVES" in d:
        d["per_wave"] = {c: round(v / d["SQ_WAVES"], 1) for c, v in d.items() if c.startswith("SQ_INSTS")}
if k1:
    json.dump(k1, open(f"profiles/{tag}_k1_pmc_final.json", "w"), indent=1, sort_keys=True)
# the scan's device timeline: per pass [us from the first workgroup's start to the last one's end, idle us in front of it,
# launches per chunk], the plan pass's phase stamps and the walk's event counts (bench.py config.*)
tl = {}
for name, label in (("tl", "cfg3_in_run"), ("tl_depth0", "cfg3_alone"), ("tl_cfg5", "cfg5_12mhz_d40_in_run"),
                    ("tl_cfg5_depth0", "cfg5_12mhz_d40_alone")):
    path = f"{src}/{name}.json"
    if os.path.exists(path) and os.path.getsize(path) > 10:
        e = json.load(open(path))
        c = e["config"]
        tl[label] = {"Msamples_per_s": e["value"], "ms_per_step": e["ms_per_step"], "scan_stage_ms": e["roofline"]["stage_ms"]["scan"],
                     "passes_us_dur_gap_launches": c.get("scan_timeline_us"), "walk_events": c.get("walk_events"),
                     "plan_phase_us": c.get("plan_phase_us"), "scan": c.get("scan")}
if tl:
    json.dump(tl, open(f"profiles/{tag}_scan_timeline.json", "w"), indent=1)
    for k, v in tl.items():
        p = v["passes_us_dur_gap_launches"] or {}
        print(k, v["Msamples_per_s"], "scan", v["scan_stage_ms"], "dur", round(sum(x[0] for x in p.values())), "gap",
              round(sum(x[1] for x in p.values())))

# what the group protocol costs on one GPU (tools/group_bench.py: one JSON line per mode)
gb = f"{src}/group_bench.txt"
if os.path.exists(gb) and os.path.getsize(gb) > 10:
    runs = [json.loads(l) for l in open(gb) if l.startswith("{")]
    json.dump({"what": "tools/group_bench.py on one MI355X: a group of one member over a 12 MHz stream in 64 Mi-sample chunks, device-resident, "
                       "the next super-step staged ahead -- plain (chunks straight into the member's ring, no hand-off) and with "
                       "group_loopback (slice and overlap through the landing buffers, state export -> ncclSend / ncclRecv to itself -> "
                       "import in front of every chunk)", "runs": runs}, open(f"profiles/{tag}_group_bench.json", "w"), indent=1)
    for r in runs:
        print("group", r["mode"], r["Msamples_per_s"], r["ms_per_chunk"])

for f in (f"profiles/{tag}_kernel_stats.csv", f"profiles/{tag}_kernel_stats_depth0.csv",
          f"profiles/{tag}_kernel_stats_cfg5_12mhz_d40.csv", f"profiles/{tag}_kernel_stats_cfg5_12mhz_d40_depth0.csv"):
    if os.path.exists(f):
        print("==", f)
        print("".join(open(f).readlines()[:24]))
