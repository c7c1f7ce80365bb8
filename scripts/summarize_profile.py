#!/usr/bin/env python3
"""Condense a gpurun_out/profile_<tag>/ directory (scripts/profile_bench.sh) into the small
files committed under profiles/: the rocprofv3 kernel stats, and HBM traffic per kernel from
the FETCH_SIZE / WRITE_SIZE passes (KB units as rocprofv3 reports them)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))


def short(name):
    return name.replace("hipkkt::", "").replace("void ", "").split("(")[0]


traffic = collections.defaultdict(lambda: dict(calls=0, fetch_kb=0.0, write_kb=0.0))
for key, d in (("fetch_kb", "pmc_fetch"), ("write_kb", "pmc_write")):
    f = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))[0]
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        traffic[k][key] += float(r["Counter_Value"])
        if key == "fetch_kb":
            traffic[k]["calls"] += 1
with open(os.path.join(dst, f"{tag}_hbm_traffic.csv"), "w") as out:
    out.write("kernel,calls,FETCH_SIZE_KB_total,WRITE_SIZE_KB_total,fetch_KB_per_call,write_KB_per_call\n")
    for k, v in sorted(traffic.items(), key=lambda kv: -(kv[1]["fetch_kb"] + kv[1]["write_kb"])):
        c = max(v["calls"], 1)
        out.write(f"{k},{v['calls']},{v['fetch_kb']:.1f},{v['write_kb']:.1f},{v['fetch_kb']/c:.2f},{v['write_kb']/c:.2f}\n")
bench = json.loads(open(os.path.join(src, "bench.json")).read())
steps = bench["steps"] + bench["warmup"]
tri = sum(v["fetch_kb"] + v["write_kb"] for k, v in traffic.items() if k.startswith(("k_fwd", "k_bwd")))
fac = sum(v["fetch_kb"] + v["write_kb"] for k, v in traffic.items()
          if k.startswith(("k_panel", "k_schur", "k_front_wave", "k_tinv")))
ntri = bench["phases"]["trisolve"]["launches"] / bench["steps"] * steps
summary = dict(tag=tag, steps_profiled=steps,
               trisolve_hbm_MB_per_solve=tri / ntri / 1024.0,
               factor_hbm_MB_per_factorisation=fac / steps / 1024.0,
               note="FETCH_SIZE/WRITE_SIZE as reported (KB); FETCH_SIZE may under-count wide streaming reads "
                    "by up to 2x on gfx950 (MI355X_MICROARCH.md, HBM section); 8-byte-per-lane loads here are uncalibrated")
json.dump(summary, open(os.path.join(dst, f"{tag}_traffic_summary.json"), "w"), indent=1)
print(json.dumps(summary))
