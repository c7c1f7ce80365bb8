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


def newest(pattern):
    """gpurun merges every call's files into the same directories: take the latest run's"""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


stats = newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
have_bench = os.path.exists(os.path.join(src, "bench.json")) and os.path.getsize(os.path.join(src, "bench.json")) > 0
if have_bench:
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
command = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else ""


def short(name):
    return name.replace("hipkkt::", "").replace("void ", "").split("(")[0]


traffic = collections.defaultdict(lambda: dict(calls=0, fetch_kb=0.0, write_kb=0.0))
for key, d in (("fetch_kb", "pmc_fetch"), ("write_kb", "pmc_write")):
    f = newest(os.path.join(src, d, "*", "*_counter_collection.csv"))[0]
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
# MFMA pipe: SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs, so the pipe's
# utilisation inside a kernel is busy / (gui_active / 8 * 1024) = (busy / gui_active) / 128
mf = newest(os.path.join(src, "pmc_mfma", "*", "*_counter_collection.csv"))
if mf:
    busy = collections.defaultdict(lambda: dict(calls=0, busy=0.0, gui=0.0))
    for r in csv.DictReader(open(mf[0])):
        k = short(r["Kernel_Name"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[k]["busy"] += float(r["Counter_Value"])
            busy[k]["calls"] += 1
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            busy[k]["gui"] += float(r["Counter_Value"])
    with open(os.path.join(dst, f"{tag}_mfma_busy.csv"), "w") as out:
        out.write("# %s -- separate pass: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE\n" % command)
        out.write("# mfma_pipe_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)\n")
        out.write("kernel,calls,SQ_VALU_MFMA_BUSY_CYCLES_total,GRBM_GUI_ACTIVE_total,mfma_pipe_utilisation\n")
        for k, v in sorted(busy.items(), key=lambda kv: -kv[1]["busy"]):
            if v["busy"] > 0:
                out.write(f"{k},{v['calls']},{v['busy']:.0f},{v['gui']:.0f},{v['busy'] / max(v['gui'], 1.0) / 128.0:.4f}\n")
if not have_bench:
    json.dump(dict(tag=tag, command=command, note="kernel stats, HBM traffic and MFMA counters only (no bench line for this command)"),
              open(os.path.join(dst, f"{tag}_traffic_summary.json"), "w"), indent=1)
    sys.exit(0)
bench = json.loads(open(os.path.join(src, "bench.json")).read())
# bench.py runs the steps three times (timed, un-instrumented sequential, instrumented sequential) after the warm-up;
# from round 3 on its line says how many value updates (factorisations) the process ran
steps = bench.get("updates_in_run", 3 * bench["steps"] + bench["warmup"])
TRI = ("k_fwd", "k_bwd", "k_top_solve")
FAC = ("k_panel", "k_schur", "k_front", "k_subtree", "k_tinv", "k_winv")


def total(prefixes, key):
    return sum(v[key] for k, v in traffic.items() if k.startswith(prefixes))


# Calibration of FETCH_SIZE on this access width (MI355X_MICROARCH.md, HBM: "other access widths are
# uncalibrated: calibrate on a known byte count in your own access pattern").  k_sum2 (y = a + b), k_absmax and
# k_pack_rhs are pure 8-byte-per-lane coalesced streams of known length N doubles per operand in this very run.
N = bench["config"]["N"]
calib = {}
for k, nread, nwrite in (("k_sum2", 2, 1), ("k_absmax", 1, 0), ("k_pack_rhs", 1, 1)):
    if k in traffic and traffic[k]["calls"]:
        c = traffic[k]["calls"]
        calib[k] = dict(expected_read_KB=nread * N * 8 / 1024.0, FETCH_SIZE_KB=traffic[k]["fetch_kb"] / c,
                        expected_write_KB=nwrite * N * 8 / 1024.0, WRITE_SIZE_KB=traffic[k]["write_kb"] / c)
ratios = [v["expected_read_KB"] / v["FETCH_SIZE_KB"] for v in calib.values() if v["FETCH_SIZE_KB"] > 0]
fetch_scale = round(sum(ratios) / len(ratios)) if ratios else 2          # measured: 2 (as for 16 B/lane reads)
ntri = bench["phases"]["trisolve"]["launches"] / bench["steps"] * steps        # (profiled with --sequential-solves)
tri_raw = (total(TRI, "fetch_kb") + total(TRI, "write_kb")) / ntri / 1024.0
fac_raw = (total(FAC, "fetch_kb") + total(FAC, "write_kb")) / steps / 1024.0
tri = (fetch_scale * total(TRI, "fetch_kb") + total(TRI, "write_kb")) / ntri / 1024.0
fac = (fetch_scale * total(FAC, "fetch_kb") + total(FAC, "write_kb")) / steps / 1024.0
summary = dict(tag=tag, command=command, steps_profiled=steps, csrc_sha16=bench["config"].get("csrc_sha16"),
               trisolve_hbm_MB_per_solve=tri, factor_hbm_MB_per_factorisation=fac,
               trisolve_hbm_MB_per_solve_uncorrected=tri_raw, factor_hbm_MB_per_factorisation_uncorrected=fac_raw,
               fetch_size_scale=fetch_scale, calibration=calib,
               note="FETCH_SIZE/WRITE_SIZE from separate rocprofv3 --pmc passes (KB as reported; counter collection serialises the kernels, so these passes run with HIPKKT_FACTOR_OVERLAP=0 -- same kernels and bytes, the top levels' tiles behind their panels instead of beside them).  On gfx950 FETCH_SIZE "
                    "reports 1/2 of the bytes of coalesced streaming reads (MI355X_MICROARCH.md, HBM section); the "
                    "calibration rows show the same factor for this code's 8-byte-per-lane streams, so reads are "
                    "doubled (an upper bound for the gather-type accesses).  WRITE_SIZE is exact.  Infinity-cache hits "
                    "are counted, not excluded.")
json.dump(summary, open(os.path.join(dst, f"{tag}_traffic_summary.json"), "w"), indent=1)
print(json.dumps(summary))
