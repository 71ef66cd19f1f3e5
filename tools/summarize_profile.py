#!/usr/bin/env python3
"""Condense a scripts_prof.sh output directory (rocprofv3 CSVs) into profiles/<tag>_*.

    python tools/summarize_profile.py gpurun_out/prof_<tag> <tag> <lattices>

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim),
profiles/<tag>_summary.json (per-kernel dispatch durations split by grid size, PMC counters per
dispatch with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md §HBM) and refreshes
profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag, lattices = sys.argv[1], sys.argv[2], int(sys.argv[3])
frames_per_lattice = int(sys.argv[4]) if len(sys.argv) > 4 else 50000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)


def one(pattern):
    files = glob.glob(os.path.join(src, pattern))
    return files[0] if files else None


summary = {"tag": tag, "lattices_per_launch": lattices, "source": "rocprofv3 --kernel-trace --stats / --pmc (separate passes)"}
ks = one("stats/*/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
kt = one("stats/*/*kernel_trace.csv")
if kt:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "hash_" in name or "at::" in name or "rocclr" in name:
            continue
        grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
        per[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    summary["dispatch_ms"] = [
        {"kernel": k, "grid_threads": g, "calls": len(v), "avg_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v)}
        for (k, g), v in sorted(per.items())]
pmc = {}
sq_ms = collections.defaultdict(list)      # dispatch durations inside the SQ / GRBM counter pass (for the clock under load)
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = one(f"{name}/*/*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "hash_" in k or "at::" in k or "rocclr" in k:
            continue
        key = (k, int(r["Grid_Size"]))
        pmc.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if name == "pmc_sq" and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            sq_ms[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
if not sq_ms:
    kt_sq = one("pmc_sq/*/*kernel_trace.csv")
    if kt_sq:
        for r in csv.DictReader(open(kt_sq)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
            sq_ms[(k, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
rows = []
for (k, g), c in sorted(pmc.items()):
    rows.append({"kernel": k, "grid_threads": g, "counters_per_dispatch": {n: sum(v) / len(v) for n, v in c.items()}})
summary["pmc"] = rows
# the few-lattice forms, one cfg2 lattice per launch
single = {}
for name in ("pmc_fetch_single", "pmc_write_single"):
    f = one(f"{name}/*/*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        single.setdefault((k, int(r["Grid_Size"])), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
summary["pmc_one_lattice"] = [
    {"kernel": k, "grid_threads": g, "counters_per_dispatch": {n: sum(v) / len(v) for n, v in c.items()},
     "hbm_bytes": (sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 2048 if "FETCH_SIZE" in c else 0)
                  + (sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024 if "WRITE_SIZE" in c else 0)}
    for (k, g), c in sorted(single.items())]
# HBM traffic of the dominant (forward) kernel's batch launch: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
# FETCH_SIZE reports half of a coalesced streaming read (MI355X_MICROARCH.md §HBM) -> x2.
for r in rows:
    if "forward_ck" in r["kernel"] and "false" in r["kernel"] and r["grid_threads"] == lattices * 64:
        c = r["counters_per_dispatch"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            read_b = c["FETCH_SIZE"] * 1024 * 2
            write_b = c["WRITE_SIZE"] * 1024
            summary["forward_hbm_bytes_per_launch"] = {"read_corrected_x2": read_b, "write": write_b, "total": read_b + write_b}
            # whole step: every kernel of the batch launch (forward, backtrace, gather), same correction
            job = 0.0
            for q in rows:
                if q["grid_threads"] >= lattices:
                    cc = q["counters_per_dispatch"]
                    job += cc.get("FETCH_SIZE", 0.0) * 2048 + cc.get("WRITE_SIZE", 0.0) * 1024
            frames = lattices * frames_per_lattice
            valu = c.get("SQ_INSTS_VALU")
            # shader clock while the forward kernel ran: GRBM_GUI_ACTIVE counts busy cycles of all 8 XCDs
            clock = None
            dur = sq_ms.get((r["kernel"], r["grid_threads"]))
            if dur and c.get("GRBM_GUI_ACTIVE"):
                clock = c["GRBM_GUI_ACTIVE"] / 8.0 / (sum(dur) / len(dur) * 1e6)
            summary["forward_clock_ghz"] = clock
            summary["job_hbm_bytes_per_step"] = job
            summary["forward_valu_per_frame"] = valu / frames if valu else None
            with open(os.path.join(out_dir, "pmc_traffic.json"), "wt") as f:
                json.dump({"lattices": lattices, "hbm_bytes_per_launch": read_b + write_b, "tag": tag,
                           "job_hbm_bytes_per_step": job,
                           "forward_valu_per_frame": valu / frames if valu else None,
                           "forward_clock_ghz": clock,
                           "note": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 of forward_ck_kernel<4,false>, one launch; "
                                   "job = the same sum over the step's forward, backtrace and gather launches; "
                                   "forward_valu_per_frame = SQ_INSTS_VALU of the forward launch / frames; "
                                   "forward_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / the launch's duration in the same counter pass"}, f)
bj = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(bj):
    try:
        summary["bench_line_under_rocprof"] = json.loads(open(bj).read().strip().splitlines()[-1])
    except Exception:
        pass
with open(os.path.join(out_dir, f"{tag}_summary.json"), "wt") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary.get("dispatch_ms", []), indent=1))
print(summary.get("forward_hbm_bytes_per_launch"))
