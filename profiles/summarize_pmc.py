#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection.csv files (one pass per counter group)
into one per-kernel table: mean counter value per dispatch.

    python profiles/summarize_pmc.py [--frames=N] gpurun_out/pmc_*/p_counter_collection.csv > profiles/rNN_pmc_summary.csv

--frames=N: frames one dispatch of the network kernels covered in the profiled run (32 for the VGA batch on one
stream, 64 for the HD batch); written as column frames_per_dispatch so that bench.py can scale `roofline.traffic` to
the frames of its own launches.

HBM bytes per dispatch, as MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE under-counts wide (16 B/lane) coalesced
reads by exactly 2x, WRITE_SIZE is exact -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs).
"""
import collections
import csv
import sys

frames = 32
paths = []
for a in sys.argv[1:]:
    if a.startswith("--frames="):
        frames = int(a.split("=", 1)[1])
    else:
        paths.append(a)
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for path in paths:
    for r in csv.DictReader(open(path)):
        tab[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in tab.values() for c in k})
# a pass that failed silently would leave a summary without its counters, which bench.py would then read as measured
# traffic: every kernel must carry all three groups
need = ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE")
missing = {k: [c for c in need if c not in v] for k, v in tab.items() if not k.startswith("__amd") and any(c not in v for c in need)}
if missing or not tab:
    sys.stderr.write("summarize_pmc: counter groups missing (a rocprofv3 --pmc pass failed?): %s\n" % (missing or "no input rows"))
    sys.exit(1)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches"] + counters + ["hbm_bytes_per_dispatch", "mfma_util", "frames_per_dispatch"])
for k, v in tab.items():
    if k.startswith("__amd"):
        continue
    mean = {c: sum(x) / len(x) for c, x in v.items()}
    n = max(len(x) for x in v.values())
    hbm = (2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean else ""
    util = ""
    if mean.get("GRBM_GUI_ACTIVE"):
        util = round(mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * mean["GRBM_GUI_ACTIVE"] / 8), 4)
    w.writerow([k, n] + [round(mean.get(c, float("nan")), 1) for c in counters] + [hbm and round(hbm), util, frames])
