#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into small JSON/CSV files for profiles/.

    python tools/pmc_summary.py <dir-with-rocprofv3-csv-output>... > summary.json

Reads every *_counter_collection.csv (PMC passes) and *_kernel_trace.csv / *_kernel_stats.csv found
below the given directories; per kernel: launches, mean duration, mean counter value per launch.
FETCH_SIZE / WRITE_SIZE are reported in KB as rocprofv3 prints them, plus `*_bytes_corrected` following
MI355X_MICROARCH.md (HBM section): FETCH_SIZE on gfx950 tallies 128-B requests at 64 B -> x2 for wide
coalesced reads; WRITE_SIZE is exact for streaming stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


# FETCH_SIZE calibration factor per kernel (default 2).  Rounds 1-2 used 1.0 for k_fast_cells (a stand-in stream with the FAST
# tile shape read 1.12 x its unique bytes uncorrected); round 3 showed that reading to be wrong: with the workgroups of a frame
# kept on one XCD (orb.hip xcd_frame_block) the REAL launches read 0.59 x their unique bytes uncorrected -- impossible -- and
# 1.19 x with the guide's x2 (profiles/r03c_fetch_calibration.json).  The stand-in, like the kernel, had been re-reading every
# tile halo / shared 128-B line once per XCD.
CAL = {}


def short(name):
    n = name.split("(")[0]
    return n.replace("sd::", "")


def main():
    out = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                c = r["Counter_Name"]
                out[k][c] += float(r["Counter_Value"])
                cnt[k][c] += 1
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                out[k]["duration_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                cnt[k]["duration_ns"] += 1
    res = {}
    for k in sorted(out):
        e = {}
        for c in sorted(out[k]):
            # a counter row appears once per dispatch (per dimension instance rows are summed by rocprofv3 csv)
            e[c + "_per_launch"] = out[k][c] / max(cnt[k][c], 1)
            e[c + "_samples"] = cnt[k][c]
        if "FETCH_SIZE_per_launch" in e:
            # rocprofv3's FETCH_SIZE tallies every EA read request at 64 B (its 128-B term, TCC_BUBBLE, stays 0 on gfx950):
            # x2 (MI355X_MICROARCH.md; re-measured for dwordx4 and plain dword streams, profiles/r02_fetch_calibration.json, and
            # on the real FAST / pyramid launches, profiles/r03c_fetch_calibration.json).  Raw and x2 are both reported.
            raw = e["FETCH_SIZE_per_launch"] * 1024
            e["FETCH_bytes_raw_per_launch"] = raw
            e["FETCH_bytes_x2_per_launch"] = raw * 2
            e["FETCH_bytes_corrected_per_launch"] = raw * CAL.get(k.replace("void ", "").split("<")[0], 2.0)
        if "WRITE_SIZE_per_launch" in e:
            e["WRITE_bytes_per_launch"] = e["WRITE_SIZE_per_launch"] * 1024
        res[k] = e
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
