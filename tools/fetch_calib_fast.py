#!/usr/bin/env python3
"""FETCH_SIZE of the REAL k_fast_cells launches against the unique bytes their tiles cover (VERDICT r2 task 4: calibrate on
the real kernel instead of a stand-in stream).

    python tools/fetch_calib_fast.py <dir with the rocprofv3 --pmc FETCH_SIZE pass of bench.py> [frames per launch]

Unique bytes per frame of a launch = the union, over the launch's cells, of the dword-aligned tile every workgroup stages
(zone + 3-px halo, rows zy0 - 3 .. zy0 + zh + 3; computed from sd_orb_plan_info, i.e. from the same plan the kernel runs on).
Level 0 reads the caller's frames (640-byte rows), the other levels the padded pyramid.  The launches of one step are told
apart by their grid size (cells x frames).  At 1024 frames the level-0 launch covers 1024 x 0.27 MB = 280 MB of frames --
beyond the 256 MiB Infinity Cache -- but the SAME frames are read by the pyramid's level-0 copy at the same time, and levels
1+ were written moments before: a ratio below 1 means hits in the Infinity Cache, above 1 re-reads / partial-line overfetch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdslam_amd  # noqa: E402

d = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
CFG, W, H = (1000, 1.2, 8, 20), 640, 480
info = sdslam_amd.plan_info(*CFG, W, H)
merge_from = sdslam_amd.get_option("extract.fast_merge_from")
launches = {("level0 (frames, direct)" if l == 0 else f"level{l}"): [l] for l in range(min(merge_from, CFG[2]))}
if merge_from < CFG[2]:
    launches[f"levels{merge_from}-{CFG[2] - 1} (merged)"] = list(range(merge_from, CFG[2]))
uniq, grid = {}, {}
for name, levels in launches.items():
    total, ncells = 0, 0
    for l in levels:
        lw, lh = int(info["levels"][l, 0]), int(info["levels"][l, 1])
        edge = 0 if l == 0 else 19
        pw, ph = (W, H) if l == 0 else (lw + 38, lh + 38)
        mask = np.zeros((ph, (pw + 3) // 4 * 4), bool)
        for c in info["cells"][info["cells"][:, 0] == l]:
            _, zx0, zy0, zw, zh, ev = (int(v) for v in c)
            if zw <= 0 or zh <= 0:
                continue
            ncells += 1
            xs = zx0 - 3 + edge
            xa = xs & ~3
            tp = ((xs - xa + zw + 6 + 3) >> 2) * 4
            y0 = zy0 - 3 + edge
            mask[y0:y0 + zh + 6, xa:xa + tp] = True
        total += int(mask.sum())
        ncells += int(((info["cells"][:, 0] == l) & (info["cells"][:, 3] <= 0)).sum())
    uniq[name] = total
    grid[name] = ncells
# the launches of a step come in a fixed order (level 0 straight from the frames, level 1, level 2, the merged small
# levels: orb.hip pipeline_body); levels 0 and 1 have the same grid size, so position in dispatch order tells them apart
disp = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fast_cells" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            disp.append((int(r["Dispatch_Id"]), int(r["Grid_Size"]), float(r["Counter_Value"]) * 1024.0))
disp.sort()
rows = defaultdict(list)
names = list(launches)
for k, (_, g, v) in enumerate(disp):
    name = names[k % len(names)]
    if g == grid[name] * 256 * B:      # Grid_Size is in work-items: cells x 256 threads x frames
        rows[name].append(v)
out = {}
for name in launches:
    g = grid[name] * 256 * B
    v = rows.get(name, [])
    if not v:
        out[name] = {"unique_bytes_per_launch": uniq[name] * B, "launches_seen": 0, "grid_size_expected": g,
                     "grid_sizes_present": sorted({x[1] for x in disp})}
        continue
    raw = float(np.mean(v))
    out[name] = {"unique_bytes_per_launch": uniq[name] * B, "launches_seen": len(v), "FETCH_SIZE_bytes_raw": raw,
                 "raw_over_unique": raw / (uniq[name] * B), "x2_over_unique": 2 * raw / (uniq[name] * B)}
tot_raw = sum(o.get("FETCH_SIZE_bytes_raw", 0) for o in out.values())
tot_u = sum(o["unique_bytes_per_launch"] for o in out.values())
out["all FAST launches of a step"] = {"unique_bytes": tot_u, "FETCH_SIZE_bytes_raw": tot_raw, "raw_over_unique": tot_raw / tot_u, "x2_over_unique": 2 * tot_raw / tot_u}
# ---- the pyramid (k_pyr_split, one launch per level in level order): unique bytes read by launch l = the interior of its
# source (the caller's frame for level 0, level l-1 of the padded pyramid otherwise); every border pixel is computed from the same
# source rows / columns.  Written: the padded level (64-B aligned rows).
pdisp = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pyr_split" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            pdisp.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024.0))
pdisp.sort()
NL = CFG[2]
prow = defaultdict(list)
for k, (_, v) in enumerate(pdisp):
    prow[k % NL].append(v)
ptot_raw = ptot_u = 0
for l in range(NL):
    sw, sh = (W, H) if l == 0 else (int(info["levels"][l - 1, 0]), int(info["levels"][l - 1, 1]))
    u = sw * sh * B
    if prow[l]:
        raw = float(np.mean(prow[l]))
        out[f"pyramid level {l} (k_pyr_split)"] = {"unique_source_bytes_per_launch": u, "launches_seen": len(prow[l]), "FETCH_SIZE_bytes_raw": raw,
                                                   "raw_over_unique": raw / u, "x2_over_unique": 2 * raw / u}
        ptot_raw += raw
        ptot_u += u
if ptot_u:
    out["all pyramid launches of a step"] = {"unique_source_bytes": ptot_u, "FETCH_SIZE_bytes_raw": ptot_raw, "raw_over_unique": ptot_raw / ptot_u,
                                             "x2_over_unique": 2 * ptot_raw / ptot_u}
json.dump(out, sys.stdout, indent=1)
print()
