#!/usr/bin/env python3
"""Print one steady-state step of a rocprofv3 --kernel-trace CSV as a timeline (start / end / duration in us relative to
the start of a k_orient_desc launch): which kernels overlap, where the critical chain is.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python bench.py --steps 20 --no-extras --no-cpu-baseline
    python tools/timeline.py gpurun_out/tl/*/t_kernel_trace.csv [step_index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
od = [i for i, r in enumerate(rows) if "k_orient_desc" in r["Kernel_Name"]]
i0, i1 = od[k], od[k + 1]
t0 = int(rows[i0]["Start_Timestamp"])
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= int(rows[i1]["End_Timestamp"])]
sel.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in sel:
    n = r["Kernel_Name"].split("(")[0].replace("sd::", "").replace("void ", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{n:20s} q{r['Queue_Id']:>2s} {s:9.1f} {e:9.1f} {e - s:8.1f}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
