#!/bin/bash
# Copy the summaries of a tools/run_profiles.sh + tools/final_bench.sh pair from gpurun_out/ into profiles/ and drop the previous pair.
#   tools/refresh_profiles.sh NEWPROF NEWBENCH OLDPROF OLDBENCH      e.g. r03l r03m r03j r03k
set -e
NP=$1; NB=$2; OP=$3; OB=$4
S=gpurun_out/prof_$NP
cp $S/trace/t_kernel_stats.csv profiles/${NP}_kernel_stats_full_b1024.csv
cp $S/trace_serialized/t_kernel_stats.csv profiles/${NP}_kernel_stats_serialized_b1024.csv
cp $S/pmc_summary.json profiles/${NP}_pmc_summary_b1024.json
cp $S/fetch_calibration_fast.json profiles/${NP}_fetch_calibration.json
grep '^{' $S/trace.log | tail -1 > profiles/${NP}_bench_under_kernel_trace.json
grep '^{' $S/trace_serialized.log | tail -1 > profiles/${NP}_bench_serialized.json
python tools/timeline.py $S/trace/t_kernel_trace.csv > profiles/${NP}_timeline_one_step.txt 2>/dev/null || true
for f in gpurun_out/${NB}_bench_*.json; do grep '^{' $f | tail -1 > profiles/$(basename $f); done
grep '^{' gpurun_out/${NB}_bench_reloc.txt | tail -1 > profiles/${NB}_bench_reloc.json || true
git rm -q --cached profiles/${OP}_* profiles/${OB}_* 2>/dev/null || true
rm -f profiles/${OP}_* profiles/${OB}_*
ls profiles | grep "${NP}_\|${NB}_"
