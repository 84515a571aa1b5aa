#!/bin/bash
# Kernel trace of a short bench.py run, in the pipeline and with every kernel alone (AMD_SERIALIZE_KERNEL=3): mean duration
# and launches per step of every kernel, printed side by side.   tools/trace_quick.sh TAG [bench.py args]
set -e -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tq_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-extras $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipe -o t -- python $ROOT/bench.py $ARGS > $OUT/pipe.log 2>&1
AMD_SERIALIZE_KERNEL=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alone -o t -- python $ROOT/bench.py $ARGS > $OUT/alone.log 2>&1
python - <<PY
import csv,glob,collections
def load(d):
    r=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv"%d,recursive=True):
        for x in csv.DictReader(open(f)):
            k=x["Kernel_Name"].split("(")[0].replace("sd::","").replace("void ","")
            r[k][0]+=1; r[k][1]+=float(x["End_Timestamp"])-float(x["Start_Timestamp"])
    return r
p,a=load("pipe"),load("alone")
steps=23.0
print("%-28s %8s %10s %10s %10s %10s"%("kernel","calls/st","us pipe","us alone","ms/st pipe","ms/st alone"))
tp=ta=0
for k in sorted(p,key=lambda k:-p[k][1]):
    if p[k][0]<steps*0.9: continue
    c=p[k][0]/steps; up=p[k][1]/p[k][0]/1e3; ua=a[k][1]/max(a[k][0],1)/1e3
    tp+=c*up/1e3; ta+=c*ua/1e3
    print("%-28s %8.1f %10.1f %10.1f %10.3f %10.3f"%(k[:28],c,up,ua,c*up/1e3,c*ua/1e3))
print("sum ms/step: pipe %.2f alone %.2f"%(tp,ta))
import json
for d in ("pipe","alone"):
    l=[x for x in open("$OUT/%s.log"%d) if x.startswith("{")]
    if l: j=json.loads(l[-1]); print(d,"%.1f k frames/s, %.2f ms/step"%(j["value"]/1e3,j["ms_per_step"]))
PY
