set -e
for o in "" "--option extract.fast_lds_whole_kb=32" "--option extract.fast_lds_whole_kb=48" "--option extract.fast_lds_whole_kb=64" "--option extract.fast_lds_kb=16" "--option extract.fast_lds_kb=32" "--option extract.fast_merge_from=2" "--option extract.fast_merge_from=4" "--option track.align_start=1" "--option track.stream_priority=1" "--option track.stream_priority=0" ""; do
  python bench.py --no-cpu-baseline --no-extras --steps 100 $o > /tmp/o.json
  python -c "
import json;d=json.load(open('/tmp/o.json'));print('$o', round(d['value']/1e3,1), {k:round(v,2) for k,v in d['stages_ms_per_step'].items()})"
done
