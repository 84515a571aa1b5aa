#!/bin/bash
# One bench run per option setting (full step): tools/sweep_options.sh "opt=val" "opt=val opt2=val" ...   ("" = defaults)
set -e
for o in "$@"; do
  args=""; for kv in $o; do args="$args --option $kv"; done
  python bench.py --no-cpu-baseline --no-extras --steps 100 $args > /tmp/o.json
  python -c "
import json;d=json.load(open('/tmp/o.json'));print('[$o]', round(d['value']/1e3,1), {k:round(v,2) for k,v in d['stages_ms_per_step'].items()})"
done
