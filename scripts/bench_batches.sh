#!/bin/bash
# img/s against the per-GPU batch: shows how much of the step depends on a layer's working set fitting the Infinity Cache
for b in "$@"; do
  python bench.py --batch $b --no-cpu-baseline --no-kernel-profile 2>/dev/null > /tmp/bb.json
  python -c "import json; d=json.load(open('/tmp/bb.json')); print('batch', $b, round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms')"
done
