#!/bin/bash
# usage: tools/ab_bench.sh ENVVAR v1 v2 ...   -- A/B a runtime switch on ONE box (box-to-box spread is ~1 %)
var=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    env $var=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', round(d['value']), round(d['ms_per_step'],4))"
  done
done
