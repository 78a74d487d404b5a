#!/bin/bash
# same-box A/B of two builds of libechohip: tools/ab_lib.sh <other.so> [bench args...]; prints value / ms_per_step of interleaved bench.py runs
OTHER=$1; shift
for r in 1 2; do
  for which in new old; do
    if [ $which = old ]; then export ECHO_LIB_PATH=$OTHER; else unset ECHO_LIB_PATH; fi
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-legs --no-roofline --no-c5 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$which', round(d['value'], 2), d['unit'].split()[0], round(d['ms_per_step'], 1), 'ms/step', d['dtype'])"
  done
done
