#!/bin/bash
# sweep strips-per-group / consumers-per-strip for the headline workload
for cfg in "1 6" "1 8" "2 6" "2 5"; do
  set -- $cfg
  echo "== ns=$1 nc=$2"
  python bench.py --no-cpu --steps 20 --warmup 3 --ns $1 --nc $2 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
  python scripts/strip_times.py 16384 16384 0 $1 $2 2>&1 | grep -E "^   0:|hops|last end"
done
