timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
for i in 1 2; do for d in 1 0; do
  echo -n "direct=$d: "; timeout -k 10 200 python bench.py --no-cpu --steps 20 --warmup 3 --direct $d 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['placement_trials_ms']; print('%.1f GCUPS  direct=%s min %.3f [%s]'%(d['value'],d['config']['direct_handoff'],min(t),' '.join('%.2f'%x for x in t)))
"
done; done
timeout -k 10 100 python scripts/strip_times.py 16384 16384 0 1 8 2>&1 | grep -E "^   0:|hops|last end|strips mean"
