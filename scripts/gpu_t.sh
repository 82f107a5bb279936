for i in 1 2 3 4 5 6 7 8; do python bench.py --no-cpu --steps 20 --warmup 3 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['placement_trials_ms']; print('%.1f GCUPS  [%s]'%(d['value'],' '.join('%.2f'%x for x in t)))
"; done
