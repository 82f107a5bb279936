for i in 1 2; do for cfg in "8 2" "7 3" "7 2"; do set -- $cfg
  echo -n "nc=$1 importers=$2: "; python bench.py --no-cpu --steps 20 --warmup 3 --ns 1 --nc $1 --importers $2 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['placement_trials_ms']; print('%.1f GCUPS  min %.3f [%s]'%(d['value'],min(t),' '.join('%.2f'%x for x in t)))
"
done; done
