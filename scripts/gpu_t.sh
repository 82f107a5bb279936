for n in 20480 24576 32768; do for cfg in "--ns 2 --nc 4" "--ns 1 --nc 8"; do
  echo -n "n=$n $cfg: "; python bench.py --no-cpu --cols $n --rows $n --steps 6 --warmup 1 --placement-trials 4 $cfg 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['placement_trials_ms']; print('%.1f GCUPS  [%s]'%(d['value'],' '.join('%.2f'%x for x in t)))
"
done; done
