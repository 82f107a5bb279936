for i in 1 2 3; do for cfg in "--ns 2 --nc 4" "--ns 1 --nc 8"; do
  echo -n "65536 i32 $cfg: "; python bench.py --no-cpu --cols 65536 --rows 65536 --steps 3 --warmup 1 --placement-trials 3 $cfg 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['placement_trials_ms']; print('%.1f GCUPS  [%s]'%(d['value'],' '.join('%.1f'%x for x in t)))
"
done; done
