for i in 1 2 3; do python bench.py --no-cpu --steps 20 --warmup 3 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['placement_trials_ms']; print('%.1f GCUPS  [%s]'%(d['value'],' '.join('%.2f'%x for x in t)))
"; done
python bench.py --no-cpu --steps 20 --warmup 3 --placement-trials 1 2>&1 | grep -o '"value": [0-9.]*'
python bench.py --no-cpu --steps 10 --warmup 2 --cols 8192 --rows 8192 2>&1 | grep -o '"value": [0-9.]*\|"placement_trials_ms": [^]]*]' | tr '\n' ' '; echo
python bench.py --no-cpu --steps 5 --warmup 2 --cols 32768 --rows 32768 --placement-trials 3 2>&1 | grep -o '"value": [0-9.]*\|"placement_trials_ms": [^]]*]' | tr '\n' ' '; echo
