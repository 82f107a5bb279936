timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 || exit 1
for i in 1 2 3; do for v in base prev; do
  if [ $v = base ]; then unset SWHIP_LIBRARY; else export SWHIP_LIBRARY=$PWD/build/libswhip_$v.so; fi
  echo -n "$v: "; python bench.py --no-cpu --steps 20 --warmup 3 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['config']['placement_trials_ms']; print('%.1f GCUPS  min %.3f [%s]'%(d['value'],min(t),' '.join('%.2f'%x for x in t)))
"
done; done
