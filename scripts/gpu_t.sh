
for i in 1 2; do
for v in base prev; do
  if [ $v = base ]; then unset SWHIP_LIBRARY; else export SWHIP_LIBRARY=$PWD/build/libswhip_$v.so; fi
  echo -n "$v: "; python bench.py --no-cpu --steps 20 --warmup 3 2>&1 | grep -o '"value": [0-9.]*\|"placement_trials_ms": [^]]*]' | tr '\n' ' '; echo
done; done
unset SWHIP_LIBRARY
python scripts/strip_times.py 16384 16384 0 2 4 2>&1 | grep -E "^   0:|hops|last end"
