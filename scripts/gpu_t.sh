for i in 1 2 3; do
for v in base plain; do
  if [ $v = base ]; then unset SWHIP_LIBRARY; else export SWHIP_LIBRARY=$PWD/build/libswhip_$v.so; fi
  echo -n "$v: "; python bench.py --no-cpu --steps 30 --warmup 3 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done; done
