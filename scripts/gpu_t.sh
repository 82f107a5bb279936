run() { echo -n "$*: "; python bench.py --no-cpu "$@" 2>&1 | grep -o '"value": [0-9.]*' | tr '\n' ' '; echo; }
for cfg in "--ns 2 --nc 4" "--ns 1 --nc 8"; do
run --cols 4096 --rows 4096 --steps 20 --warmup 3 $cfg
run --cols 8192 --rows 8192 --steps 20 --warmup 3 $cfg
run --steps 20 --warmup 3 $cfg
run --steps 20 --warmup 3 $cfg
run --cols 32768 --rows 32768 --steps 5 --warmup 1 --placement-trials 3 $cfg
run --cols 65536 --rows 65536 --steps 3 --warmup 1 --placement-trials 1 $cfg
run --cols 65536 --rows 65536 --steps 3 --warmup 1 --placement-trials 1 --h64 $cfg
run --mode batch --cols 1024 --rows 1024 --pairs 20000 --steps 3 --warmup 1 $cfg
run --mode batch --cols 1024 --rows 1024 --pairs 512 --store --steps 3 --warmup 1 $cfg
done
