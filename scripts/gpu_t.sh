run() { echo -n "$*: "; python bench.py --no-cpu "$@" 2>&1 | grep -o '"value": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' '; echo; }
run --steps 20 --warmup 3
run --steps 20 --warmup 3 --p8
run --cols 32768 --rows 32768 --steps 5 --warmup 1 --placement-trials 2
run --cols 32768 --rows 32768 --steps 5 --warmup 1 --placement-trials 2 --p8
run --cols 65536 --rows 65536 --steps 3 --warmup 1 --placement-trials 1
run --cols 65536 --rows 65536 --steps 3 --warmup 1 --placement-trials 1 --p8
run --cols 65536 --rows 65536 --steps 3 --warmup 1 --placement-trials 1 --h64
run --cols 65536 --rows 65536 --steps 3 --warmup 1 --placement-trials 1 --h64 --p8
