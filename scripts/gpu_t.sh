for i in 1 2 3; do python bench.py --no-cpu --steps 20 --warmup 3 2>&1 | grep -o '"value": [0-9.]*\|"placement_trials_ms": [^]]*]' | tr '\n' ' '; echo; done
python bench.py --no-cpu --steps 20 --warmup 3 --placement-trials 1 2>&1 | grep -o '"value": [0-9.]*'
python bench.py --no-cpu --cols 65536 --rows 65536 --h64 --placement-trials 3 --steps 4 --warmup 1 2>&1 | grep -o '"value": [0-9.]*\|"placement_trials_ms": [^]]*]' | tr '\n' ' '
