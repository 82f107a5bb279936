timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
echo -n "16384 i32: "; python bench.py --no-cpu --steps 20 --warmup 3 2>&1 | grep -o '"value": [0-9.]*'
echo -n "32768 i64 auto: "; python bench.py --no-cpu --cols 32768 --rows 32768 --h64 --steps 10 --warmup 2 2>&1 | grep -o '"value": [0-9.]*'
echo -n "32768 i64 nt: "; python bench.py --no-cpu --cols 32768 --rows 32768 --h64 --steps 10 --warmup 2 --store-policy 2 2>&1 | grep -o '"value": [0-9.]*'
echo -n "65536 i64 auto: "; python bench.py --no-cpu --cols 65536 --rows 65536 --h64 --steps 5 --warmup 1 2>&1 | grep -o '"value": [0-9.]*'
echo -n "65536 i64 nt: "; python bench.py --no-cpu --cols 65536 --rows 65536 --h64 --steps 5 --warmup 1 --store-policy 2 2>&1 | grep -o '"value": [0-9.]*'
echo -n "65536 i32 auto: "; python bench.py --no-cpu --cols 65536 --rows 65536 --steps 5 --warmup 1 2>&1 | grep -o '"value": [0-9.]*'
echo -n "batch: "; python bench.py --no-cpu --mode batch --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-300
