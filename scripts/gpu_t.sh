python scripts/strip_times.py 16384 16384 0 1 8 2>&1 | grep -E "^   0:|hops|last end|export->|strips mean"
python scripts/strip_times.py 16384 16384 0 2 4 2>&1 | grep -E "^   0:|hops|last end|export->|strips mean"
python scripts/strip_times.py 16384 16384 2 1 8 2>&1 | grep -E "^   0:|hops|last end"
