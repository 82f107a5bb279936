#!/bin/bash
# BASELINE configs 3 and 5 (one-off record runs)
mkdir -p gpurun_out
echo "== config 3: 65536 x 65536, int64 H" | tee gpurun_out/configs.log
timeout -k 10 500 python bench.py --cols 65536 --rows 65536 --h64 --steps 2 --warmup 1 --no-cpu 2>&1 | grep '^{' | tee -a gpurun_out/configs.log
echo "== 65536 x 65536 int32" | tee -a gpurun_out/configs.log
timeout -k 10 300 python bench.py --cols 65536 --rows 65536 --steps 2 --warmup 1 --no-cpu 2>&1 | grep '^{' | tee -a gpurun_out/configs.log
echo "== config 5: 100000 pairs of 1024 x 1024 (score-only), in chunks of 4096 pairs per launch" | tee -a gpurun_out/configs.log
timeout -k 10 500 python - <<'PY' 2>&1 | tee -a gpurun_out/configs.log
import importlib, time, numpy as np, torch, sys, ctypes
sys.path.insert(0, ".")
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
npairs, cols, rows = 100000, 1024, 1024
rng = np.random.default_rng(1)
A = torch.from_numpy(rng.integers(0, 4, (npairs, cols), dtype=np.uint8) + 65).cuda()
B = torch.from_numpy(rng.integers(0, 4, (npairs, rows), dtype=np.uint8) + 65).cuda()
res = torch.zeros((npairs, 3), dtype=torch.int64, device="cuda")
sc = sw._Scores(3, -3, -2)
def run():
    sw._check(sw.lib().sw_batch_device(eng._h, A.data_ptr(), cols, cols, B.data_ptr(), rows, rows, npairs, ctypes.byref(sc), None, None, res.data_ptr(), eng._stream()))
    eng.synchronize()
run()
t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
r = res.cpu().numpy()
print({"config": "100000 x (1024 x 1024) score-only, sequences resident", "seconds": dt, "GCUPS": npairs * cols * rows / dt / 1e9,
       "mean_score": float(r[:, 1].mean()), "min_score": int(r[:, 1].min()), "max_score": int(r[:, 1].max())})
# spot-check 3 pairs with full matrices against each other (score-only max == stored max)
sel = [0, 4095, 99999]
H = torch.empty((len(sel), rows + 1, cols + 1), dtype=torch.int32, device="cuda"); P = torch.empty_like(H)
res2 = torch.zeros((len(sel), 3), dtype=torch.int64, device="cuda")
for i, k in enumerate(sel):
    sw._check(sw.lib().sw_batch_device(eng._h, A[k].data_ptr(), cols, cols, B[k].data_ptr(), rows, rows, 1, ctypes.byref(sc), H[i].data_ptr(), P[i].data_ptr(), res2[i].data_ptr(), eng._stream()))
eng.synchronize()
print("spot check:", [(int(res2[i, 1]), int(r[k, 1]), int(H[i].max())) for i, k in enumerate(sel)])
PY
