# Classify single matrices: a strided partial-line store probe (one 4-byte store per 252 bytes, like a strip's segments)
import sys, torch, numpy as np
rows = cols = 16384
def probe(M, reps=5):
    M[:, 5::63].fill_(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): M[:, 5::63].fill_(1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
blocks, ts = [], []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 220):
    try:
        M = torch.empty((rows + 1, cols + 1), dtype=torch.int32, device="cuda")
    except RuntimeError:
        break
    blocks.append(M); ts.append(probe(M))
    if len(ts) % 16 == 0:
        print("%3d: " % (len(ts) - 16) + " ".join("%3.0f" % x for x in ts[-16:]) + "   va %x" % blocks[-16].data_ptr(), flush=True)
ts = np.array(ts)
print(f"{len(ts)} blocks: min {ts.min():.0f} us  max {ts.max():.0f} us; below 180 us: {np.sum(ts < 180)}")
