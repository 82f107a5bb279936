#!/usr/bin/env python3
"""Whole-matrix digests of BASELINE configs 3 and 4 (65536^2 and 262144^2, seed 1) from the streaming oracle.

Run once in the build container (config 4: ~10 min on one core, 1.1 GB of checkpoints); writes tests/golden/big_digests.json.
The GPU tests compare the device's per-row checksums, arg-max, band halo rows and traced-back path against these instead of
re-running the oracle on the GPU box.  Before the big runs the checkpointed functions are checked against the materialising
oracle (itself pinned to the reference, tests/test_oracle.py) on a 3000 x 1000 case.

Per problem: maxPos, maxScore, pathLen, fnv1a64 of the path (int64 linear indices in walk order), its last index, and
fnv1a64 of the uint64 row-checksum arrays csH / csP / csP1 (P after the traceback) -- whole matrix and per 32768-row band
(rows lo+1..hi) -- plus fnv1a64 of the int32 H row at every multiple of 32768 (what one band hands to the next) and the band's own arg-max (maxPos as a
linear index of the whole matrix)."""
import json, os, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import Oracle  # noqa: E402

BAND = 32768


def digest(o, cols, rows, seed, every):
    a, b = o.generate(cols, rows, seed)
    t0 = time.time()
    st = o.fill_streaming_with_path(a, b, every=every)
    d = dict(cols=cols, rows=rows, seed=seed, maxPos=st["max_pos"], maxScore=st["max_score"], pathLen=len(st["path"]),
             fnv_path=f"{o.fnv(st['path']):016x}", path_end=int(st["path"][-1]) if len(st["path"]) else -1,
             fnv_csH=f"{o.fnv(st['csH']):016x}", fnv_csP=f"{o.fnv(st['csP']):016x}", fnv_csP1=f"{o.fnv(st['csP1']):016x}",
             bands=[])
    for lo in range(0, rows, BAND):
        hi = min(rows, lo + BAND)
        e = dict(lo=lo, hi=hi, maxScore=int(st["band_best"][lo // BAND]), maxPos=int(st["band_pos"][lo // BAND]))
        for k in ("csH", "csP", "csP1"):
            e["fnv_" + k] = f"{o.fnv(st[k][lo + 1:hi + 1]):016x}"
        if hi % every == 0:
            e["fnv_bottom_H"] = f"{o.fnv(st['ckpt'][hi // every]):016x}"
        d["bands"].append(e)
    d["oracle_seconds"] = round(time.time() - t0, 1)
    return d


def selfcheck(o):
    a, b = o.generate(3000, 1000, 5)
    H, P, mp = o.fill(a, b)
    st = o.fill_streaming_with_path(a, b, every=64)
    P1 = P.copy()
    path = o.backtrack(P1, mp)
    assert st["max_pos"] == mp and np.array_equal(st["path"], path)
    assert np.array_equal(st["csH"], o.row_checksums(H)) and np.array_equal(st["csP"], o.row_checksums(P))
    assert np.array_equal(st["csP1"], o.row_checksums(P1))
    assert np.array_equal(st["ckpt"], H[::64])
    st = o.fill_streaming_with_path(a, b, every=50, band_rows=300)
    for k in range(4):
        blk = H[300 * k + 1: min(1000, 300 * (k + 1)) + 1]
        assert st["band_best"][k] == blk.max() and st["band_pos"][k] == (300 * k + 1) * 3001 + int(np.argmax(blk))


def main():
    o = Oracle()
    selfcheck(o)
    out = os.path.join(HERE, "big_digests.json")
    res = json.load(open(out)) if os.path.exists(out) else {}
    for name, (c, r, s) in {"rand_65536x65536_s1": (65536, 65536, 1), "rand_262144x262144_s1": (262144, 262144, 1)}.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        res[name] = digest(o, c, r, s, 256)
        print(name, {k: v for k, v in res[name].items() if k != "bands"}, flush=True)
        json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
