#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (runs only where /root/reference exists).

Runs oracle/_ref/ref_harness (the reference's own generate/similarityScore/backtrack from
/root/reference/serial_smithW.c, compiled in place by oracle/Makefile) and stores
  * small cases in full  -> tests/golden/<name>.npz  (a, b, H, P0, P1, path, meta)
  * large cases as hashes -> tests/golden/hashes.json (fnv1a64 of H / P0 / P1, maxPos, score, pathLen,
    plus row-checksum digests used by the streaming / on-device verifiers)
Fixtures are data only; no reference source text is stored.
"""
import ctypes, json, os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
OUT = os.path.join(ROOT, "tests", "golden")
CS_MUL = 0x9E3779B97F4A7C15


def fnv1a64(buf: bytes) -> int:
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.swo_fnv1a64.restype = ctypes.c_uint64
    lib.swo_fnv1a64.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    return int(lib.swo_fnv1a64(buf, len(buf)))


def row_checksums(X: np.ndarray) -> np.ndarray:
    m = X.shape[1]
    w = (np.arange(1, m + 1, dtype=np.uint64) * np.uint64(CS_MUL))
    with np.errstate(over="ignore"):
        return (X.astype(np.uint32).astype(np.uint64) * w[None, :]).sum(axis=1, dtype=np.uint64)


def run(cols, rows, seed, builtin=False):
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "g")
        args = [HARNESS, "builtin", p] if builtin else [HARNESS, str(cols), str(rows), str(seed), p]
        subprocess.run(args, check=True)
        meta = [int(x) for x in open(p + ".meta").read().split()]
        cols, rows, maxpos, score, plen = meta
        rd = lambda ext, dt: np.fromfile(p + ext, dtype=dt)
        shape = (rows + 1, cols + 1)
        return dict(a=rd(".a", np.uint8), b=rd(".b", np.uint8),
                    H=rd(".H", np.int32).reshape(shape), P0=rd(".P0", np.int32).reshape(shape),
                    P1=rd(".P1", np.int32).reshape(shape), path=rd(".path", np.int64),
                    meta=np.array([cols, rows, seed, maxpos, score, plen], dtype=np.int64))


def main():
    if not os.path.exists(HARNESS):
        sys.exit("oracle/_ref/ref_harness missing: run `make -C oracle` in the build container")
    full = {"kat_builtin": (8, 9, 1, True), "rand_1x1_s1": (1, 1, 1, False), "rand_8x9_s1": (8, 9, 1, False),
            "rand_300x200_s1": (300, 200, 1, False), "rand_256x256_s1": (256, 256, 1, False),
            "rand_65x130_s7": (65, 130, 7, False), "rand_1x77_s3": (1, 77, 3, False),
            "rand_77x1_s3": (77, 1, 3, False), "rand_129x64_s11": (129, 64, 11, False)}
    hashes = {}
    for name, (c, r, s, bi) in full.items():
        g = run(c, r, s, bi)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **g)
        hashes[name] = summarize(g)
        print(name, hashes[name])
    for (c, r, s) in [(1024, 1024, 1), (4096, 4096, 1), (1000, 3000, 5), (3000, 1000, 5), (2049, 2047, 9)]:
        name = f"rand_{c}x{r}_s{s}"
        g = run(c, r, s)
        hashes[name] = summarize(g)
        print(name, hashes[name])
    json.dump(hashes, open(os.path.join(OUT, "hashes.json"), "w"), indent=1, sort_keys=True)


def summarize(g):
    cols, rows, seed, maxpos, score, plen = [int(x) for x in g["meta"]]
    csH, csP = row_checksums(g["H"]), row_checksums(g["P0"])
    return dict(cols=cols, rows=rows, seed=seed, maxPos=maxpos, maxScore=score, pathLen=plen,
                fnvH=f"{fnv1a64(g['H'].tobytes()):016x}", fnvP0=f"{fnv1a64(g['P0'].tobytes()):016x}",
                fnvP1=f"{fnv1a64(g['P1'].tobytes()):016x}",
                fnv_csH=f"{fnv1a64(csH.tobytes()):016x}", fnv_csP=f"{fnv1a64(csP.tobytes()):016x}",
                a_head=bytes(g["a"][:32]).decode(), b_head=bytes(g["b"][:32]).decode(),
                path_min=int(g["path"].min()) if plen else -1)


if __name__ == "__main__":
    main()
