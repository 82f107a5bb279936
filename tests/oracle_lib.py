"""ctypes wrapper over oracle/liboracle.so (the CPU restatement, TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")


class _Scores(ctypes.Structure):
    _fields_ = [("match", ctypes.c_int32), ("mismatch", ctypes.c_int32), ("gap", ctypes.c_int32)]


def _seq(x):
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), np.uint8).copy()
    if isinstance(x, str):
        return np.frombuffer(x.encode(), np.uint8).copy()
    return np.ascontiguousarray(x, dtype=np.uint8)


class Oracle:
    def __init__(self):
        so = os.path.join(ODIR, "liboracle.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", ODIR, "liboracle.so"], check=True)
        L = ctypes.CDLL(so)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        L.swo_generate.argtypes = [i64, i64, ctypes.c_uint32, vp, vp]
        L.swo_nelement.restype = i64
        L.swo_nelement.argtypes = [i64, i64, i64]
        L.swo_first_diag_element.argtypes = [i64, i64, i64, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        for f in (L.swo_fill_rowmajor,):
            f.restype = i64
            f.argtypes = [vp, i64, vp, i64, ctypes.POINTER(_Scores), vp, vp]
        L.swo_fill_wavefront.restype = i64
        L.swo_fill_wavefront.argtypes = [vp, i64, vp, i64, ctypes.POINTER(_Scores), vp, vp, ctypes.c_int]
        L.swo_backtrack.restype = i64
        L.swo_backtrack.argtypes = [vp, i64, i64, vp, i64]
        L.swo_fnv1a64.restype = ctypes.c_uint64
        L.swo_fnv1a64.argtypes = [vp, ctypes.c_size_t]
        L.swo_row_checksums.argtypes = [vp, i64, i64, vp]
        L.swo_fill_streaming.restype = i64
        L.swo_fill_streaming.argtypes = [vp, i64, vp, i64, ctypes.POINTER(_Scores), vp, vp, ctypes.POINTER(ctypes.c_int32), vp]
        L.swo_fill_streaming_ckpt.restype = i64
        L.swo_fill_streaming_ckpt.argtypes = [vp, i64, vp, i64, ctypes.POINTER(_Scores), vp, vp, ctypes.POINTER(ctypes.c_int32), i64, vp, i64, vp, vp]
        L.swo_path_from_ckpt.restype = i64
        L.swo_path_from_ckpt.argtypes = [vp, i64, vp, i64, ctypes.POINTER(_Scores), i64, vp, i64, vp, i64, vp]
        self.L = L

    def generate(self, cols, rows, seed=1):
        a = np.zeros(cols + 1, np.uint8)
        b = np.zeros(rows + 1, np.uint8)
        self.L.swo_generate(cols, rows, seed, a.ctypes.data, b.ctypes.data)
        return a[:cols].copy(), b[:rows].copy()

    def n_element(self, i, m, n):
        return int(self.L.swo_nelement(i, m, n))

    def first_diag_element(self, i, m, n):
        si, sj = ctypes.c_int64(), ctypes.c_int64()
        self.L.swo_first_diag_element(i, m, n, ctypes.byref(si), ctypes.byref(sj))
        return int(si.value), int(sj.value)

    def fill(self, a, b, scores=(3, -3, -2), wavefront=False, threads=1):
        a, b = _seq(a), _seq(b)
        cols, rows = len(a), len(b)
        H = np.zeros((rows + 1, cols + 1), np.int32)
        P = np.zeros((rows + 1, cols + 1), np.int32)
        sc = _Scores(*scores)
        ap = np.concatenate([a, np.zeros(1, np.uint8)])
        bp = np.concatenate([b, np.zeros(1, np.uint8)])
        if wavefront:
            mp = self.L.swo_fill_wavefront(ap.ctypes.data, cols, bp.ctypes.data, rows, ctypes.byref(sc), H.ctypes.data, P.ctypes.data, threads)
        else:
            mp = self.L.swo_fill_rowmajor(ap.ctypes.data, cols, bp.ctypes.data, rows, ctypes.byref(sc), H.ctypes.data, P.ctypes.data)
        return H, P, int(mp)

    def fill_band(self, a, b, top, scores=(3, -3, -2)):
        """Rows below a given halo row `top` (H values of the row above), same recurrence."""
        a, b = _seq(a), _seq(b)
        cols, rows = len(a), len(b)
        H = np.zeros((rows + 1, cols + 1), np.int32)
        P = np.zeros((rows + 1, cols + 1), np.int32)
        H[0] = top
        sc = _Scores(*scores)
        ap = np.concatenate([a, np.zeros(1, np.uint8)])
        bp = np.concatenate([b, np.zeros(1, np.uint8)])
        self.L.swo_fill_rowmajor(ap.ctypes.data, cols, bp.ctypes.data, rows, ctypes.byref(sc), H.ctypes.data, P.ctypes.data)
        return H, P

    def backtrack(self, P, max_pos):
        rows1, m = P.shape
        path = np.zeros(rows1 + m + 2, np.int64)
        n = self.L.swo_backtrack(P.ctypes.data, m, int(max_pos), path.ctypes.data, len(path))
        return path[:n].copy()

    def fnv(self, arr):
        arr = np.ascontiguousarray(arr)
        return int(self.L.swo_fnv1a64(arr.ctypes.data, arr.nbytes))

    def row_checksums(self, X):
        X = np.ascontiguousarray(X, np.int32)
        cs = np.zeros(X.shape[0], np.uint64)
        self.L.swo_row_checksums(X.ctypes.data, X.shape[0], X.shape[1], cs.ctypes.data)
        return cs

    def fill_streaming(self, a, b, scores=(3, -3, -2)):
        a, b = _seq(a), _seq(b)
        cols, rows = len(a), len(b)
        csH = np.zeros(rows + 1, np.uint64)
        csP = np.zeros(rows + 1, np.uint64)
        bottom = np.zeros(cols + 1, np.int32)
        ms = ctypes.c_int32()
        sc = _Scores(*scores)
        mp = self.L.swo_fill_streaming(a.ctypes.data, cols, b.ctypes.data, rows, ctypes.byref(sc), csH.ctypes.data,
                                       csP.ctypes.data, ctypes.byref(ms), bottom.ctypes.data)
        return dict(csH=csH, csP=csP, max_pos=int(mp), max_score=int(ms.value), bottom=bottom)


    def fill_streaming_with_path(self, a, b, scores=(3, -3, -2), every=256, band_rows=32768):
        """Whole-matrix digests without the matrix: row checksums of H, of P before and after the traceback, the
        path itself, and the H rows at multiples of `every` (band halo rows)."""
        a, b = _seq(a), _seq(b)
        cols, rows = len(a), len(b)
        csH = np.zeros(rows + 1, np.uint64)
        csP = np.zeros(rows + 1, np.uint64)
        ckpt = np.zeros((rows // every + 1, cols + 1), np.int32)
        ms = ctypes.c_int32()
        sc = _Scores(*scores)
        nb = -(-rows // band_rows)
        band_best, band_pos = np.zeros(nb, np.int32), np.zeros(nb, np.int64)
        mp = self.L.swo_fill_streaming_ckpt(a.ctypes.data, cols, b.ctypes.data, rows, ctypes.byref(sc), csH.ctypes.data,
                                            csP.ctypes.data, ctypes.byref(ms), every, ckpt.ctypes.data, band_rows, band_best.ctypes.data,
                                            band_pos.ctypes.data)
        path = np.zeros(rows + cols + 2, np.int64)
        delta = np.zeros(rows + 1, np.uint64)
        n = self.L.swo_path_from_ckpt(a.ctypes.data, cols, b.ctypes.data, rows, ctypes.byref(sc), every, ckpt.ctypes.data,
                                      mp, path.ctypes.data, len(path), delta.ctypes.data)
        with np.errstate(over="ignore"):
            csP1 = csP + delta
        return dict(csH=csH, csP=csP, csP1=csP1, max_pos=int(mp), max_score=int(ms.value), path=path[:n].copy(),
                    ckpt=ckpt, every=every, band_best=band_best, band_pos=band_pos)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_hashes():
    import json
    return json.load(open(os.path.join(GOLDEN, "hashes.json")))
