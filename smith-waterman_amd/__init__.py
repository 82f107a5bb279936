"""smith-waterman_amd -- host-side mirror (ctypes) of the C-ABI in include/swhip.h.

The product is libswhip.so (hand-written gfx950 HIP kernels + a C-ABI); this module is plumbing
for tests / bench / Python callers: torch provides device memory and streams only.  There is NO
CPU fallback: if libswhip.so is missing, importing the library handle raises.

Names follow the reference (paths relative to the reference repository):
  generate()            serial_smithW.c:334-361
  Engine.fill()         fill loop + similarityScore, serial_smithW.c:141-145,187-256;
                        smithWaterman(a,b,w,h,H,P,&maxloc), rotated-cuda/sw-rotated-omp.cc:192-209
  Engine.traceback()    backtrack(), serial_smithW.c:262-277
  n_element / first_diag_element    omp_smithW.c:260-291
"""
from __future__ import annotations

import ctypes
import os
import weakref
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SWHIP_LIBRARY") or os.path.join(_HERE, "libswhip.so")  # override: A/B builds of the kernels

NONE, UP, LEFT, DIAGONAL, PATH = 0, 1, 2, 3, -1
DEFAULT_SCORES = (3, -3, -2)  # serial_smithW.c:59-61


class SwError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"swhip error {code}: {msg}")
        self.code = code


class _Scores(ctypes.Structure):
    _fields_ = [("match", ctypes.c_int32), ("mismatch", ctypes.c_int32), ("gap", ctypes.c_int32)]


class _Result(ctypes.Structure):
    _fields_ = [("max_pos", ctypes.c_int64), ("max_score", ctypes.c_int64), ("path_len", ctypes.c_int64)]


# every symbol include/swhip.h declares: (restype, argtypes)
_vp, _i64, _i32, _u32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t
ABI = {
    "sw_last_error": (ctypes.c_char_p, []),
    "sw_version": (ctypes.c_char_p, []),
    "sw_generate": (_i32, [_i64, _i64, _u32, _vp, _vp]),
    "sw_read_fasta": (_i32, [ctypes.c_char_p, _i64, _vp, _i64, ctypes.POINTER(_i64)]),
    "sw_nelement": (_i64, [_i64, _i64, _i64]),
    "sw_first_diag_element": (None, [_i64, _i64, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "sw_create": (_i32, [_i32, ctypes.POINTER(_vp)]),
    "sw_destroy": (None, [_vp]),
    "sw_fill_device": (_i32, [_vp, _vp, _i64, _vp, _i64, ctypes.POINTER(_Scores), _vp, _i32, _vp, _vp, _vp, _vp]),
    "sw_fill_tile_device": (_i32, [_vp, _vp, _i64, _vp, _i64, ctypes.POINTER(_Scores), _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "sw_batch_device": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, ctypes.POINTER(_Scores), _vp, _vp, _vp, _vp]),
    "sw_batch_device_ex": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, ctypes.POINTER(_Scores), _vp, _vp, _i32, _vp, _vp]),
    "sw_batch_traceback_device": (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "sw_fill_band_device": (_i32, [_vp, _vp, _i64, _vp, _i64, _i64, ctypes.POINTER(_Scores), _vp, _i32, _vp, _i32, _vp, _u32, _vp, _u32, _vp,
                                   _i32, _i32, _vp, _vp]),
    "sw_multi_create": (_i32, [ctypes.POINTER(_i32), _i32, _vp, _i64, _vp, _i64, _i32, _i32, ctypes.POINTER(_vp)]),
    "sw_multi_fill": (_i32, [_vp, ctypes.POINTER(_Scores), _i32, ctypes.POINTER(_Result)]),
    "sw_multi_traceback": (_i32, [_vp, ctypes.POINTER(_i64)]),
    "sw_multi_band_info": (_i32, [_vp, _i32, ctypes.POINTER(_i32), ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "sw_multi_nbands": (_i32, [_vp]),
    "sw_multi_seconds": (ctypes.c_double, [_vp]),
    "sw_multi_free": (None, [_vp]),
    "sw_align_auto": (_i32, [_vp, _vp, _i64, _vp, _i64, ctypes.POINTER(_Scores), _vp, _vp, ctypes.POINTER(_Result), ctypes.POINTER(_i32)]),
    "sw_align_auto_multi": (_i32, [_vp, ctypes.POINTER(_i32), _i32, _vp, _i64, _vp, _i64, ctypes.POINTER(_Scores), _vp, _vp, ctypes.POINTER(_Result),
                                   ctypes.POINTER(_i32), _i64]),
    "sw_p_to_p2_device": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _vp]),
    "sw_p2_to_p32_device": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "sw_traceback_p2_device": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp]),
    "sw_fill_cpu": (_i32, [_vp, _i64, _vp, _i64, ctypes.POINTER(_Scores), _vp, _vp, ctypes.POINTER(_Result)]),
    "sw_fill_host": (_i32, [_vp, _vp, _i64, _vp, _i64, ctypes.POINTER(_Scores), _vp, _vp, ctypes.POINTER(_Result)]),
    "sw_traceback_device": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "sw_traceback_host": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, ctypes.POINTER(_i64)]),
    "sw_traceback_host_ex": (_i32, [_vp, _i32, _i64, _i64, _i64, _vp, _i64, ctypes.POINTER(_i64)]),
    "sw_traceback_device_ex": (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "sw_fill_device_ex": (_i32, [_vp, _vp, _i64, _vp, _i64, ctypes.POINTER(_Scores), _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    "sw_row_checksums_device": (_i32, [_vp, _vp, _i32, _i64, _i64, _vp, _vp]),
    "sw_p8_to_p32_device": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "sw_alloc_outputs": (_i32, [_vp, _vp, _i64, _vp, _i64, ctypes.POINTER(_Scores), _i32, _i32, _i32, ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                ctypes.POINTER(ctypes.c_float)]),
    "sw_free_outputs": (_i32, [_vp, _vp, _vp]),
    "sw_place_pair_ratio": (_i32, [_vp, _sz, _vp, _sz, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
    "sw_device_malloc": (_i32, [_vp, _sz, ctypes.POINTER(_vp)]),
    "sw_device_free": (_i32, [_vp, _vp]),
    "sw_memcpy_h2d": (_i32, [_vp, _vp, _vp, _sz]),
    "sw_memcpy_d2h": (_i32, [_vp, _vp, _vp, _sz]),
    "sw_synchronize": (_i32, [_vp, _vp]),
    "sw_set_option": (_i32, [_vp, ctypes.c_char_p, _i64]),
    "sw_get_option": (_i64, [_vp, ctypes.c_char_p]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load libswhip.so (built in-tree by `make -C smith-waterman_amd`). Fails loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `make -C {_HERE}` (hipcc --offload-arch=gfx950); "
                              "there is no CPU fallback")
        try:  # torch bundles its own HIP/HSA runtime: let it load first so the process has ONE
            import torch  # noqa: F401  (libswhip.so then binds to the already-loaded libamdhip64)
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise SwError(rc, lib().sw_last_error().decode())


def generate(cols: int, rows: int, seed: int = 1):
    """Random DNA pair exactly as the reference's generate() (seed 1 == serial_smithW.c).
    Returns (a, b) as uint8 arrays of length cols / rows."""
    a = np.zeros(cols + 1, np.uint8)
    b = np.zeros(rows + 1, np.uint8)
    _check(lib().sw_generate(cols, rows, seed, a.ctypes.data, b.ctypes.data))
    return a[:cols].copy(), b[:rows].copy()


def read_fasta(path: str, record: int = 0):
    """One record of a FASTA file as a uint8 array (upper-cased, white space dropped); see sw_read_fasta."""
    n = _i64()
    _check(lib().sw_read_fasta(os.fsencode(path), record, None, 0, ctypes.byref(n)))
    seq = np.zeros(max(1, n.value), np.uint8)
    _check(lib().sw_read_fasta(os.fsencode(path), record, seq.ctypes.data, n.value, ctypes.byref(n)))
    return seq[:n.value].copy()


def n_element(i: int, m: int, n: int) -> int:
    return int(lib().sw_nelement(i, m, n))


def first_diag_element(i: int, m: int, n: int):
    si, sj = _i64(), _i64()
    lib().sw_first_diag_element(i, m, n, ctypes.byref(si), ctypes.byref(sj))
    return int(si.value), int(sj.value)


def traceback_host(P: np.ndarray, max_pos: int):
    """backtrack() on a host int32 (or compact int8) P, modified in place. Returns the visited linear indices."""
    rows1, m = P.shape
    assert P.dtype in (np.int32, np.int8) and P.flags.c_contiguous
    path = np.zeros(rows1 + m + 2, np.int64)
    n = _i64()
    _check(lib().sw_traceback_host_ex(P.ctypes.data, P.dtype.itemsize, m - 1, rows1 - 1, int(max_pos), path.ctypes.data, len(path),
                                      ctypes.byref(n)))
    return path[: n.value].copy()


def _as_seq(x) -> np.ndarray:
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), np.uint8)
    if isinstance(x, str):
        return np.frombuffer(x.encode(), np.uint8)
    return np.ascontiguousarray(x, dtype=np.uint8)


@dataclass
class Fill:
    """Device-resident result of one DP fill. H: (rows+1, cols+1) int32|int64, P: int32|int8, both torch
    CUDA tensors (either may be None: not written); res: int64[3] tensor (max_pos, max_score, path_len)."""
    H: "torch.Tensor"
    P: "torch.Tensor"
    res: "torch.Tensor"
    cols: int
    rows: int

    def free(self):
        """Release a pair that came from Engine.alloc_outputs now (sw_free_outputs); the tensors must not be used afterwards."""
        owner = getattr(self, "_owner", None)
        self.H = self.P = None
        if owner is not None:
            owner.free()
            self._owner = None

    def result(self):
        r = self.res.cpu().tolist()
        if r[2] < 0:
            raise SwError(-62, f"in-kernel hand-off wait timed out (code {-r[2]})")
        return {"max_pos": r[0], "max_score": r[1], "path_len": r[2]}


class _RawDevice:
    """A device allocation of the library seen by torch (zero-copy, __cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}
        self._owner = owner


class _Outputs:
    """Keeps a sw_alloc_outputs pair alive; released with sw_free_outputs when the last tensor that views it is dropped (the tensors
    hold this object through _RawDevice), by Fill.free(), or at the latest by Engine.close().  The engine only keeps a WEAK
    reference: a pair whose Fill was dropped is freed by the garbage collector, not held until the engine goes."""

    def __init__(self, engine, dH, dP):
        self.engine, self.dH, self.dP = engine, dH, dP
        engine._outputs.add(self)

    def free(self):
        """sw_free_outputs, once (the pair must not outlive its context)."""
        if self.dH is None and self.dP is None:
            return
        try:
            if self.engine._h:
                lib().sw_free_outputs(self.engine._h, self.dH, self.dP)
        except Exception:
            pass
        self.dH = self.dP = None

    __del__ = free


class Engine:
    """One GPU's fill engine (sw_ctx). Uses torch only for device buffers / current stream."""

    def __init__(self, device: int = 0):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("smith-waterman_amd.Engine needs a GPU (no CPU fallback exists)")
        self.torch = torch
        self.device = device
        torch.cuda.set_device(device)
        h = _vp()
        self._outputs = weakref.WeakSet()   # live sw_alloc_outputs pairs (weak: dropping a Fill frees its pair; close() frees survivors)
        _check(lib().sw_create(device, ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            for o in list(getattr(self, "_outputs", [])):
                o.free()
            try:
                lib().sw_destroy(self._h)
            except Exception:  # interpreter shutdown: module globals are already gone
                pass
            self._h = None

    __del__ = close

    def set_option(self, name: str, value: int):
        _check(lib().sw_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        return int(lib().sw_get_option(self._h, name.encode()))

    def _stream(self):
        return _vp(self.torch.cuda.current_stream(self.device).cuda_stream)

    def to_device(self, seq):
        """Sequence -> padded uint8 device tensor (16 spare bytes; the kernel reads b in 16-B windows)."""
        t = self.torch
        s = _as_seq(seq)
        d = t.zeros(len(s) + 16, dtype=t.uint8, device=f"cuda:{self.device}")
        if len(s):
            d[: len(s)] = t.from_numpy(s.copy())
        return d, len(s)

    def alloc(self, cols: int, rows: int, h_dtype=None, p_dtype=None, want_h: bool = True, want_p: bool = True):
        """Plain output buffers (torch allocations): H int32|int64, P int32 (the reference layout) or int8 (compact P, same
        codes); either may be left out (matrix-less fills).  For the placement that makes a big fill fast see alloc_outputs."""
        t = self.torch
        h_dtype = h_dtype or t.int32
        p_dtype = p_dtype or t.int32
        assert p_dtype in (t.int32, t.int8)
        dev = f"cuda:{self.device}"
        H = t.empty((rows + 1, cols + 1), dtype=h_dtype, device=dev) if want_h else None
        P = t.empty((rows + 1, cols + 1), dtype=p_dtype, device=dev) if want_p else None
        return Fill(H, P, t.zeros(3, dtype=t.int64, device=dev), cols, rows)

    def alloc_outputs(self, d_a, d_b, cols: int, rows: int, h_dtype=None, p_dtype=None, trials: int = 0, scores=DEFAULT_SCORES):
        """Output buffers through the C-ABI allocator sw_alloc_outputs (what a C caller gets): candidate placements
        tried with real fills of this problem, the fastest kept.  Returns (Fill, [ms of every candidate tried])."""
        t = self.torch
        h_dtype = h_dtype or t.int32
        p_dtype = p_dtype or t.int32
        n = trials if trials > 0 else 16
        ms = (ctypes.c_float * n)()
        dH, dP = _vp(), _vp()
        sc = _Scores(*scores)
        self.synchronize()
        _check(lib().sw_alloc_outputs(self._h, d_a.data_ptr(), cols, d_b.data_ptr(), rows, ctypes.byref(sc), 8 if h_dtype == t.int64 else 4,
                                      1 if p_dtype == t.int8 else 4, trials, ctypes.byref(dH), ctypes.byref(dP), ms))
        owner = _Outputs(self, dH.value, dP.value)
        shape = (rows + 1, cols + 1)
        H = t.as_tensor(_RawDevice(dH.value, shape, "<i8" if h_dtype == t.int64 else "<i4", owner), device=f"cuda:{self.device}")
        P = t.as_tensor(_RawDevice(dP.value, shape, "|i1" if p_dtype == t.int8 else "<i4", owner), device=f"cuda:{self.device}")
        out = Fill(H, P, t.zeros(3, dtype=t.int64, device=f"cuda:{self.device}"), cols, rows)
        out._owner = owner
        return out, [x for x in ms if x > 0]

    def fill_into(self, out: Fill, d_a, d_b, scores=DEFAULT_SCORES, top=None):
        """Asynchronous fill on torch's current stream into pre-allocated buffers."""
        t = self.torch
        sc = _Scores(*scores)
        hb = 8 if (out.H is not None and out.H.dtype == t.int64) else 4
        _check(lib().sw_fill_device_ex(self._h, d_a.data_ptr(), out.cols, d_b.data_ptr(), out.rows, ctypes.byref(sc),
                                       out.H.data_ptr() if out.H is not None else None, hb,
                                       out.P.data_ptr() if out.P is not None else None, out.P.element_size() if out.P is not None else 4,
                                       top.data_ptr() if top is not None else None, out.res.data_ptr(), self._stream()))
        return out

    def fill(self, a, b, scores=DEFAULT_SCORES, h_dtype=None, top=None, p_dtype=None, want_h: bool = True, want_p: bool = True) -> Fill:
        d_a, cols = self.to_device(a)
        d_b, rows = self.to_device(b)
        out = self.alloc(cols, rows, h_dtype, p_dtype, want_h=want_h, want_p=want_p)
        if top is not None:
            top = self.torch.as_tensor(np.ascontiguousarray(top, np.int32)).to(f"cuda:{self.device}")
        self.fill_into(out, d_a, d_b, scores, top)
        self.synchronize()
        return out

    def fill_tile(self, H, P, i0: int, j0: int, trows: int, tcols: int, d_a, d_b, res, scores=DEFAULT_SCORES,
                  top=None, left=None, right=None):
        """Asynchronous fill of the tile rows i0+1..i0+trows x cols j0+1..j0+tcols of the matrices H, P
        (torch CUDA tensors with row stride H.shape[1]).  d_a / d_b: the FULL padded device sequences;
        top: int32 tensor with tcols+1 H values of row i0 (or None), left: trows+1 values of column j0,
        right: output tensor (trows+1) for column j0+tcols.  i0 must be a multiple of 16."""
        t = self.torch
        assert i0 % 16 == 0, "tile rows must start at a multiple of 16 (16-byte aligned b window)"
        sc = _Scores(*scores)
        stride = H.shape[1]
        hb = 8 if H.dtype == t.int64 else 4
        corner = i0 * stride + j0
        _check(lib().sw_fill_tile_device(
            self._h, d_a.data_ptr() + j0, tcols, d_b.data_ptr() + i0, trows, ctypes.byref(sc),
            H.data_ptr() + corner * hb, hb, P.data_ptr() + corner * 4, stride,
            top.data_ptr() if top is not None else None, left.data_ptr() if left is not None else None,
            right.data_ptr() if right is not None else None, res.data_ptr(), self._stream()))

    def fill_band(self, d_a, cols, d_b, rows, total_rows, H, P, res, top_gran=None, top_tag=0, bot_gran=None, bot_tag=0, bot_done=None,
                  reserve_cus=0, concurrent=False, scores=DEFAULT_SCORES):
        """Asynchronous band-resident launch (sw_fill_band_device): H/P (rows+1, cols+1) band-local tensors or None;
        top_gran / bot_gran: int64 tensors with cols+1 granules (tag << 32 | H); bot_done: int32 tensor, one per strip
        (device, or pinned host memory)."""
        t = self.torch
        sc = _Scores(*scores)
        hb = 8 if (H is not None and H.dtype == t.int64) else 4
        ptr = lambda x: x.data_ptr() if x is not None else None
        _check(lib().sw_fill_band_device(self._h, d_a.data_ptr(), cols, d_b.data_ptr(), rows, total_rows, ctypes.byref(sc), ptr(H), hb,
                                         ptr(P), P.element_size() if P is not None else 4, ptr(top_gran), top_tag, ptr(bot_gran), bot_tag,
                                         ptr(bot_done), reserve_cus, 1 if concurrent else 0, res.data_ptr(), self._stream()))

    def batch(self, a_all, b_all, scores=DEFAULT_SCORES, store: bool = False, p_dtype=None, store_h=None, traceback: bool = False,
              want_paths: bool = False):
        """npairs independent problems (BASELINE config 5).  a_all: (npairs, cols) uint8, b_all: (npairs, rows).
        store: write P (int32, or int8 with p_dtype=torch.int8) and -- unless store_h is False -- H of every pair.
        traceback: also run backtrack() per pair (needs P); results[:, 2] then holds the path lengths.
        Returns (results[npairs,3] int64 tensor, H, P) -- plus the paths tensor (npairs, cols+rows+2) when want_paths."""
        d_a, d_b, cols, rows = self.batch_to_device(a_all, b_all)
        return self.batch_device(d_a, d_b, cols, rows, scores, store, p_dtype, store_h, traceback, want_paths)

    def batch_to_device(self, a_all, b_all):
        """Host sequences of a batch -> device tensors in the layout sw_batch_device wants (rows of b padded to 16 bytes)."""
        t = self.torch
        a_all = np.ascontiguousarray(a_all, np.uint8)
        b_all = np.ascontiguousarray(b_all, np.uint8)
        npairs, cols = a_all.shape
        rows = b_all.shape[1]
        dev = f"cuda:{self.device}"
        bstr = (rows + 15) // 16 * 16
        d_a = t.from_numpy(a_all).to(dev)
        bpad = np.zeros((npairs, bstr), np.uint8)
        bpad[:, :rows] = b_all
        return d_a, t.from_numpy(bpad).to(dev), cols, rows

    def batch_device(self, d_a, d_b, cols: int, rows: int, scores=DEFAULT_SCORES, store: bool = False, p_dtype=None, store_h=None,
                     traceback: bool = False, want_paths: bool = False, out=None):
        """sw_batch_device_ex (+ sw_batch_traceback_device) on sequences already resident in HBM (batch_to_device).  `out`:
        (res, H, P) tensors of an earlier call to write into again."""
        t = self.torch
        npairs = d_a.shape[0]
        dev = f"cuda:{self.device}"
        if out is not None:
            res, H, P = out
        else:
            res = t.zeros((npairs, 3), dtype=t.int64, device=dev)
            H = P = None
            if store:
                if store_h is None or store_h:
                    H = t.empty((npairs, rows + 1, cols + 1), dtype=t.int32, device=dev)
                P = t.empty((npairs, rows + 1, cols + 1), dtype=p_dtype or t.int32, device=dev)
        sc = _Scores(*scores)
        _check(lib().sw_batch_device_ex(self._h, d_a.data_ptr(), d_a.shape[1], cols, d_b.data_ptr(), d_b.shape[1], rows, npairs, ctypes.byref(sc),
                                        H.data_ptr() if H is not None else None, P.data_ptr() if P is not None else None,
                                        P.element_size() if P is not None else 4, res.data_ptr(), self._stream()))
        paths = None
        if traceback:
            assert P is not None, "the traceback walks P"
            cap = cols + rows + 2
            if want_paths:
                paths = t.zeros((npairs, cap), dtype=t.int64, device=dev)
            _check(lib().sw_batch_traceback_device(self._h, P.data_ptr(), P.element_size(), cols, rows, npairs,
                                                   paths.data_ptr() if paths is not None else None, cap, res.data_ptr(), self._stream()))
        self.synchronize()
        if bool((res[:, 2] < 0).any().item()):
            raise SwError(-62, "in-kernel hand-off wait timed out")
        return (res, H, P, paths) if want_paths else (res, H, P)

    def traceback(self, out: Fill, max_pos: int | None = None, want_path: bool = True):
        """backtrack() on the device P (negates the path in place). Returns the path indices."""
        t = self.torch
        if max_pos is None:
            max_pos = out.result()["max_pos"]
        cap = out.cols + out.rows + 2
        path = t.zeros(cap if want_path else 1, dtype=t.int64, device=out.P.device)
        _check(lib().sw_traceback_device_ex(self._h, out.P.data_ptr(), out.P.element_size(), out.cols, out.rows, int(max_pos),
                                            path.data_ptr() if want_path else None, cap, out.res.data_ptr(), self._stream()))
        self.synchronize()
        n = int(out.res[2].item())
        return path[:n].cpu().numpy() if want_path else n

    def row_checksums(self, X):
        t = self.torch
        cs = t.zeros(X.shape[0], dtype=t.int64, device=X.device)
        _check(lib().sw_row_checksums_device(self._h, X.data_ptr(), X.element_size(), X.shape[0], X.shape[1],
                                             cs.data_ptr(), self._stream()))
        self.synchronize()
        return cs.cpu().numpy().view(np.uint64)

    def widen_p(self, P8):
        """int8 predecessor matrix -> the reference's int32 layout (sw_p8_to_p32_device)."""
        t = self.torch
        out = t.empty(P8.shape, dtype=t.int32, device=P8.device)
        _check(lib().sw_p8_to_p32_device(self._h, P8.data_ptr(), out.data_ptr(), P8.numel(), self._stream()))
        return out

    def pack_p2(self, P, want_bits: bool = True):
        """int8 / int32 predecessor matrix -> the 2-bit format (sw_p_to_p2_device): (P2 uint8 tensor with 4 cells per byte, path bitmap
        int32 tensor with 1 bit per cell -- set where P was negative, i.e. on a traced path -- or None)."""
        t = self.torch
        n = P.numel()
        P2 = t.empty(((n + 31) // 32) * 8, dtype=t.uint8, device=P.device)
        bits = t.empty((n + 31) // 32, dtype=t.int32, device=P.device) if want_bits else None
        _check(lib().sw_p_to_p2_device(self._h, P.data_ptr(), P.element_size(), P2.data_ptr(), bits.data_ptr() if want_bits else None, n, self._stream()))
        return P2, bits

    def unpack_p2(self, P2, bits, shape):
        """2-bit matrix (+ optional path bitmap) -> the reference's int32 layout (sw_p2_to_p32_device)."""
        t = self.torch
        out = t.empty(shape, dtype=t.int32, device=P2.device)
        _check(lib().sw_p2_to_p32_device(self._h, P2.data_ptr(), bits.data_ptr() if bits is not None else None, out.data_ptr(), out.numel(), self._stream()))
        return out

    def traceback_p2(self, P2, cols: int, rows: int, max_pos: int, bits=None, want_path: bool = True):
        """backtrack() on a 2-bit matrix: the path is marked in `bits` (zeroed by the caller) instead of negating P.  Returns the path
        indices (or the length)."""
        t = self.torch
        cap = cols + rows + 2
        path = t.zeros(cap if want_path else 1, dtype=t.int64, device=P2.device)
        res = t.zeros(3, dtype=t.int64, device=P2.device)
        _check(lib().sw_traceback_p2_device(self._h, P2.data_ptr(), cols, rows, int(max_pos), bits.data_ptr() if bits is not None else None,
                                            path.data_ptr() if want_path else None, cap, res.data_ptr(), self._stream()))
        self.synchronize()
        n = int(res[2].item())
        return path[:n].cpu().numpy() if want_path else n

    def synchronize(self):
        _check(lib().sw_synchronize(self._h, self._stream()))


def smith_waterman(a, b, scores=DEFAULT_SCORES, device: int = 0, backtrack: bool = True):
    """Whole-pipeline convenience with host arrays, shaped like the reference's program:
    returns dict(H, P, max_pos, max_score, path) with P negated along the path when backtrack."""
    eng = Engine(device)
    try:
        out = eng.fill(a, b, scores)
        r = out.result()
        path = eng.traceback(out, r["max_pos"]) if backtrack else np.zeros(0, np.int64)
        return {"H": out.H.cpu().numpy(), "P": out.P.cpu().numpy(), "max_pos": r["max_pos"],
                "max_score": r["max_score"], "path": path}
    finally:
        eng.close()


class MultiFill:
    """One matrix over several GPUs of this process (sw_multi_*): row bands, band-resident launches, peer-copied halos.
    devices may repeat an id (bands then share that GPU)."""

    def __init__(self, devices, a, b, p_dtype="int32", want_h=True):
        self.a, self.b = _as_seq(a).copy(), _as_seq(b).copy()
        self.cols, self.rows = len(self.a), len(self.b)
        self.pbytes = 1 if str(p_dtype).endswith("int8") else 4
        dv = (_i32 * len(devices))(*devices)
        h = _vp()
        _check(lib().sw_multi_create(dv, len(devices), self.a.ctypes.data, self.cols, self.b.ctypes.data, self.rows, self.pbytes, 1 if want_h else 0,
                                     ctypes.byref(h)))
        self._h = h
        self.want_h = want_h

    def fill(self, scores=DEFAULT_SCORES, nchunks=64):
        sc, r = _Scores(*scores), _Result()
        _check(lib().sw_multi_fill(self._h, ctypes.byref(sc), nchunks, ctypes.byref(r)))
        return {"max_pos": r.max_pos, "max_score": r.max_score, "seconds": lib().sw_multi_seconds(self._h)}

    def traceback(self):
        n = _i64()
        _check(lib().sw_multi_traceback(self._h, ctypes.byref(n)))
        return n.value

    def bands(self):
        """[(device, lo, hi, H (numpy or None), P (numpy int32))] copied to the host."""
        out = []
        for g in range(lib().sw_multi_nbands(self._h)):
            dev, lo, hi, dH, dP = _i32(), _i64(), _i64(), _vp(), _vp()
            _check(lib().sw_multi_band_info(self._h, g, ctypes.byref(dev), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(dH), ctypes.byref(dP)))
            shape = (hi.value - lo.value + 1, self.cols + 1)
            import torch
            with torch.cuda.device(dev.value):
                H = None
                if self.want_h:
                    H = np.zeros(shape, np.int32)
                    torch.cuda.synchronize()
                    _hip_d2h(H, dH.value)
                P = np.zeros(shape, np.int8 if self.pbytes == 1 else np.int32)
                _hip_d2h(P, dP.value)
            out.append((dev.value, lo.value, hi.value, H, P.astype(np.int32)))
        return out

    def band_tensors(self):
        """[(device, lo, hi, H, P)] as torch tensors that alias the band-local device matrices (no copy; H None when not kept).
        Row 0 of a band is its halo row (= row lo of the whole matrix)."""
        import torch
        out = []
        for g in range(lib().sw_multi_nbands(self._h)):
            dev, lo, hi, dH, dP = _i32(), _i64(), _i64(), _vp(), _vp()
            _check(lib().sw_multi_band_info(self._h, g, ctypes.byref(dev), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(dH), ctypes.byref(dP)))
            shape = (hi.value - lo.value + 1, self.cols + 1)
            H = torch.as_tensor(_RawDevice(dH.value, shape, "<i4", self), device=f"cuda:{dev.value}") if self.want_h else None
            P = torch.as_tensor(_RawDevice(dP.value, shape, "|i1" if self.pbytes == 1 else "<i4", self), device=f"cuda:{dev.value}")
            out.append((dev.value, lo.value, hi.value, H, P))
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib().sw_multi_free(self._h)
            self._h = None

    __del__ = close


def _hip_d2h(arr: np.ndarray, dptr: int):
    import torch
    n = arr.nbytes
    t = torch.as_tensor(_RawDevice(dptr, (n,), "|u1", None), device="cuda")
    arr.view(np.uint8).reshape(-1)[:] = t.cpu().numpy()


def fill_host(engine: "Engine", a, b, scores=DEFAULT_SCORES):
    """sw_fill_host: the drop-in for the reference's fill loop on HOST buffers -- what `main` owns after its two
    callocs (serial_smithW.c:96-103) goes in, H, P and maxPos come back in the reference's layout.  Returns dict with H, P, max_pos, max_score."""
    a, b = _as_seq(a).copy(), _as_seq(b).copy()
    cols, rows = len(a), len(b)
    H = np.zeros((rows + 1, cols + 1), np.int32)
    P = np.zeros((rows + 1, cols + 1), np.int32)
    sc, r = _Scores(*scores), _Result()
    _check(lib().sw_fill_host(engine._h, a.ctypes.data, cols, b.ctypes.data, rows, ctypes.byref(sc), H.ctypes.data, P.ctypes.data, ctypes.byref(r)))
    return {"H": H, "P": P, "max_pos": r.max_pos, "max_score": r.max_score}


def align_auto(a, b, scores=DEFAULT_SCORES, engine: "Engine | None" = None, devices=None, multi_min_cells: int = 0):
    """sw_align_auto: host fill for tiny problems, the GPU of `engine` otherwise; host traceback.  With `devices` (a list of GPU ids,
    ids may repeat) sw_align_auto_multi: host / one GPU / row bands over all of them, chosen by size.  Returns dict like
    smith_waterman() plus used_gpu and executor (0 host, 1 one GPU, 2 several)."""
    a, b = _as_seq(a).copy(), _as_seq(b).copy()
    cols, rows = len(a), len(b)
    H = np.zeros((rows + 1, cols + 1), np.int32)
    P = np.zeros((rows + 1, cols + 1), np.int32)
    sc, r, used = _Scores(*scores), _Result(), _i32()
    h = engine._h if engine is not None else None
    if devices is None:
        _check(lib().sw_align_auto(h, a.ctypes.data, cols, b.ctypes.data, rows, ctypes.byref(sc), H.ctypes.data, P.ctypes.data, ctypes.byref(r),
                                   ctypes.byref(used)))
        ex = 1 if used.value else 0
    else:
        dv = (_i32 * max(1, len(devices)))(*devices)
        _check(lib().sw_align_auto_multi(h, dv, len(devices), a.ctypes.data, cols, b.ctypes.data, rows, ctypes.byref(sc), H.ctypes.data, P.ctypes.data,
                                         ctypes.byref(r), ctypes.byref(used), multi_min_cells))
        ex = used.value
    return {"H": H, "P": P, "max_pos": r.max_pos, "max_score": r.max_score, "path_len": r.path_len, "used_gpu": ex > 0, "executor": ex}
