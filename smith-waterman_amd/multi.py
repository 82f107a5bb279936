"""multi.py -- one DP matrix over several GPUs.

Decomposition: contiguous ROW BANDS, one per rank (rank g owns rows g*B+1 .. (g+1)*B, B a multiple of
16); each band is processed in COLUMN CHUNKS left to right.  Tile (band g, chunk k) needs the bottom row
of tile (g-1, k) -- received point-to-point from rank g-1 -- and the right column of tile (g, k-1), which
stays on the rank.  After a tile its bottom row segment goes to rank g+1, so the ranks form a software
pipeline: rank g starts chunk k as soon as rank g-1 has finished chunk k.  There is no data-path
collective; the only collective is one all_reduce(MAX) of the packed arg-max key at the end.
Backend: torch.distributed ("nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The tile engine is pluggable: GpuTiles drives libswhip.so (sw_fill_tile_device); the tests plug in the
CPU oracle to check the scheduler itself.  The reference has no multi-GPU code; the
result must equal serial_smithW.c on the whole matrix.
"""
from __future__ import annotations

import numpy as np

KEY_MASK = (1 << 40) - 1


def band_bounds(rows: int, world: int):
    """Rows per band: multiples of 16 (tile rows start on 16-byte aligned windows of b), last band takes the rest."""
    per = -(-rows // world)
    per = -(-per // 16) * 16
    return [(min(g * per, rows), min((g + 1) * per, rows)) for g in range(world)]


def chunk_bounds(cols: int, nchunks: int):
    per = -(-cols // nchunks)
    per = -(-per // 63) * 63   # whole strips
    out, c = [], 0
    while c < cols:
        out.append((c, min(c + per, cols)))
        c += per
    return out


class GpuTiles:
    """Tile engine on one GPU: band-local H/P (band_rows+1 x cols+1) in HBM, tiles through the C-ABI."""

    def __init__(self, engine, a, b_band, cols, band_rows, scores, h_dtype=None):
        t = engine.torch
        self.eng, self.t, self.scores = engine, t, scores
        self.d_a, _ = engine.to_device(a)
        self.d_b, _ = engine.to_device(b_band)
        dev = f"cuda:{engine.device}"
        self.H = t.zeros((band_rows + 1, cols + 1), dtype=h_dtype or t.int32, device=dev)
        self.P = t.zeros((band_rows + 1, cols + 1), dtype=t.int32, device=dev)
        self.res = t.zeros(3, dtype=t.int64, device=dev)
        self.left = t.zeros(band_rows + 1, dtype=t.int32, device=dev)
        self.right = t.zeros(band_rows + 1, dtype=t.int32, device=dev)
        self.band_rows, self.cols = band_rows, cols
        self.device = dev

    def new_row(self, n):
        return self.t.zeros(n, dtype=self.t.int32, device=self.device)

    def fill_tile(self, j0, j1, top, first_chunk):
        """top: int32 tensor (j1-j0+1) with the row above (corner first) or None for the first band."""
        if top is not None:
            self.H[0, j0:j1 + 1] = top.to(self.H.dtype)
        self.eng.fill_tile(self.H, self.P, 0, j0, self.band_rows, j1 - j0, self.d_a, self.d_b, self.res, self.scores,
                           top=top, left=None if first_chunk else self.left, right=self.right)
        self.eng.synchronize()
        self.left, self.right = self.right, self.left
        r = self.res.cpu().tolist()
        if r[2] < 0:
            from . import SwError
            raise SwError(-62, "in-kernel hand-off wait timed out")
        bottom = self.H[self.band_rows, j0:j1 + 1].to(self.t.int32)
        # tile-local arg-max -> band-local (row, col)
        row, col = divmod(r[0], self.cols + 1) if r[1] > 0 else (0, 0)
        return bottom, (r[1], row, col + j0 if r[1] > 0 else 0)

    def walk(self, pos):
        """backtrack() inside the band from band-local linear index pos; returns (visited count, last visited index)."""
        t = self.t
        cap = self.band_rows + self.cols + 2
        path = t.zeros(cap, dtype=t.int64, device=self.device)
        import ctypes
        from . import lib, _check
        _check(lib().sw_traceback_device(self.eng._h, self.P.data_ptr(), self.cols, self.band_rows, int(pos), path.data_ptr(), cap,
                                         self.res.data_ptr(), self.eng._stream()))
        self.eng.synchronize()
        n = int(self.res[2].item())
        return n, (int(path[n - 1].item()) if n else -1)

    def pred_of(self, idx):
        """where the negated cell idx points (band-local index)"""
        m = self.cols + 1
        pr = -int(self.P.view(-1)[idx].item())
        return idx - m - 1 if pr == 3 else idx - m if pr == 1 else idx - 1

    def matrices(self):
        return self.H.cpu().numpy(), self.P.cpu().numpy()


class BandPipeline:
    def __init__(self, dist, rank, world, a, b, scores=(3, -3, -2), nchunks=8, make_tiles=None):
        self.dist, self.rank, self.world = dist, rank, world
        self.a, self.b = np.asarray(a, np.uint8), np.asarray(b, np.uint8)
        self.cols, self.rows = len(self.a), len(self.b)
        self.bands = band_bounds(self.rows, world)
        self.lo, self.hi = self.bands[rank]
        self.chunks = chunk_bounds(self.cols, nchunks)
        self.active = self.hi > self.lo
        self.tiles = make_tiles(self.a, self.b[self.lo:self.hi], self.cols, self.hi - self.lo, scores) if self.active else None
        self.scores = scores

    def _prev(self):
        return self.rank - 1 if self.rank > 0 and self.bands[self.rank - 1][1] > self.bands[self.rank - 1][0] else None

    def _next(self):
        nxt = self.rank + 1
        return nxt if nxt < self.world and self.bands[nxt][1] > self.bands[nxt][0] else None

    def fill(self):
        """Fill this rank's band; returns the global (max_score, max_pos) (identical on every rank)."""
        dist = self.dist
        best = (0, 0, 0)
        pending = []
        if self.active:
            tl = self.tiles
            prev, nxt = self._prev(), self._next()
            for k, (j0, j1) in enumerate(self.chunks):
                top = None
                if prev is not None:
                    top = tl.new_row(j1 - j0 + 1)
                    dist.recv(top, src=prev)
                bottom, cand = tl.fill_tile(j0, j1, top, k == 0)
                if nxt is not None:
                    pending.append(dist.isend(bottom.contiguous(), dst=nxt))
                # lowest linear index among equal scores: earlier rows first, then earlier columns
                if cand[0] > best[0] or (cand[0] == best[0] and cand[0] > 0 and (cand[1], cand[2]) < (best[1], best[2])):
                    best = cand
            for w in pending:
                w.wait()
        # global arg-max: max score, ties -> lowest global linear index (the serial scan's rule)
        gidx = (self.lo + best[1]) * (self.cols + 1) + best[2] if best[0] > 0 else 0
        import torch
        key = torch.tensor([(best[0] << 40) | (KEY_MASK - gidx) if best[0] > 0 else 0], dtype=torch.int64)
        if dist.get_backend() == "nccl":
            key = key.cuda()
        dist.all_reduce(key, op=dist.ReduceOp.MAX)
        k = int(key.item())
        return (k >> 40, KEY_MASK - (k & KEY_MASK)) if k else (0, 0)

    def traceback(self, max_pos):
        """Distributed backtrack(): the band holding max_pos walks first, then each band above takes over at
        its last row.  Returns the total path length (identical on every rank)."""
        import torch
        dist = self.dist
        m = self.cols + 1
        grow, gcol = divmod(int(max_pos), m)
        total = 0
        state = torch.tensor([grow, gcol, 0, 1], dtype=torch.int64)  # row, col, path_len, still_walking
        for g in range(self.world - 1, -1, -1):
            lo, hi = self.bands[g]
            if self.rank == g and hi > lo and state[3].item() and lo < state[0].item() <= hi:
                pos = (int(state[0].item()) - lo) * m + int(state[1].item())
                n, last = self.tiles.walk(pos)
                if n:
                    nxt = self.tiles.pred_of(last)
                    r, c = divmod(nxt, m)
                    state = torch.tensor([lo + r, c, int(state[2].item()) + n, 1 if (r == 0 and lo > 0) else 0], dtype=torch.int64)
                else:
                    state[3] = 0
            st = state.cuda() if dist.get_backend() == "nccl" else state
            dist.broadcast(st, src=g)
            state = st.cpu()
        return int(state[2].item())


# =====================================================================================================
# Band-resident pipeline (round 2): ONE persistent launch per rank.  The halo row between two bands travels
# as {tag, H} granules (sw_fill_band_device): rank g's kernel writes the granules of its last row strip by
# strip and raises a per-strip flag in host-pinned memory; this module forwards finished column chunks to
# rank g+1 (RCCL / gloo point-to-point), whose kernel -- launched at the same time -- starts each strip the
# moment that strip's 64 granules have landed.  No launch per tile, no host synchronisation inside a band;
# the only collective stays the one all_reduce(MAX) of the packed arg-max key.
# =====================================================================================================
class BandResident(BandPipeline):
    def __init__(self, dist, rank, world, engine, a, b, scores=(3, -3, -2), nchunks=64, p_dtype=None, want_h=True,
                 reserve_cus=16, timeout_s=120.0, placement=True):
        """placement=False: plain allocations (the placement search fills with the whole GPU: not when ranks share one)."""
        import torch
        self.dist, self.rank, self.world, self.eng, self.scores = dist, rank, world, engine, scores
        self.a, self.b = np.asarray(a, np.uint8), np.asarray(b, np.uint8)
        self.cols, self.rows = len(self.a), len(self.b)
        self.bands = band_bounds(self.rows, world)
        self.lo, self.hi = self.bands[rank]
        self.active = self.hi > self.lo
        self.timeout_s = timeout_s
        self.tag = 0
        self.nccl = dist.is_initialized() and dist.get_backend() == "nccl"
        t = torch
        dev = f"cuda:{engine.device}"
        S = (self.cols + 62) // 63
        per = max(1, -(-S // max(1, nchunks)))                      # strips per forwarded chunk
        self.chunks = [(s0, min(S, s0 + per)) for s0 in range(0, S, per)]
        self.S = S
        self.reserve = reserve_cus if world > 1 else 0
        if self.active:
            br = self.hi - self.lo
            self.d_a, _ = engine.to_device(self.a)
            self.d_b, _ = engine.to_device(self.b[self.lo:self.hi])
            self.placement_ms = []
            if placement and (world == 1 or self.nccl) and want_h:
                # where H and P lie in HBM moves the fill by 15-30 %: take them from the C-ABI allocator,
                # which puts them into different classes of the HBM (candidates classified by a store probe).  The probe wants the GPU
                # to itself, so not when several ranks may share one (gloo rehearsals).
                self._out, self.placement_ms = engine.alloc_outputs(self.d_a, self.d_b, self.cols, br, p_dtype=p_dtype, scores=scores,
                                                                    trials=self._placement_trials(br, p_dtype))
                self.H, self.P = self._out.H, self._out.P
                self.H[0].zero_()
                self.P[0].zero_()
            else:
                self.H = t.zeros((br + 1, self.cols + 1), dtype=t.int32, device=dev) if want_h else None
                self.P = t.zeros((br + 1, self.cols + 1), dtype=p_dtype or t.int32, device=dev)
            self.res = t.zeros(3, dtype=t.int64, device=dev)
            self.top = t.zeros(self.cols + 1, dtype=t.int64, device=dev) if self._prev() is not None else None
            self.bot = t.zeros(self.cols + 1, dtype=t.int64, device=dev) if self._next() is not None else None
            self.done = t.zeros(S, dtype=t.int32).pin_memory() if self.bot is not None else None
            self.side = t.cuda.Stream(device=dev)
            self.tiles = self        # the distributed traceback of BandPipeline walks self.P
            self.band_rows = br
        if self.nccl and world > 1:
            # RCCL builds a communicator per rank pair at the first send/recv: do that here, down the chain, not while
            # band kernels wait for each other
            probe = t.zeros(2, dtype=t.int64, device=dev)
            if self.active and self._prev() is not None:
                dist.recv(probe, src=self._prev())
            if self.active and self._next() is not None:
                dist.send(probe, dst=self._next())
            t.cuda.synchronize()

    # columns of chunk k (granule indices): strips [s0, s1) own columns 63*s0+1 .. 63*s1; column 0 rides with the first chunk
    def _placement_trials(self, br, p_dtype):
        # every candidate holds one more copy of P until the search ends: fewer candidates for bands that fill the HBM
        cells = (br + 1) * (self.cols + 1)
        total = cells * (4 + (1 if p_dtype is not None and p_dtype.itemsize == 1 else 4))
        return 0   # (round 4: the allocator classifies candidates with its store probe -- no trial fills, losers of a big search freed at once)

    # Cut at even granule indices: every forwarded piece is then 16-byte aligned, so no transport has a reason to move a
    # granule in pieces smaller than its 8 bytes (a granule is valid only as a whole).  A granule that is cut off rides
    # with the next chunk; its strip is finished by then.
    def _cols_of(self, k):
        s0, s1 = self.chunks[k]
        c0 = 0 if s0 == 0 else (63 * s0 + 1) & ~1
        c1 = self.cols + 1 if s1 >= self.S else (63 * s1 + 1) & ~1
        return c0, c1

    def fill(self):
        import time
        import torch
        dist = self.dist
        self.tag += 1
        tag = self.tag
        best_key = 0
        if self.active:
            prev, nxt = self._prev(), self._next()
            recvs = []
            stage = []
            nk = len(self.chunks)
            kr = 0                 # next chunk to receive (gloo: blocking receives interleaved with the sends)

            def post_recv():
                # RCCL: the receive lands straight in the granule buffer the kernel polls -- the data is its own flag.  Issued
                # from the idle side stream (an op issued from the fill stream is ordered behind the running band kernel); at
                # most two are outstanding, so that a send never queues behind a long line of receives should both directions
                # share one RCCL stream.
                nonlocal kr
                c0, c1 = self._cols_of(kr)
                with torch.cuda.stream(self.side):
                    recvs.append(dist.irecv(self.top[c0:c1], src=prev))
                kr += 1

            if prev is not None and self.nccl:
                while kr < min(2, nk):
                    post_recv()
            self.eng.fill_band(self.d_a, self.cols, self.d_b, self.band_rows, self.rows, self.H, self.P, self.res,
                               top_gran=self.top, top_tag=tag if self.top is not None else 0, bot_gran=self.bot,
                               bot_tag=tag if self.bot is not None else 0, bot_done=self.done, reserve_cus=self.reserve, scores=self.scores)
            deadline = time.time() + self.timeout_s
            sends = []
            flags = self.done.numpy() if self.done is not None else None
            ks = 0                 # next chunk to send
            while (nxt is not None and ks < nk) or (prev is not None and kr < nk):
                progressed = False
                if prev is not None and kr < nk:
                    if self.nccl:
                        if recvs[kr - 2].is_completed():
                            post_recv()
                            progressed = True
                    else:
                        c0, c1 = self._cols_of(kr)
                        buf = torch.empty(c1 - c0, dtype=torch.int64)
                        dist.recv(buf, src=prev)     # ranks are chained: the sender forwards chunks in order
                        with torch.cuda.stream(self.side):
                            self.top[c0:c1].copy_(buf, non_blocking=False)
                        kr += 1
                        progressed = True
                if nxt is not None and ks < nk:
                    s0, s1 = self.chunks[ks]
                    if bool((flags[s0:s1] == tag).all()):
                        c0, c1 = self._cols_of(ks)
                        if self.nccl:
                            with torch.cuda.stream(self.side):
                                sends.append(dist.isend(self.bot[c0:c1], dst=nxt))
                        else:
                            with torch.cuda.stream(self.side):
                                buf = self.bot[c0:c1].to("cpu")
                            sends.append(dist.isend(buf, dst=nxt))
                            stage.append(buf)
                        ks += 1
                        progressed = True
                if progressed:
                    deadline = time.time() + self.timeout_s
                elif time.time() > deadline:
                    raise RuntimeError(f"rank {self.rank}: band pipeline stalled at chunk {ks}/{nk} (send) {kr}/{nk} (recv)")
            for w in recvs + sends:
                w.wait()
            self.eng.synchronize()
            r = self.res.cpu().tolist()
            if r[2] < 0:
                from . import SwError
                raise SwError(-62, "band kernel: hand-off wait timed out")
            if r[1] > 0:
                row, col = divmod(r[0], self.cols + 1)
                gidx = (self.lo + row) * (self.cols + 1) + col
                best_key = (r[1] << 40) | (KEY_MASK - gidx)
        import torch
        key = torch.tensor([best_key], dtype=torch.int64)
        if self.nccl:
            key = key.cuda()
        if dist.is_initialized() and self.world > 1:
            dist.all_reduce(key, op=dist.ReduceOp.MAX)
        k = int(key.item())
        return (k >> 40, KEY_MASK - (k & KEY_MASK)) if k else (0, 0)

    # ---- the interface BandPipeline.traceback() needs from its tile engine
    def walk(self, pos):
        t = self.eng.torch
        cap = self.band_rows + self.cols + 2
        path = t.zeros(cap, dtype=t.int64, device=self.P.device)
        from . import lib, _check
        _check(lib().sw_traceback_device_ex(self.eng._h, self.P.data_ptr(), self.P.element_size(), self.cols, self.band_rows, int(pos),
                                            path.data_ptr(), cap, self.res.data_ptr(), self.eng._stream()))
        self.eng.synchronize()
        n = int(self.res[2].item())
        return n, (int(path[n - 1].item()) if n else -1)

    def pred_of(self, idx):
        m = self.cols + 1
        pr = -int(self.P.view(-1)[idx].item())
        return idx - m - 1 if pr == 3 else idx - m if pr == 1 else idx - 1

    def matrices(self):
        return (self.H.cpu().numpy() if self.H is not None else None), self.P.cpu().numpy()
