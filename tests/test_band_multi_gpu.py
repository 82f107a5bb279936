"""GPU: the band-resident multi-rank pipeline (multi.BandResident): one persistent launch per rank, halo rows forwarded
as granules.  World size 2 over gloo with both ranks on the one GPU of the test box (each launch capped to under half
of the CUs), and over nccl (= RCCL) when the box has two GPUs.  The stitched H / P, the arg-max and the distributed
traceback must equal serial_smithW on the whole matrix."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


class _Work:
    """A transfer of the stand-in transport below: gloo moves a host copy (a thread waits for it: gloo's point-to-point work
    objects only learn of their completion inside wait()), the device copy follows."""
    def __init__(self, work, done=None, keep=None):
        import threading
        self.keep, self.err = keep, None
        self.finished = threading.Event()

        def run():
            try:
                work.wait()
                if done is not None:
                    done()
            except Exception as e:   # noqa: BLE001 -- reported by wait()
                self.err = e
            self.finished.set()
        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def is_completed(self):
        return self.finished.is_set()

    def wait(self):
        self.finished.wait()
        if self.err is not None:
            raise self.err


class _RcclStandIn:
    """Looks like torch.distributed over nccl to BandResident (device tensors in isend / irecv / send / recv / all_reduce /
    broadcast, backend name "nccl"), runs over gloo through host copies.  It cannot test RCCL; it does drive every line of the
    pipeline's RCCL branch -- receives posted ahead from the side stream, completion polling, sends as chunks finish -- with the
    real kernels, on a box where two RCCL ranks cannot share the one GPU."""
    def __init__(self, dist):
        self.d = dist
        self.ReduceOp = dist.ReduceOp
        self.stream = torch.cuda.Stream()      # (RCCL has streams of its own: a copy on the fill stream would wait for the band kernel)

    def _land(self, t, buf):
        with torch.cuda.stream(self.stream):
            t.copy_(buf)
        self.stream.synchronize()

    def is_initialized(self):
        return True

    def get_backend(self):
        return "nccl"

    def irecv(self, t, src):
        buf = torch.empty(t.shape, dtype=t.dtype)
        return _Work(self.d.irecv(buf, src=src), done=lambda: self._land(t, buf), keep=buf)

    def isend(self, t, dst):
        buf = t.cpu()
        return _Work(self.d.isend(buf, dst=dst), keep=buf)

    def recv(self, t, src):
        self.irecv(t, src).wait()

    def send(self, t, dst):
        self.isend(t, dst).wait()

    def all_reduce(self, t, op=None):
        buf = t.cpu()
        self.d.all_reduce(buf, op=op)
        t.copy_(buf)

    def broadcast(self, t, src):
        buf = t.cpu()
        self.d.broadcast(buf, src=src)
        t.copy_(buf)


def _worker(rank, world, port, backend, cols, rows, seed, p8, outdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    standin = backend == "standin"
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev}"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if standin:
        dist = _RcclStandIn(dist)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sw = importlib.import_module("smith-waterman_amd")
    multi = importlib.import_module("smith-waterman_amd.multi")
    import oracle_lib
    a, b = oracle_lib.Oracle().generate(cols, rows, seed)
    eng = sw.Engine(dev)
    if backend != "nccl":    # two persistent launches share one GPU: each takes well under half of the CUs
        eng.set_option("max_blocks", max(8, eng.get_option("num_cus") // 2 - 16))
    eng.set_option("band_wait_ms", 30000)
    pipe = multi.BandResident(dist, rank, world, eng, a, b, nchunks=8, p_dtype=torch.int8 if p8 else None, reserve_cus=16, timeout_s=60,
                              placement=not standin)
    for _ in range(2):      # twice: the second fill meets the first one's granules with the old tag
        score, pos = pipe.fill()
    plen = pipe.traceback(pos)
    H, P = pipe.matrices()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), H=H, P=P.astype(np.int32), meta=np.array([score, pos, plen, pipe.lo, pipe.hi]))
    eng.close()
    (dist.d if standin else dist).destroy_process_group()


def _run(tmp_path, oracle, backend, world, cols, rows, p8):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, backend, cols, rows, 5, p8, str(tmp_path)), nprocs=world, join=True)
    a, b = oracle.generate(cols, rows, 5)
    H, P, mp_ = oracle.fill(a, b)
    path = oracle.backtrack(P, mp_)   # P is now negated along the path
    for r in range(world):
        g = np.load(tmp_path / f"r{r}.npz")
        score, pos, plen, lo, hi = [int(x) for x in g["meta"]]
        assert (score, pos) == (int(H.flat[mp_]), mp_) and plen == len(path)
        assert np.array_equal(g["H"][1:], H[lo + 1:hi + 1]), f"rank {r} H"
        assert np.array_equal(g["P"][1:], P[lo + 1:hi + 1]), f"rank {r} P (incl. negated path)"


@pytest.mark.parametrize("p8", [False, True], ids=["p32", "p8"])
def test_two_ranks_on_one_gpu_gloo(tmp_path, oracle, p8):
    _run(tmp_path, oracle, "gloo", 2, 3000, 1200, p8)


@pytest.mark.parametrize("cols,rows", [(3000, 1200), (5040, 2016)], ids=["one_column_kernel", "two_column_kernel"])
def test_rccl_branch_control_flow_over_a_stand_in_transport(tmp_path, oracle, cols, rows):
    _run(tmp_path, oracle, "standin", 2, cols, rows, True)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the RCCL transport of the band pipeline)")
def test_two_ranks_rccl(tmp_path, oracle):
    _run(tmp_path, oracle, "nccl", 2, 20000, 4000, True)
