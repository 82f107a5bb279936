"""CPU: the C-ABI library loads, exports every symbol include/swhip.h declares, and its host-only
entry points (generator, wavefront indexing, host traceback, argument checking) match the
reference fixtures.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle_lib import ROOT, golden, golden_hashes


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "swhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sw_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(swamd):
    L = swamd.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"libswhip.so does not export {n}"
    assert sorted(swamd.ABI) == names, "python binding table and header disagree"
    assert b"gfx950" in L.sw_version()


@pytest.mark.parametrize("name", sorted(golden_hashes()))
def test_generate_matches_reference(swamd, oracle, name):
    h = golden_hashes()[name]
    if name == "kat_builtin":
        return
    a, b = swamd.generate(h["cols"], h["rows"], h["seed"])
    assert bytes(a[:32]).decode() == h["a_head"] and bytes(b[:32]).decode() == h["b_head"]
    oa, ob = oracle.generate(h["cols"], h["rows"], h["seed"])
    assert np.array_equal(a, oa) and np.array_equal(b, ob)


def test_generate_full_fixture(swamd):
    g = golden("rand_256x256_s1")
    a, b = swamd.generate(256, 256, 1)
    assert np.array_equal(a, g["a"]) and np.array_equal(b, g["b"])
    g = golden("rand_65x130_s7")
    a, b = swamd.generate(65, 130, 7)
    assert np.array_equal(a, g["a"]) and np.array_equal(b, g["b"])


def test_wavefront_indexing_matches_oracle(swamd, oracle):
    for (m, n) in [(9, 10), (10, 9), (2, 2), (2, 9), (9, 2), (40, 17), (257, 257)]:
        for i in range(1, m + n - 3 + 1):
            assert swamd.n_element(i, m, n) == oracle.n_element(i, m, n)
            assert swamd.first_diag_element(i, m, n) == oracle.first_diag_element(i, m, n)


@pytest.mark.parametrize("name", ["kat_builtin", "rand_8x9_s1", "rand_1x1_s1", "rand_256x256_s1", "rand_300x200_s1", "rand_1x77_s3", "rand_77x1_s3"])
def test_traceback_host_matches_reference(swamd, name):
    g = golden(name)
    P = g["P0"].copy()
    path = swamd.traceback_host(P, int(g["meta"][3]))
    assert np.array_equal(P, g["P1"]) and np.array_equal(path, g["path"])


def test_read_fasta(swamd, tmp_path):
    """sw_read_fasta: records, comments, wrapped lines, lower case, CRLF, headerless files, missing records."""
    f = tmp_path / "x.fa"
    f.write_bytes(b";comment\n>seq1 first\nACGT\nacgt \r\nNN\n>seq2\n\nTT\tGG\n>empty\n>last\nA")
    assert bytes(swamd.read_fasta(str(f))) == b"ACGTACGTNN"
    assert bytes(swamd.read_fasta(str(f), 1)) == b"TTGG"
    assert bytes(swamd.read_fasta(str(f), 2)) == b""
    assert bytes(swamd.read_fasta(str(f), 3)) == b"A"
    with pytest.raises(swamd.SwError):
        swamd.read_fasta(str(f), 4)
    with pytest.raises(swamd.SwError):
        swamd.read_fasta(str(tmp_path / "missing.fa"))
    g = tmp_path / "plain.txt"
    g.write_bytes(b"GATTACA\nGATT\n")
    assert bytes(swamd.read_fasta(str(g))) == b"GATTACAGATT"
    with pytest.raises(swamd.SwError):
        swamd.read_fasta(str(g), 1)
    n = ctypes.c_int64()
    buf = ctypes.create_string_buffer(4)
    assert swamd.lib().sw_read_fasta(str(f).encode(), 0, buf, 4, ctypes.byref(n)) == 0   # truncated copy, full length reported
    assert n.value == 10 and buf.raw == b"ACGT"


def test_argument_errors(swamd):
    L = swamd.lib()
    assert L.sw_generate(-1, 4, 1, None, None) == -22 and b"sw_generate" in L.sw_last_error()
    assert L.sw_traceback_host(None, 4, 4, 0, None, 0, None) == -22
    assert L.sw_fill_device(None, None, 4, None, 4, None, None, 4, None, None, None, None) == -22
    assert L.sw_set_option(None, b"x", 1) == -22


def test_create_without_gpu_fails_loudly(swamd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = swamd.lib().sw_create(0, ctypes.byref(h))
    assert rc == -19 and h.value is None
    with pytest.raises(RuntimeError):
        swamd.Engine(0)


@pytest.mark.parametrize("name", ["kat_builtin", "rand_8x9_s1", "rand_1x1_s1", "rand_256x256_s1", "rand_300x200_s1", "rand_65x130_s7"])
def test_align_auto_host_side_matches_reference(swamd, name):
    """sw_align_auto without a GPU context: the host fill (sw_fill_cpu, product code -- not the oracle) + host traceback
    reproduce the reference fixtures bit for bit: H, P after the traceback, maxPos, score, path length."""
    g = golden(name)
    r = swamd.align_auto(g["a"], g["b"])
    assert not r["used_gpu"]
    assert np.array_equal(r["H"], g["H"]) and np.array_equal(r["P"], g["P1"])
    assert r["max_pos"] == int(g["meta"][3]) and r["max_score"] == int(g["meta"][4]) and r["path_len"] == len(g["path"])


def test_align_auto_multi_without_devices_is_the_host_leg(swamd):
    """sw_align_auto_multi with no device and no context: executor 0 (sw_fill_cpu + host traceback), the reference's outputs."""
    g = golden("rand_300x200_s1")
    r = swamd.align_auto(g["a"], g["b"], devices=[])
    assert r["executor"] == 0 and not r["used_gpu"]
    assert np.array_equal(r["H"], g["H"]) and np.array_equal(r["P"], g["P1"])
    assert r["max_pos"] == int(g["meta"][3]) and r["path_len"] == len(g["path"])
    assert swamd.lib().sw_align_auto_multi(None, None, 2, None, 4, None, 4, None, None, None, None, None, 0) == -22
