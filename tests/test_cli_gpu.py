"""GPU: the smithW host program (reference command line, serial_smithW.c:71-180) through the C-ABI."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle_lib import ROOT

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "smith-waterman_amd", "smithW")
REF_DEBUG = os.path.join(ROOT, "oracle", "_ref", "serial_smithW_debug")


def run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=120)


def test_builtin_known_answer_run():
    r = run()
    assert r.returncode == 0, r.stderr
    assert "Elapsed time for scoring matrix computation:" in r.stdout and "Elapsed time for backtracking:" in r.stdout
    assert "Verifying correctness using builtin data =1" in r.stdout and "maxPos = 69, H[maxPos] = 13" in r.stdout


def matrices(text):
    sim = text.split("Similarity Matrix:")[1].split("Predecessor Matrix:")[0]
    H = [[int(x) for x in ln.split()] for ln in sim.strip().splitlines() if re.match(r"^[-\d\s]+$", ln) and ln.strip()]
    pred = re.sub(r"\x1b\[[0-9;]*m", "", text.split("Predecessor Matrix:")[1])
    P = [[c for c in ln if c in "↑←↖-"] for ln in pred.strip().splitlines()]
    P = [row for row in P if len(row) == len(H[0])]
    return np.array(H), P


def test_dump_matches_oracle_and_reference_binary(oracle):
    r = run("40", "30", "--dump")
    assert r.returncode == 0, r.stderr
    H, P = matrices(r.stdout)
    a, b = oracle.generate(40, 30, 1)
    oH, oP, mp = oracle.fill(a, b)
    assert np.array_equal(H, oH)
    sym = {0: "-", 1: "↑", 2: "←", 3: "↖"}
    assert [[sym[abs(int(v))] for v in row] for row in oP] == P
    # The unmodified reference program built with -DDEBUG prints the same two matrices.  It only
    # survives square inputs (its debug printing walks off its buffers when cols != rows and the
    # process dies with SIGSEGV/SIGABRT mid-output), so the binary-vs-binary check uses 36x36.
    if os.path.exists(REF_DEBUG):
        r = run("36", "36", "--dump")
        assert r.returncode == 0, r.stderr
        H, P = matrices(r.stdout)
        ref = subprocess.run([REF_DEBUG, "36", "36"], capture_output=True, text=True, timeout=60)
        assert ref.returncode == 0
        rH, rP = matrices(ref.stdout)
        assert H.shape == (37, 37) and len(P) == 37
        assert np.array_equal(H, rH) and P == rP


def test_bad_usage_and_scores():
    assert run("--bogus").returncode == 2
    r = run("7", "5", "--scores", "5", "-3", "-4", "--dump")
    assert r.returncode == 0 and "Similarity Matrix" in r.stdout


def test_fasta_input_matches_oracle(oracle, tmp_path):
    """Real sequences instead of generate(): --fasta A B, then the same dump / known-answer path."""
    rng = np.random.default_rng(5)
    a = rng.choice(list(b"ACGT"), 300).astype(np.uint8)
    b = np.concatenate([a[40:220], rng.choice(list(b"ACGT"), 60).astype(np.uint8)])
    fa, fb = tmp_path / "a.fa", tmp_path / "b.fa"
    fa.write_bytes(b">a\n" + b"\n".join(bytes(a[i:i + 70]) for i in range(0, len(a), 70)) + b"\n")
    fb.write_bytes(b">skip\nAAAA\n>b\n" + bytes(b).lower() + b"\n")
    r = run("--fasta", str(fa), str(fb), "--record-b", "1", "--dump")
    assert r.returncode == 0, r.stderr
    H, P = matrices(r.stdout)
    oH, oP, mp = oracle.fill(a, b)
    assert H.shape == (len(b) + 1, len(a) + 1) and np.array_equal(H, oH)
    assert f"maxPos = {mp}, H[maxPos] = {int(oH.flat[mp])}" in r.stdout or str(mp) in r.stdout
    assert run("--fasta", str(fa), str(tmp_path / "nope.fa")).returncode != 0


def test_multi_gpu_flag_two_bands_on_one_gpu(oracle):
    """smithW --devices 0,0: the multi-GPU path of the host program (sw_multi_*), two row bands sharing the test box's GPU."""
    r = run("3000", "1100", "--devices", "0,0")
    assert r.returncode == 0, r.stderr
    a, b = oracle.generate(3000, 1100, 1)
    H, P, mp = oracle.fill(a, b)
    path = oracle.backtrack(P, mp)
    assert f"maxPos = {mp}, H[maxPos] = {int(H.flat[mp])}, path length = {len(path)}" in r.stdout and "(2 row bands)" in r.stdout


REF_OMP_DEBUG = os.path.join(ROOT, "oracle", "_ref", "omp_smithW_debug")


@pytest.mark.skipif(not os.path.exists(REF_OMP_DEBUG), reason="oracle/_ref/omp_smithW_debug is built in the build container only")
def test_labelled_dump_is_byte_identical_to_the_reference_openmp_program(tmp_path):
    """--dump-labels: the header-row printers of omp_smithW.c:426-483.  That program seeds rand() with time(), so its
    sequences are read back from its own dump (header row = a, row labels = b) and given to the CLI as FASTA files; the two
    matrix blocks, escape sequences of the red traceback path included, must then be identical byte for byte."""
    ref = subprocess.run([REF_OMP_DEBUG, "37", "29"], capture_output=True, text=True, timeout=120, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert ref.returncode == 0, ref.stderr
    block = ref.stdout.split("\nSimilarity Matrix:\n")[1]
    lines = block.split("\n")
    a = "".join(lines[0].split("\t")[2:]).strip()
    b = "".join(ln.split("\t")[0] for ln in lines[2:2 + 29])
    assert len(a) == 37 and len(b) == 29 and set(a + b) <= set("ACGT")
    fa, fb = tmp_path / "a.fa", tmp_path / "b.fa"
    fa.write_text(">a\n" + a + "\n")
    fb.write_text(">b\n" + b + "\n")
    r = run("--fasta", str(fa), str(fb), "--dump-labels")
    assert r.returncode == 0, r.stderr
    mine = r.stdout.split("\nSimilarity Matrix:\n")[1]
    # omp_smithW.c keeps the first maximum in anti-diagonal order, serial_smithW.c (our rule) the lowest linear index: with a tied
    # maximum the two programs may trace different paths, so the red path is only compared when the maximum is unique
    cells = [int(x) for ln in lines[1:2 + 29] for x in ln.split("\t") if re.fullmatch(r"-?\d+", x)]
    if cells.count(max(cells)) == 1:
        assert mine.rstrip("\n") == block.rstrip("\n")
    else:
        strip = lambda t: re.sub(r"\x1b\[[0-9;]*m", "", t).rstrip("\n")
        assert strip(mine) == strip(block)
