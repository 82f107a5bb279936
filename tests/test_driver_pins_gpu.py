"""Driver-run (`-m gpu`) copies of the checks that pin the oracle and the host half of the path, so that they are observed on
the GPU box too and not only in the build container:
  * the oracle against the fixtures generated from the real reference (tests/test_oracle.py),
  * sw_nelement / sw_first_diag_element (nElement / calcFirstDiagElement, omp_smithW.c:260-291) against the oracle
    (tests/test_abi.py) -- and, below, against a matrix filled on the GPU: walking the anti-diagonals in the reference's
    wavefront order and recomputing every cell from its three neighbours must reproduce the device H and P."""
import numpy as np
import pytest

from test_abi import test_generate_full_fixture, test_traceback_host_matches_reference, test_wavefront_indexing_matches_oracle  # noqa: F401
from test_oracle import (test_backtrack_matches_reference, test_builtin_known_answers, test_fill_matches_reference,  # noqa: F401
                         test_generate_and_hashes, test_glibc_rand_first_draws, test_readme_screenshot_scoring, test_wavefront_indexing)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cols,rows", [(300, 200), (64, 257), (513, 96)])
def test_wavefront_order_reproduces_the_device_matrices(engine, swamd, cols, rows):
    """omp_smithW.c:203-216 with the C-ABI's nElement / calcFirstDiagElement over H and P filled by the HIP kernel: every cell
    of anti-diagonal i is (si - j, sj + j), each cell is visited exactly once, and similarityScore's recurrence
    (omp_smithW.c:331-388) applied to the device values of its neighbours gives the device values of the cell."""
    a, b = swamd.generate(cols, rows, 5)
    out = engine.fill(a, b)
    H, P = out.H.cpu().numpy().astype(np.int64), out.P.cpu().numpy()
    m, n = cols + 1, rows + 1
    seen = np.zeros((n, m), np.int32)
    for i in range(1, m + n - 3 + 1):
        ne = swamd.n_element(i, m, n)
        si, sj = swamd.first_diag_element(i, m, n)
        rr = si - np.arange(ne)
        cc = sj + np.arange(ne)
        assert ((rr + cc) == i + 1).all() and rr.min() >= 1 and cc.max() <= cols
        seen[rr, cc] += 1
        diag = H[rr - 1, cc - 1] + np.where(a[cc - 1] == b[rr - 1], 3, -3)
        up = H[rr - 1, cc] - 2
        left = H[rr, cc - 1] - 2
        best = np.maximum(np.maximum(diag, up), np.maximum(left, 0))
        pred = np.where(best == 0, 0, np.where(diag == best, 3, np.where(up == best, 1, 2)))
        assert np.array_equal(H[rr, cc], best) and np.array_equal(P[rr, cc], pred), f"anti-diagonal {i}"
    assert (seen[1:, 1:] == 1).all() and seen[0].sum() == 0 and seen[:, 0].sum() == 0
