"""Host-side scalars of the heat-kernel path: Chebyshev coefficients and the spectral bound.

The device does the N x N work (csrc/heat.hip); what is computed here is O(order) or one
sparse eigenvalue estimate, exactly the quantities the reference obtains from PyGSP
(model/HSD.py:50-53, model/multiscale_HSD.py:27-30).
"""
from __future__ import annotations

import numpy as np


def cheby_coeff_heat(scale, lmax, m=50, quad=None):
    """pygsp.filters.approximations.compute_cheby_coeff for pygsp.filters.Heat(G, tau=scale*lmax)
    (model/HSD.py:52-53): kernel g(x) = exp(-scale * x) on [0, lmax],
    c_j = 2/Nq * sum_k g(a1 cos t_k + a2) cos(j t_k), t_k = pi (k + 1/2)/Nq, Nq = m + 1."""
    nq = (m + 1) if quad is None else int(quad)
    a1 = a2 = lmax / 2.0
    t = np.pi * (np.arange(nq) + 0.5) / nq
    g = np.exp(-scale * (a1 * np.cos(t) + a2))
    j = np.arange(m + 1)[:, None]
    return (2.0 / nq) * (np.cos(j * t[None, :]) @ g)


def exact_order(scale, lmax, tol=4e-15, max_order=20000):
    """Order at which the Chebyshev series of exp(-scale x) on [0, lmax] has decayed to rounding:
    the coefficients behave like 2 exp(-z) I_k(z), z = scale*lmax/2, i.e. they fall
    super-exponentially once k exceeds ~z.  The first run of 8 coefficients below tol*max marks
    the end of the series (what follows is quadrature noise)."""
    z = scale * lmax / 2.0
    m = int(min(max_order, 2.0 * z + 64))
    c = np.abs(cheby_coeff_heat(scale, lmax, m, quad=2 * (m + 1)))
    small = c < max(tol * c.max(), 2e-15)          # 2e-15: absolute noise floor of the quadrature
    run = 0
    for k in range(m + 1):
        run = run + 1 if small[k] else 0
        if run == 8:
            return max(k - 7, 2)
    return m


def exact_coefficients(scale, lmax):
    """Coefficients that reproduce exp(-scale L) to rounding (the reference's approx=False branch,
    model/HSD.py:61-63, without an O(N^3) eigendecomposition)."""
    m = exact_order(scale, lmax)
    return cheby_coeff_heat(scale, lmax, m, quad=2 * (m + 1) + 64)


def estimate_lmax(lap_rowptr, lap_col, lap_val, n):
    """PyGSP Graph.estimate_lmax as the reference uses it (model/HSD.py:51, model/multiscale_HSD.py:27-30): the
    largest eigenvalue by ARPACK (k=1, tol=5e-3, ncv=min(n,10)) times 1.01 -- exactly that product, also where it
    exceeds the Anderson-Morley bound (stars, complete and regular bipartite graphs reach the bound, so clipping
    to it would move MultiHSD's scales by up to 1 % against the reference).  PyGSP itself falls back to a
    degree bound with a warning when ARPACK does not converge; here the Anderson-Morley bound
    max over edges (d_u + d_v) takes that role, also for graphs too small for ARPACK (n < 3), and the fallback
    is logged, never silent."""
    import logging
    if n >= 3:
        try:
            from scipy.sparse import csr_matrix
            from scipy.sparse.linalg import ArpackError, ArpackNoConvergence, eigsh
        except ImportError as e:
            logging.getLogger("hsd_amd").warning("estimate_lmax: scipy unavailable (%s); using the Anderson-Morley bound", e)
        else:
            L = csr_matrix((lap_val, lap_col, lap_rowptr), shape=(n, n))
            try:
                return float(eigsh(L, k=1, tol=5e-3, ncv=min(n, 10), return_eigenvectors=False)[0]) * 1.01
            except (ArpackNoConvergence, ArpackError) as e:
                logging.getLogger("hsd_amd").warning("estimate_lmax: ARPACK failed (%s); using the Anderson-Morley bound", e)
    return spectral_bound(lap_rowptr, lap_col, lap_val, n)


def spectral_bound(lap_rowptr, lap_col, lap_val, n):
    """Guaranteed upper bound of the Laplacian spectrum (Anderson-Morley)."""
    diag = np.zeros(n)
    rows = np.repeat(np.arange(n), np.diff(lap_rowptr))
    on_diag = rows == lap_col
    diag[rows[on_diag]] = lap_val[on_diag]
    off = ~on_diag
    return float(np.max(diag[rows[off]] + diag[lap_col[off]])) if off.any() else float(max(diag.max(), 1e-12))
