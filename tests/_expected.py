"""Test helper (test infrastructure, imports the oracle): the result of ``P * P`` + cleanup for an operator WITHOUT duplicate rows and
without accidental coincidences among its pair products, assembled from the C oracle's pair coefficients instead of from its 10^8
materialised product rows (27 GB at BASELINE cfg3).

What the reference does (utils.py:230-279 after base.py:783-792): product row ``o*N + i`` is ``A_i ^ A_o`` with coefficient
``v[o, i]`` (left factor i, right factor o); duplicate rows merge AT THEIR FIRST OCCURRENCE with coefficients added in input order
(np.add.at), then ``abs(c) > 1e-15`` keeps.  For distinct rows A_i the only coincidences are the identity (all pairs (i, i), first
at index 0, coefficients added for i = 0 .. N-1) and the twins (o, i) / (i, o), first at the index with the smaller o.  So the
result is: identity; then the pairs o < i in (o, i) order with coefficient ``v[o, i] + v[i, o]`` (exactly 0 for anticommuting
factors).  ``tests/test_oracle_golden.py::test_squared_builder_equals_oracle_mul`` pins this builder to ``oracle_c.mul``."""
import numpy as np
from oracle import oracle_c as oc


def squared_expected(rows, coeff, thr=1e-15, block=500):
    """-> (o_idx, i_idx, coeff) of the kept terms in output order; the identity term, if kept, comes first as (o, i) = (0, 0)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    coeff = np.ascontiguousarray(coeff, dtype=np.complex128)
    N = rows.shape[0]
    V = oc.mul_allpairs_coeff(rows, coeff, rows, coeff, True).reshape(N, N)        # V[o, i]
    VT = np.empty_like(V)
    for b in range(0, N, block):                                                   # blocked transpose (a strided V.T add is 4x slower)
        VT[:, b:b + block] = V[b:b + block].T
    ident = np.add.accumulate(np.diagonal(V))[-1]                                  # sequential, like np.add.at on one slot
    o_parts, i_parts, c_parts = [], [], []
    if abs(ident) > thr:
        o_parts.append(np.zeros(1, dtype=np.int64)); i_parts.append(np.zeros(1, dtype=np.int64)); c_parts.append(np.array([ident]))
    cols = np.arange(N)
    for b in range(0, N, block):
        S = V[b:b + block] + VT[b:b + block]
        mask = (np.abs(S) > thr) & (cols[None, :] > np.arange(b, min(N, b + block))[:, None])
        o_rel, i_idx = np.nonzero(mask)
        o_parts.append(o_rel + b); i_parts.append(i_idx); c_parts.append(S[o_rel, i_idx])
    return np.concatenate(o_parts), np.concatenate(i_parts), np.concatenate(c_parts)
