"""Markov state model + PCCA+ memberships for the MSM soft-count decoder (post_hoc.py:1286-1336).

The reference calls a THIRD-PARTY package here -- ``deeptime`` (pinned ``^0.4.5`` in /root/reference/pyproject.toml:58;
``deeptime.markov.TransitionCountEstimator(lagtime, count_mode="sliding")``,
``deeptime.markov.msm.MaximumLikelihoodMSM(reversible=True)`` and ``MarkovStateModel.pcca(m)``) -- which is absent from
this image and from /root/reference (no vendored copy).  This module restates the PUBLISHED algorithms those three calls
implement; **parity with deeptime is unpinned** (nothing here could be run against it), the tests check the defining
properties instead (row-stochastic T, detailed balance, the maximum-likelihood fixed point, memberships in the simplex,
recovery of planted metastable blocks):

* sliding count matrix: C[i, j] = #{t : x_t = i, x_{t+lag} = j} over all trajectories;
* largest strongly connected set of the count graph (deeptime fits the MSM on ``submodel_largest``);
* reversible maximum-likelihood transition matrix by the fixed-point iteration of Prinz et al., J. Chem. Phys. 134,
  174105 (2011) / Trendelkamp-Schroer et al., J. Chem. Phys. 143, 174101 (2015):
  x_ij <- (c_ij + c_ji) / (c_i / x_i + c_j / x_j), T_ij = x_ij / x_i, iterated to a relative change of the row sums
  below 1e-8 (deeptime's ``maxerr``);
* PCCA+ (Roeblitz & Weber, Adv. Data Anal. Classif. 7, 147 (2013)) as implemented in msmtools / deeptime: the m
  dominant right eigenvectors, pi-orthonormalised, inner-simplex initial guess, Nelder-Mead refinement of the
  transformation matrix under the feasibility fill-in, memberships clipped to [0, 1] and row-normalised.

Host-side numpy / scipy: post-hoc statistics on the trainer's outputs, not on the hot path.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def sliding_count_matrix(dtrajs: Sequence[np.ndarray], lagtime: int, n_states: int = None) -> np.ndarray:
    """C[i, j] = number of pairs (x_t, x_{t+lag}) = (i, j), every t of every trajectory ("sliding" counting)."""
    lag = int(lagtime)
    n = int(n_states) if n_states is not None else int(max(int(np.max(d)) for d in dtrajs if len(d))) + 1
    C = np.zeros((n, n), dtype=np.float64)
    for d in dtrajs:
        d = np.asarray(d, dtype=np.int64)
        if d.shape[0] > lag:
            np.add.at(C, (d[:-lag], d[lag:]), 1.0)
    return C


def largest_connected_set(C: np.ndarray) -> np.ndarray:
    """States of the largest strongly connected component of the graph with an edge i -> j where C[i, j] > 0
    (ties: the component found first in state order), sorted."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    n_comp, labels = connected_components(csr_matrix(C > 0), directed=True, connection="strong")
    sizes = np.bincount(labels, minlength=n_comp)
    return np.flatnonzero(labels == int(np.argmax(sizes))).astype(np.int32)


def reversible_mle(C: np.ndarray, maxiter: int = 1000000, maxerr: float = 1e-8) -> Tuple[np.ndarray, np.ndarray]:
    """(T, pi): the reversible maximum-likelihood transition matrix of a (strongly connected) count matrix and its
    stationary distribution."""
    C = np.asarray(C, dtype=np.float64)
    c_i = C.sum(axis=1)
    if np.any(c_i <= 0):
        raise ValueError("count matrix has an empty row: restrict it to its connected set first")
    C2 = C + C.T
    X = C2 / C2.sum()
    x = X.sum(axis=1)
    for _ in range(int(maxiter)):
        denom = (c_i / x)[:, None] + (c_i / x)[None, :]
        X = C2 / denom
        X /= X.sum()
        x_new = X.sum(axis=1)
        err = np.max(np.abs(x_new - x) / np.maximum(x, 1e-300))
        x = x_new
        if err < maxerr:
            break
    T = X / x[:, None]
    return T, x / x.sum()


def _isa(evec: np.ndarray, m: int) -> Tuple[np.ndarray, np.ndarray]:
    """Inner simplex algorithm: m rows of the eigenvector matrix that span the largest simplex -> (chi, rot)."""
    c = evec[:, :m]
    ortho = np.copy(c)
    ind = np.zeros(m, dtype=np.int64)
    ind[0] = int(np.argmax(np.linalg.norm(c, axis=1)))
    ortho -= c[ind[0]][None, :]
    for k in range(1, m):
        temp = np.copy(ortho[ind[k - 1]])
        ortho -= np.outer(ortho @ temp, temp)
        dist = np.linalg.norm(ortho, axis=1)
        ind[k] = int(np.argmax(dist))
        ortho /= dist[ind[k]]
    rot = np.linalg.inv(c[ind])
    return c @ rot, rot


def _fill(rot_crop: np.ndarray, evec: np.ndarray) -> np.ndarray:
    """Completes the (m-1) x (m-1) free block to a feasible m x m transformation (row sums zero, partition of unity)."""
    x, y = rot_crop.shape
    row_sums = rot_crop.sum(axis=1, keepdims=True)
    rc = np.concatenate((-row_sums, rot_crop), axis=1)
    tmp = -(evec[:, 1:] @ rc)
    col_max = tmp.max(axis=0, keepdims=True)
    rot = np.concatenate((col_max, rc), axis=0)
    return rot / col_max.sum()


def pcca_memberships(T: np.ndarray, pi: np.ndarray, m: int) -> np.ndarray:
    """(n, m) PCCA+ memberships of a reversible transition matrix."""
    from scipy.optimize import fmin
    n = T.shape[0]
    m = int(m)
    if m > n:
        raise ValueError("more macrostates than states")
    # reversible: T is similar to a symmetric matrix -> real spectrum, stable eigenvectors
    sq = np.sqrt(pi)
    S = (sq[:, None] * T) / sq[None, :]
    w, V = np.linalg.eigh(0.5 * (S + S.T))
    order = np.argsort(-w)[:m]
    evec = V[:, order] / sq[:, None]                      # right eigenvectors of T
    evec /= np.sqrt(np.sum(evec * evec * pi[:, None], axis=0))[None, :]   # pi-orthonormal
    evec[:, 0] = np.abs(evec[:, 0])                        # the constant one, positive
    for k in range(1, m):                                  # sign convention: largest-magnitude entry positive
        if evec[np.argmax(np.abs(evec[:, k])), k] < 0:
            evec[:, k] = -evec[:, k]
    if m == 1:
        return np.ones((n, 1))
    _chi, rot = _isa(evec, m)
    crop = rot[1:, 1:]
    shape = crop.shape

    def objective(vec):
        r = _fill(vec.reshape(shape), evec)
        return -float(np.sum(r * r / r[0][None, :]))

    opt = fmin(objective, crop.reshape(-1), disp=False)
    chi = evec @ _fill(opt.reshape(shape), evec)
    chi = np.clip(chi, 0.0, 1.0)
    return chi / chi.sum(axis=1, keepdims=True)


def fit_pcca_memberships(dtrajs: List[np.ndarray], lagtime: int, n_macro: int):
    """The reference's ``_fit_pcca_memberships`` (post_hoc.py:1286-1336) on the restated estimators: (active microstate
    ids (n_active,) int32, memberships (n_active, n_macro) float32 -- padded with zero columns and row-normalised when
    fewer than n_macro macrostates are possible) or (None, None) when fewer than two states are connected."""
    C = sliding_count_matrix(dtrajs, lagtime)
    active = largest_connected_set(C)
    if active.shape[0] < 2:
        return None, None
    T, pi = reversible_mle(C[np.ix_(active, active)])
    k = int(min(n_macro, active.shape[0]))
    if k < 2:
        return None, None
    chi_eff = pcca_memberships(T, pi, k).astype(np.float32)
    if chi_eff.shape[1] == n_macro:
        return active, chi_eff
    chi = np.zeros((active.shape[0], int(n_macro)), dtype=np.float32)
    chi[:, : chi_eff.shape[1]] = chi_eff
    rs = chi.sum(axis=1, keepdims=True)
    good = rs.squeeze(-1) > 0
    chi[good] /= rs[good]
    return active, chi
