"""Model parameters of VCSMC.__init__ / get_Q / get_stationary_probs (vcsmc.py:110-148), evaluated on
the host in NumPy (16 + 4 + 2(N-1) doubles; negligible) and handed to the device by phylo_set_model."""
from __future__ import annotations

import numpy as np


def jc_Q(A=4):
    """vcsmc.py:126-129: off-diagonal 1/A, diagonal -(A-1)/A."""
    Q = np.zeros((A, A)) + 1 / A
    np.fill_diagonal(Q, -(A - 1) / A)
    return Q


def init_y_q(A=4):
    """vcsmc.py:122: the initial value of the 'Qmatrix' variable: 1/A with a zero diagonal."""
    y = np.zeros((A, A)) + 1 / A
    np.fill_diagonal(y, 0.0)
    return y


def get_Q(y_q):
    """vcsmc.py:138-148: off-diagonal = row-softmax of y_q (diagonal excluded), diagonal = -row sum."""
    y_q = np.asarray(y_q, dtype=np.float64)
    A = y_q.shape[0]
    e = np.exp(y_q)
    np.fill_diagonal(e, 0.0)
    denom = np.stack([e.sum(axis=1)] * A, axis=1)
    q_entry = e * (1 / denom)
    Q = q_entry.copy()
    np.fill_diagonal(Q, -q_entry.sum(axis=1))
    return Q


def get_stationary_probs(y_station):
    """vcsmc.py:133-136: softmax(y_station), shape [1, A]."""
    e = np.exp(np.asarray(y_station, dtype=np.float64))
    return np.expand_dims(e / e.sum(), axis=0)


def branch_rates(N, branch_prior):
    """vcsmc.py:119-120: exp(variable), variable initialised to branch_prior (runner.py:38-41)."""
    return np.exp(np.zeros(N - 1) + branch_prior)
