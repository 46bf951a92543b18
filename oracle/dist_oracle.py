"""NumPy restatement of the row-block-cyclic sharded schedule.  TEST INFRASTRUCTURE ONLY.

Mirrors ``gaussianprocesspathmodelling_amd/csrc/gpx_shard.inc`` step for step (owner
factors the diagonal block, broadcast, local panel solve, padded all-gather + un-permute,
local trailing update; forward solve with a broadcast per panel, backward solve with a
reduce per panel; K*, mean and variance partials + all-reduce) with NumPy/SciPy blocks and
``torch.distributed`` (gloo) collectives, so the schedule's arithmetic can be checked
against the single-process oracle on CPU with world_size > 1.  The reference has no
distributed code at all; the algorithm is SURVEY.md §8(e).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import cholesky, solve_triangular

from oracle.gp_oracle import kernel_matrix


class NumpyCollectives:
    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def bcast(self, arr, root):
        t = self.torch.from_numpy(arr)
        self.dist.broadcast(t, src=root, group=self.group)
        return arr

    def allgather(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr))
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t, group=self.group)
        return [o.numpy() for o in outs]

    def reduce_sum(self, arr, root):
        t = self.torch.from_numpy(arr.copy())
        self.dist.reduce(t, dst=root, group=self.group)
        return t.numpy() if self.rank == root else None

    def allreduce_sum(self, arr):
        t = self.torch.from_numpy(arr)
        self.dist.all_reduce(t, group=self.group)
        return arr


# the dealing of row blocks over the ranks: snake (rounds of 2 P blocks dealt 0 .. P-1, P-1 .. 0; gpx_internal.h: Deal) or
# cyclic (block g on rank g mod P).  Module-level switch so that the schedule below reads as before.
SNAKE = True


def _owner(g, P):
    if not SNAKE:
        return g % P
    pos = g % (2 * P)
    return pos if pos < P else 2 * P - 1 - pos


def _local(g, P):
    if not SNAKE:
        return g // P
    return 2 * (g // (2 * P)) + (1 if g % (2 * P) >= P else 0)


def _lb0(p, r, P):
    if not SNAKE:
        return (p - r) // P + 1 if p >= r else 0
    c, rem = divmod(p + 1, 2 * P)
    return 2 * c + (1 if rem > r else 0) + (1 if rem > 2 * P - 1 - r else 0)


def sharded_fit_predict(coll, X, y, Xs, kernel, ls, sf2, sn2, nb, one_pass=False):
    """one_pass (gpx_fit_predict on a shard, gpx_shard.inc shard_fit with query points + shard_fused_tail): rank r's slice
    of the query points — Mc = ceil(M / P) rounded up to 128 per rank — rides through the factorisation as bordered rows
    B = K(Xs_r, X): every panel (the LAST one too: one more broadcast) solves them with the diagonal block and updates all
    their remaining columns; they end as V_r^T.  mean_r = V_r^T z with z = L^-1 y replicated, var_r = sf2 - row norms;
    one all-gather of the slices."""
    P, r = coll.world, coll.rank
    N, M = len(X), len(Xs)
    Npad = -(-N // nb) * nb
    nblk = Npad // nb
    own = [g for g in range(nblk) if _owner(g, P) == r]
    nloc = len(own) * nb
    Xp = np.zeros((Npad, X.shape[1]))
    Xp[:N] = X
    # local rows of K (identity padding), columns 0..Npad
    A = np.zeros((max(nloc, 1), Npad))
    for lb, g in enumerate(own):
        rows = slice(g * nb, (g + 1) * nb)
        blk = kernel_matrix(Xp[rows], Xp, kernel, ls, sf2)
        gi = np.arange(g * nb, (g + 1) * nb)
        blk[gi >= N, :] = 0.0
        blk[:, N:] = 0.0
        blk[np.arange(nb), gi] += np.where(gi < N, sn2, 1.0)
        A[lb * nb:(lb + 1) * nb] = blk
    if one_pass:
        Mc = -(-(-(-M // P)) // 128) * 128
        q0, q1 = min(M, r * Mc), min(M, (r + 1) * Mc)
        B = kernel_matrix(Xs[q0:q1], Xp, kernel, ls, sf2) if q1 > q0 else np.zeros((0, Npad))
        B[:, N:] = 0.0
    logdet = np.zeros(1)
    for p in range(nblk):
        o, root, lo = p * nb, _owner(p, P), _local(p, P) * nb
        D = np.zeros((nb, nb))
        if r == root:
            D[:] = cholesky(A[lo:lo + nb, o:o + nb], lower=True)
            A[lo:lo + nb, o:o + nb] = D
            logdet += 2.0 * np.sum(np.log(np.diag(D)))
        if p + 1 == nblk and not one_pass:
            break
        coll.bcast(D, root)
        if one_pass:  # (a collective decision: every rank takes the last block too, also with an empty slice)
            B[:, o:o + nb] = solve_triangular(D, B[:, o:o + nb].T, lower=True).T
        if p + 1 == nblk:
            break
        l0 = _lb0(p, r, P)
        rows = nloc - l0 * nb
        if rows > 0:
            A[l0 * nb:nloc, o:o + nb] = solve_triangular(D, A[l0 * nb:nloc, o:o + nb].T, lower=True).T
        maxcnt = max((_lb0(nblk - 1, rr, P) - _lb0(p, rr, P)) * nb for rr in range(P))
        send = np.zeros((maxcnt, nb))
        send[:rows] = A[l0 * nb:nloc, o:o + nb]
        parts = coll.allgather(send)
        Pg = np.zeros(((nblk - p - 1) * nb, nb))
        for b in range(nblk - p - 1):
            g = p + 1 + b
            rr = _owner(g, P)
            idx = _local(g, P) - _lb0(p, rr, P)
            Pg[b * nb:(b + 1) * nb] = parts[rr][idx * nb:(idx + 1) * nb]
        for lb in range(l0, len(own)):
            g = own[lb]
            cols = slice((p + 1) * nb, (g + 1) * nb)
            A[lb * nb:(lb + 1) * nb, cols] -= A[lb * nb:(lb + 1) * nb, o:o + nb] @ Pg[:(g - p) * nb].T
        if one_pass:  # the bordered rows: blocks beyond the matrix, every remaining column
            B[:, (p + 1) * nb:] -= B[:, o:o + nb] @ Pg.T
    coll.allreduce_sum(logdet)

    def local(v):  # (Npad, k) -> local rows
        return np.concatenate([v[g * nb:(g + 1) * nb] for g in own]) if own else np.zeros((0, v.shape[1]))

    Yp = np.zeros((Npad, 1))
    Yp[:N, 0] = y
    z = local(Yp)
    for p in range(nblk):                      # forward: broadcast the solved block
        o, root, lo = p * nb, _owner(p, P), _local(p, P) * nb
        S = np.zeros((nb, 1))
        if r == root:
            z[lo:lo + nb] = solve_triangular(A[lo:lo + nb, o:o + nb], z[lo:lo + nb], lower=True)
            S[:] = z[lo:lo + nb]
        if p + 1 == nblk:
            break
        coll.bcast(S, root)
        l0 = _lb0(p, r, P)
        if nloc - l0 * nb > 0:
            z[l0 * nb:] -= A[l0 * nb:nloc, o:o + nb] @ S
    if one_pass:
        z_full = np.zeros((Npad, 1))
        for lb, g in enumerate(own):
            z_full[g * nb:(g + 1) * nb] = z[lb * nb:(lb + 1) * nb]
        coll.allreduce_sum(z_full)
    cneg = np.zeros((Npad, 1))
    for p in range(nblk - 1, -1, -1):          # backward: reduce the partial products
        o, root, lo = p * nb, _owner(p, P), _local(p, P) * nb
        if p + 1 < nblk:
            red = coll.reduce_sum(cneg[o:o + nb], root)
        if r == root:
            if p + 1 < nblk:
                z[lo:lo + nb] += red
            z[lo:lo + nb] = solve_triangular(A[lo:lo + nb, o:o + nb], z[lo:lo + nb], lower=True, trans="T")
            cneg[:o] -= A[lo:lo + nb, :o].T @ z[lo:lo + nb]
    alpha_full = np.zeros((Npad, 1))
    for lb, g in enumerate(own):
        alpha_full[g * nb:(g + 1) * nb] = z[lb * nb:(lb + 1) * nb]
    coll.allreduce_sum(alpha_full)

    if one_pass:
        send = np.zeros((Mc, 2))
        send[:q1 - q0, 0] = (B @ z_full)[:, 0]
        send[:q1 - q0, 1] = sf2 - np.einsum("ij,ij->i", B, B)
        parts = coll.allgather(send)
        mean, var = np.zeros(M), np.zeros(M)
        for rr in range(P):
            a0, a1 = min(M, rr * Mc), min(M, (rr + 1) * Mc)
            mean[a0:a1], var[a0:a1] = parts[rr][:a1 - a0, 0], parts[rr][:a1 - a0, 1]
        return mean, var, alpha_full[:N, 0], float(logdet[0])
    Ks = np.zeros((M, max(nloc, 0)))
    for lb, g in enumerate(own):
        blk = kernel_matrix(Xs, Xp[g * nb:(g + 1) * nb], kernel, ls, sf2)
        blk[:, np.arange(g * nb, (g + 1) * nb) >= N] = 0.0
        Ks[:, lb * nb:(lb + 1) * nb] = blk
    mean = Ks @ z if nloc else np.zeros((M, 1))
    coll.allreduce_sum(mean)
    V = Ks.copy()
    for p in range(nblk):
        o, root, lo = p * nb, _owner(p, P), _local(p, P) * nb
        S = np.zeros((M, nb))
        if r == root:
            V[:, lo:lo + nb] = solve_triangular(A[lo:lo + nb, o:o + nb], V[:, lo:lo + nb].T, lower=True).T
            S[:] = V[:, lo:lo + nb]
        if p + 1 == nblk:
            break
        coll.bcast(S, root)
        l0 = _lb0(p, r, P)
        if nloc - l0 * nb > 0:
            V[:, l0 * nb:] -= S @ A[l0 * nb:nloc, o:o + nb].T
    part = -np.einsum("ij,ij->i", V, V)
    coll.allreduce_sum(part)
    return mean[:, 0], sf2 + part, alpha_full[:N, 0], float(logdet[0])
