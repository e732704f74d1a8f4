"""Neighbour-list scheduling for cp_edgeconv_fused (init-time, host): the K-way max over a keypoint's neighbours does not depend
on their order, but the kernel's LDS traffic does.  At step k the 16 keypoints of a row fragment each read one row of the P' table
(16-byte pieces, plane layout: bank slot = row mod 16), so the 16 rows of a step should have 16 different residues mod 16.  Per
block of 16 consecutive keypoints and per step, a bipartite matching keypoint <-> residue picks, for as many keypoints as possible,
an unused neighbour whose residue nobody else reads in that step (measured: SQ_LDS_BANK_CONFLICT share of the kernel 54 % -> see
DESIGN.md).  The result is a permutation of every keypoint's list: same graph, same max, bit-identical output."""
import numpy as np


def _match(cands):
    """cands[i] = set of residues keypoint i can take; returns {i: residue} of a maximum matching (augmenting paths)"""
    owner = {}

    def try_(i, seen):
        for r in cands[i]:
            if r in seen:
                continue
            seen.add(r)
            if r not in owner or try_(owner[r], seen):
                owner[r] = i
                return True
        return False

    for i in sorted(range(len(cands)), key=lambda i: len(cands[i])):
        try_(i, set())
    return {i: r for r, i in owner.items()}


def schedule_block(rows):
    """rows: (16, K) int array of table rows; returns the reordered (16, K) array and the number of (step, lane) reads that share a
    residue with an earlier lane of their step"""
    n, K = rows.shape
    left = [list(map(int, rows[i])) for i in range(n)]
    out = np.empty_like(rows)
    clashes = 0
    for k in range(K):
        cands = [sorted({r & 15 for r in left[i]}, key=lambda res, i=i: -sum(1 for r in left[i] if (r & 15) == res)) for i in range(n)]
        m = _match(cands)
        used = set(m.values())
        for i in range(n):
            if i in m:
                pick = next(r for r in left[i] if (r & 15) == m[i])
            else:                                  # no free residue: any neighbour (a bank conflict in this step)
                pick = left[i][0]
                clashes += 1
            left[i].remove(pick)
            out[i, k] = pick
    return out, clashes


def schedule_neighbours(idx):
    """idx: (G, N, K) integer array of neighbour rows (N a multiple of 16).  Returns (reordered int32 array, clashes before, after)."""
    idx = np.asarray(idx)
    G, N, K = idx.shape
    assert N % 16 == 0
    out = np.empty((G, N, K), dtype=np.int32)
    before = after = 0
    for g in range(G):
        for t in range(N // 16):
            blk = idx[g, 16 * t:16 * t + 16]
            for k in range(K):
                res = blk[:, k] & 15
                before += 16 - len(set(res.tolist()))
            o, c = schedule_block(blk)
            out[g, 16 * t:16 * t + 16] = o
            after += c
    return out, before, after
