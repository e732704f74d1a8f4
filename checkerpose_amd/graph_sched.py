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


# ---------------------------------------------------------------------------------------------------------------------------
# Tiling of large graphs for cp_edgeconv_tiled (N > 512 keypoints: the P' table of a crop does not fit in one CU's LDS).
# kNN graphs of FPS keypoints are spatially local but FPS ORDER is not (consecutive samples are far apart), so the keypoints are
# renumbered: recursive median bisection along the widest coordinate axis gives N / 512 compact patches of exactly 512 keypoints;
# a patch's neighbours lie in the patch or in a thin rim around it (LM, 4096 keypoints: 170-370 rim rows per patch, 1.4 N rows in
# total).  A workgroup owns one patch: its LDS table holds the patch's 512 rows plus the rim ("halo") rows, and every neighbour
# list is rewritten in table-slot numbers.  The renumbering is internal to the launch program (rows are permuted once behind
# conv1x1 and un-permuted at the logits), so the boundary -- keypoint order of every input and output -- is unchanged.
def _kd_blocks(xyz, bs):
    def rec(ids):
        if len(ids) <= bs:
            return [ids]
        p = xyz[ids]
        ax = int(np.argmax(p.max(0) - p.min(0)))
        o = ids[np.argsort(p[:, ax], kind="stable")]
        h = len(ids) // 2
        h = (h + bs - 1) // bs * bs
        return rec(o[:h]) + rec(o[h:])
    return rec(np.arange(len(xyz)))


def tile_schedule(idx, p3d, block=512, hpad_unit=128, hpad_max=768):
    """idx (G, N, K) neighbour table in ORIGINAL keypoint numbering, p3d (G, 3, N) coordinates.  Returns None when N is not a
    multiple of `block` or some patch needs more than `hpad_max` halo rows; else a dict:
      perm (G, N) int32   internal row i holds original keypoint perm[g, i]        inv (G, N) int32   its inverse
      halo (G, NB, HPAD) int32   INTERNAL row ids of each patch's halo rows (padded with the patch's first own row)
      nbr  (G, NB, block, K) int16   neighbour lists as table slots: own row j -> j, halo row h -> block + h; every 16-keypoint
                                     fragment's lists ordered against LDS bank conflicts (schedule_block)
      idx_internal (G, N, K) int32   the same graph in internal numbering (for kernels that gather through L2)
      HPAD, NB, halo_rows (G, NB) actual halo sizes"""
    idx = np.asarray(idx).astype(np.int64)
    xyz_all = np.asarray(p3d, dtype=np.float64)
    G, N, K = idx.shape
    if N % block or N <= block:
        return None
    NB = N // block
    perm = np.empty((G, N), np.int32)
    inv = np.empty((G, N), np.int32)
    halos, nbrs, hsz = [], [], np.zeros((G, NB), np.int32)
    for g in range(G):
        blocks = _kd_blocks(xyz_all[g].T, block)
        assert len(blocks) == NB and all(len(b) == block for b in blocks)
        perm[g] = np.concatenate(blocks)
        inv[g][perm[g]] = np.arange(N, dtype=np.int32)
        ii = inv[g][idx[g][perm[g]]].astype(np.int64)            # (N, K) internal -> internal
        hg, ng = [], []
        for t in range(NB):
            own = ii[t * block:(t + 1) * block]                   # (block, K)
            out_of = np.unique(own[(own < t * block) | (own >= (t + 1) * block)])
            hsz[g, t] = len(out_of)
            slot = {int(r): block + j for j, r in enumerate(out_of)}
            loc = np.where((own >= t * block) & (own < (t + 1) * block), own - t * block, 0)
            far = (own < t * block) | (own >= (t + 1) * block)
            if far.any():
                loc[far] = np.vectorize(slot.get)(own[far])
            hg.append(out_of)
            ng.append(loc)
        halos.append(hg)
        nbrs.append(ng)
    hmax = int(hsz.max())
    HPAD = max((hmax + hpad_unit - 1) // hpad_unit * hpad_unit, hpad_unit)
    if HPAD > hpad_max:
        return None
    halo = np.empty((G, NB, HPAD), np.int32)
    nbr = np.empty((G, NB, block, K), np.int16)
    for g in range(G):
        for t in range(NB):
            halo[g, t] = t * block
            halo[g, t, :hsz[g, t]] = halos[g][t]
            loc = nbrs[g][t]
            for f in range(block // 16):
                loc[16 * f:16 * f + 16], _ = schedule_block(loc[16 * f:16 * f + 16])
            nbr[g, t] = loc.astype(np.int16)
    idx_int = np.stack([inv[g][idx[g][perm[g]]] for g in range(G)]).astype(np.int32)
    return dict(perm=perm, inv=inv, halo=halo, nbr=nbr, idx_internal=idx_int, HPAD=HPAD, NB=NB, halo_rows=hsz, block=block)
