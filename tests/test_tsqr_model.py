"""CPU model of the device's Householder compression (ekf_linalg.hip: k_ekf_tsqr, tsqr_wave): the row-block TSQR with the
annihilated-column skipping rule, and the tree over four partitions the fused small update uses.  What the update needs from the
compression of msckf_vio.cpp:795-817 is R^T R = H^T H and R^T (Q^T r) = H^T r; both orders of reduction must
deliver it to rounding, also for stacks with structurally zero columns and fewer rows in a block than columns."""
import numpy as np
import pytest


def reduce_block(R, B):
    """Annihilate the row block B (rows x n1) against the resident upper-triangular R (n1 x n1) with reflectors of length rows + 1."""
    n1 = R.shape[0]
    B = B.copy()
    for k in range(n1):
        ss = float(B[:, k] @ B[:, k])
        x0 = R[k, k]
        if ss < 1e-200 or ss < 1e-40 * x0 * x0:      # the device's rule: nothing (left) in this column of the block
            continue
        nrm = np.sqrt(x0 * x0 + ss)
        alpha = -nrm if x0 > 0.0 else nrm
        v0 = x0 - alpha
        beta = 2.0 / (v0 * v0 + ss)
        v = B[:, k].copy()
        for j in range(k + 1, n1):
            t = beta * (v @ B[:, j] + v0 * R[k, j])
            B[:, j] -= t * v
            R[k, j] -= t * v0
        R[k, k] = alpha
        B[:, k] = 0.0
    return R


def tsqr_rows(H, block):
    R = np.zeros((H.shape[1], H.shape[1]))
    for k0 in range(0, H.shape[0], block):
        reduce_block(R, H[k0:k0 + block])
    return R


def tsqr_tree(H, block, parts=4):
    Rs = [np.zeros((H.shape[1], H.shape[1])) for _ in range(parts)]
    blocks = [H[k0:k0 + block] for k0 in range(0, H.shape[0], block)]
    for i, b in enumerate(blocks):                    # wavefront w takes the blocks w, w + 4, ..
        reduce_block(Rs[i % parts], b)
    for w in range(1, parts):                         # wavefront 0 annihilates the other R's against its own
        reduce_block(Rs[0], np.triu(Rs[w]))
    return Rs[0]


def stack(rng, m, clones, feats):
    """A stacked [H | r] with the block structure of the filter: a feature's rows touch the columns of the clones that saw it."""
    n = 6 * clones
    H = np.zeros((m, n + 1))
    row = 0
    while row < m:
        k = int(rng.integers(2, min(clones, feats) + 1))
        c0 = int(rng.integers(0, clones - k + 1))
        rows = min(4 * k - 3, m - row)
        H[row:row + rows, 6 * c0:6 * (c0 + k)] = rng.standard_normal((rows, 6 * k))
        H[row:row + rows, n] = rng.standard_normal(rows)
        row += rows
    return H


@pytest.mark.parametrize("m,clones,block", [(760, 25, 128), (300, 30, 128), (90, 4, 128), (1500, 2, 128), (700, 50, 64)])
def test_row_block_tsqr_and_tree_give_the_gram_factor(m, clones, block):
    rng = np.random.default_rng(1234 + m + clones)
    H = stack(rng, m, clones, feats=12)
    G = H.T @ H
    scale = np.abs(G).max()
    for R in (tsqr_rows(H, block), tsqr_tree(H, block)):
        assert np.allclose(np.tril(R, -1), 0.0)
        assert np.abs(R.T @ R - G).max() <= 1e-12 * scale
    # the two reduction orders agree up to the signs of rows (a row of R and its entry of Q^T r flip together)
    Ra, Rb = tsqr_rows(H, block), tsqr_tree(H, block)
    nz = np.abs(np.diag(Ra)) > 1e-9 * np.sqrt(scale)
    sg = np.sign(np.diag(Ra))[nz] * np.sign(np.diag(Rb))[nz]
    assert np.abs(Ra[nz] - sg[:, None] * Rb[nz]).max() <= 1e-9 * np.sqrt(scale)


def test_update_is_the_same_with_compressed_and_uncompressed_measurement():
    rng = np.random.default_rng(7)
    clones, m, sigma2 = 6, 200, 0.035 ** 2
    Hr = stack(rng, m, clones, feats=6)
    H, r = Hr[:, :-1], Hr[:, -1]
    n = H.shape[1]
    A = rng.standard_normal((n, n))
    P = A @ A.T / n + 0.01 * np.eye(n)
    R1 = tsqr_tree(Hr, 128)
    Rc, qr = R1[:n, :n], R1[:n, n]
    def upd(Hm, rm):
        S = Hm @ P @ Hm.T + sigma2 * np.eye(Hm.shape[0])
        K = np.linalg.solve(S, Hm @ P).T
        return K @ rm, P - K @ Hm @ P
    dx0, P0 = upd(H, r)
    dx1, P1 = upd(Rc, qr)
    assert np.abs(dx0 - dx1).max() <= 1e-10 * max(1.0, np.abs(dx0).max())
    assert np.abs(P0 - P1).max() <= 1e-12 * np.abs(P).max()
