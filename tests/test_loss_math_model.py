"""CPU check of the hand-derived backward used by nafae_amd/csrc/simloss.hip (loss_tail_kernel,
cluster_kernel, sim_bwd_*): a numpy float64 transcription of the kernels' formulas is compared with
autograd through the oracle.  It guards the derivation; the kernels themselves are checked on the GPU
(tests/test_gpu_simloss.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import dvsa as O

EPS = 1e-5
G = os.path.join(os.path.dirname(__file__), "golden")


def kernel_model(V, W, lens, Na, Ns, Nb, Ne, Delta, lam, train):
    V = V.astype(np.float64); W = W.astype(np.float64)
    R, D = V.shape
    F, Q = Na * Ns, Na * Ne
    lens = np.asarray(lens)
    masked_q = (np.arange(Ne)[None, :] >= lens[:, None]).reshape(Q)
    # sim_max_kernel
    S_ = V @ W.T
    S_[:, masked_q] = 0
    S3 = S_.reshape(F, Nb, Q)
    Dind = S3.argmax(1)
    Smax = S3.max(1)
    # loss_tail_kernel
    Sm = Smax.reshape(Na, Ns, Q)
    mn, mx = Sm.min(1), Sm.max(1)                    # [Na,Q]
    amin, amax = Sm.argmin(1), Sm.argmax(1)
    den = mx - mn + EPS
    att = (Sm - mn[:, None]) / den[:, None]
    T = Sm * att
    div = np.where(lens == 0, 1, lens).astype(np.float64)
    Sf = T.reshape(Na, Ns, Na, Ne).sum(-1) / div     # [a,s,j]
    diag = np.stack([Sf[a, :, a] for a in range(Na)])  # [a,s]
    u1 = Sf - diag.T[None, :, :] + Delta             # [i,s,j] - diag[j,s]
    u2 = Sf - diag[:, :, None] + Delta               # [i,s,j] - diag[i,s]
    fs = np.maximum(u1, 0).mean(0).T + np.maximum(u2, 0).mean(2)
    rank = fs.mean()
    cN = 10.0 / (Na * Ns) / Na
    g1, g2 = (u1 > 0).astype(np.float64), (u2 > 0).astype(np.float64)
    dSf = cN * (g1 + g2)
    cs1 = g1.sum(0).T                                # [j,s]
    rs2 = g2.sum(2)                                  # [i,s]
    for a in range(Na):
        dSf[a, :, a] -= cN * (cs1[a] + rs2[a])
    dT = np.repeat(dSf / div, Ne, axis=2)            # [a,s,q]
    gmn = (dT * Sm * (Sm - mx[:, None] - EPS) / den[:, None] ** 2).sum(1)
    gmx = -(dT * Sm * (Sm - mn[:, None]) / den[:, None] ** 2).sum(1)
    dS = dT * (att + Sm / den[:, None])
    for a in range(Na):
        for q in range(Q):
            dS[a, amin[a, q], q] += gmn[a, q]
            dS[a, amax[a, q], q] += gmx[a, q]
    dS[:, :, masked_q] = 0
    dS = dS.reshape(F, Q)
    vis = 0.0
    dV = np.zeros_like(V)
    if train:
        tot_sum, tot_cnt = 0.0, 0
        entries = []
        for a in range(Na):
            for e in range(Ne):
                if e >= lens[a]:
                    continue
                q = a * Ne + e
                sn = att[a, :, q]
                idx = Dind.reshape(Na, Ns, Q)[a, :, q]
                g = V[idx]                                        # rows [0,Nb): frame-0 quirk
                n = np.linalg.norm(g, axis=1)
                Gm = g / (n + EPS)[:, None] * sn[:, None]
                M = 1 - Gm @ Gm.T
                np.fill_diagonal(M, 0)
                tot_sum += M.sum(); tot_cnt += int((M != 0).sum())
                dG = -2 * (Gm.sum(0)[None, :] - Gm)
                dot = (g * dG).sum(1)
                dg = sn[:, None] * (dG / (n + EPS)[:, None] - g * (dot / (n * (n + EPS) ** 2))[:, None])
                entries.append((idx, dg))
        vis = tot_sum / tot_cnt
        cscale = 10.0 * lam / tot_cnt
        for idx, dg in entries:
            for s in range(Ns):
                dV[idx[s]] += cscale * dg[s]
    loss = 10 * (rank + lam * vis) if train else 10 * rank
    # sim_bwd
    dW = np.zeros_like(W)
    for f in range(F):
        for q in range(Q):
            r = f * Nb + Dind[f, q]
            dV[r] += dS[f, q] * W[q]
            dW[q] += dS[f, q] * V[r]
    return loss, Dind, Smax, dV, dW


@pytest.mark.parametrize("name", ["c1b", "ragged", "na1", "full", "big"])
@pytest.mark.parametrize("train", [True, False])
def test_kernel_formulas_match_autograd(name, train):
    g = np.load(os.path.join(G, "dvsa_%s.npz" % name))
    Na, Ns, Nb, Ne, D = [int(x) for x in g["shape"]]
    phase = "train" if train else "eval"
    loss, Dind, Smax, dV, dW = kernel_model(g["V"], g["W"], g["lens"], Na, Ns, Nb, Ne, float(g["Delta"]),
                                            float(g["vis_lam"]), train)
    assert np.array_equal(Dind, g["D_ind_" + phase])
    np.testing.assert_allclose(Smax, g["D_sim_" + phase], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(loss, float(g["loss_" + phase]), rtol=1e-5)
    np.testing.assert_allclose(dV, g["dV_" + phase], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(dW, g["dW_" + phase], rtol=2e-4, atol=2e-6)
