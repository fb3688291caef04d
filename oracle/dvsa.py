"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement, in plain PyTorch fp32, of the reference's embedding + similarity + loss path:

  * ``vis_ebd``      <- VisEbd.forward            /root/reference/model.py:624-629
  * ``word_ebd``     <- WordEbd.forward           /root/reference/model.py:640-642
  * ``dvsa_forward`` <- DVSA.forward              /root/reference/model.py:517-614
  * ``postprocess``  <- postprocess               /root/reference/model.py:457-474

The restatement is vectorised (no Python triple loops) but keeps every quirk of the reference,
because parity means *bug-compatible*:

  Q1  masked columns (a, e >= len_a) of S_ are filled with exactly 0 in place (model.py:551), so
      they carry no gradient and their arg-max is index 0.
  Q2  the clustering term gathers ``vis_feats[maxind]`` with maxind in [0, Nb) and NO frame offset
      (model.py:562-569) -- i.e. always rows of frame 0 of segment 0.
  Q3  its denominator is ``count_nonzero`` of the masked (1 - G G^T) tensor (model.py:576-577).
  Q4  the min-max attention over the Ns frames of a segment carries gradient in the ranking term
      (model.py:587, the ``no_grad`` is commented out) but not in the clustering term (:556).
  Q5  EPS = 1e-5 (model.py:33); division vector uses max(len, 1) (model.py:502-507, :532).

Pinned against the imported reference by tests/golden/dvsa_*.npz (tests/test_oracle_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-5  # model.py:33


def vis_ebd(fc7, weight, bias, drop_mask=None, drop_scale=1.0):
    """model.py:624-629.  ``drop_mask`` (same shape as the output, 0/1) stands for nn.Dropout in
    train mode; None = eval / p=0."""
    x = fc7 / 100
    x = F.linear(x, weight, bias)
    if drop_mask is not None:
        x = x * drop_mask * drop_scale
    return torch.tanh(x)


def word_ebd(glove, weight, bias, bn_weight, bn_bias, running_mean=None, running_var=None,
             training=True, momentum=0.1, bn_eps=1e-5, drop_mask=None, drop_scale=1.0):
    """model.py:640-642: tanh(drop(bn(fc1(x)))).  BatchNorm1d in train mode uses the batch statistics
    over ALL Q rows, zero-padded slots included (model.py:730-749)."""
    x = F.linear(glove, weight, bias)
    x = F.batch_norm(x, running_mean, running_var, bn_weight, bn_bias, training, momentum, bn_eps)
    if drop_mask is not None:
        x = x * drop_mask * drop_scale
    return torch.tanh(x)


def dvsa_forward(vis_feats, word_feats, entities_length, Na, Nb, Ne, Delta, vis_lam, phase,
                 return_parts=False):
    """model.py:517-614.  vis_feats [Na*Ns*Nb, D], word_feats [Na*Ne, D] -> (D_ind i64 [F,Q],
    D_sim f32 [F,Q], margin_loss scalar)."""
    dev = vis_feats.device
    R, D = vis_feats.shape
    Ns = int(R / Na / Nb)                                              # model.py:530
    F_ = Na * Ns
    Q = Na * Ne
    lens = [int(x) for x in entities_length]
    div_vec = torch.tensor([l if l != 0 else 1 for l in lens], dtype=torch.float, device=dev)  # :532
    e_idx = torch.arange(Ne, device=dev)
    lens_t = torch.tensor(lens, device=dev)
    col_masked = (e_idx[None, :] >= lens_t[:, None])                    # [Na,Ne]  :534-538

    S_ = vis_feats @ word_feats.permute(1, 0)                           # :548
    S_ = S_.masked_fill(col_masked.view(1, Q), 0)                       # :551 (Q1)

    vis_loss = None
    parts = {}
    if phase == 'train':
        S5 = S_.view(Na, Ns, Nb, Na, Ne)                                # :555
        with torch.no_grad():
            ar = torch.arange(Na, device=dev)
            S_vis = S5[ar, :, :, ar, :]                                 # [Na,Ns,Nb,Ne] :557-559
            sim_scr, maxind = S_vis.max(2)                              # [Na,Ns,Ne]    :560
            indarr = maxind.reshape(-1)                                 # (Q2) :562-566
            mn = sim_scr.min(1, True)[0]
            mx = sim_scr.max(1, True)[0]
            sim_scr = (sim_scr - mn) / (mx - mn + EPS)                  # :567
            sim_scr = sim_scr.view(Na, Ns, Ne, 1)
        G = torch.index_select(vis_feats, 0, indarr).view(Na, Ns, Ne, -1)      # :569
        G = G / (torch.norm(G, 2, 3, True) + EPS)                       # :570
        G = G * sim_scr                                                 # :571
        G1 = G.permute(0, 2, 1, 3).contiguous().view(Na * Ne, Ns, -1)   # :572
        G2 = G.permute(0, 2, 3, 1).contiguous().view(Na * Ne, -1, Ns)   # :573
        M = 1 - torch.bmm(G1, G2).view(Na, Ne, Ns, Ns)                  # :574
        mask_vis = col_masked.view(Na, Ne, 1, 1).expand(Na, Ne, Ns, Ns).clone()   # :540-542
        eye = torch.eye(Ns, dtype=torch.bool, device=dev).view(1, 1, Ns, Ns)
        has_ent = (lens_t != 0).view(Na, 1, 1, 1)
        mask_vis = mask_vis | (eye & has_ent & (~col_masked).view(Na, Ne, 1, 1))  # :543-545
        M = M.masked_fill(mask_vis, 0)                                  # :575
        dem = int((M != 0).sum().item())                                # :576 (Q3)
        vis_loss = M.sum() / dem                                        # :577
        parts.update(vis_loss=vis_loss, dem=dem, sim_scr=sim_scr, maxind=maxind)

    S = S_.view(F_, Nb, Q)                                              # :580
    S, _ = S.max(1)                                                     # :583
    S = S.view(Na, Ns, Q)                                               # :585
    parts['S_max'] = S
    S_att = (S - S.min(1, True)[0]) / (S.max(1, True)[0] - S.min(1, True)[0] + EPS)  # :587 (Q4)
    S = S * S_att                                                       # :588
    Sf = S.view(Na, Ns, Na, Ne).sum(-1)                                 # :590
    Sf = Sf / div_vec                                                   # :592
    Sf_diag = torch.diagonal(Sf, dim1=0, dim2=2).permute(1, 0)          # [Na,Ns]  :594-597
    Sf_diag = Sf_diag.unsqueeze(2)                                      # :599
    frame_score = (F.relu(Sf - Sf_diag.permute(2, 1, 0) + Delta).mean(0).permute(1, 0)
                   + F.relu(Sf - Sf_diag + Delta).mean(2))              # :603
    if phase == 'train':
        margin_loss = (frame_score.mean() + vis_lam * vis_loss) * 10    # :606
    else:
        margin_loss = frame_score.mean() * 10
    S_sim = S_.view(F_, -1, Q)                                          # :610
    D_sim, D_ind = S_sim.max(1)                                         # :612
    parts.update(Sf=Sf, frame_score=frame_score)
    if return_parts:
        return D_ind, D_sim, margin_loss, parts
    return D_ind, D_sim, margin_loss


def postprocess(D, D_sim, Na, Ns, Nb, Ne):
    """model.py:457-474 (numpy): own-segment diagonal block + global box offsets."""
    D_t = np.asarray(D).reshape(Na, Ns, Na, Ne)
    S_t = np.asarray(D_sim).reshape(Na, Ns, Na, Ne)
    a = np.arange(Na)
    Dd = D_t[a, :, a, :].astype(int)                                    # [Na,Ns,Ne]
    Sd = S_t[a, :, a, :].astype(np.float64)
    off = (a[:, None, None] * Ns * Nb + np.arange(Ns)[None, :, None] * Nb)
    return Dd + off, Sd
