"""CPU stand-ins (plain torch autograd) for the three HIP calls of the frame-sharded DVSA, for the gloo world-size-2 test
of the exchange protocol in nafae_amd/parallel.py.  TEST INFRASTRUCTURE ONLY: the product never imports this; it is
validated against oracle.dvsa.dvsa_forward in tests/test_dp_gloo.py before it is trusted."""
import torch
import torch.nn.functional as F

EPS = 1e-5


def _tail(S_max, D_ind, V0, lens, Na, Ns, Nb, Ne, Delta, vis_lam, train):
    """model.py:553-606 restated from S_max / D_ind (what the loss-tail kernels consume) instead of from V, W."""
    Q = Na * Ne
    lens_t = torch.tensor(lens)
    div_vec = torch.tensor([l if l != 0 else 1 for l in lens], dtype=torch.float)
    col_masked = torch.arange(Ne)[None, :] >= lens_t[:, None]
    S = S_max.view(Na, Ns, Q)
    vis_loss = torch.zeros(())
    dem = 0
    if train:
        ar = torch.arange(Na)
        with torch.no_grad():
            own = S.view(Na, Ns, Na, Ne)[ar, :, ar, :]                       # [Na,Ns,Ne] = max_b of the own-segment block
            maxind = D_ind.view(Na, Ns, Na, Ne)[ar, :, ar, :]
            mn, mx = own.min(1, True)[0], own.max(1, True)[0]
            sim_scr = ((own - mn) / (mx - mn + EPS)).view(Na, Ns, Ne, 1)
        G = torch.index_select(V0, 0, maxind.reshape(-1)).view(Na, Ns, Ne, -1)   # rows [0, Nb): frame-0 quirk
        G = G / (torch.norm(G, 2, 3, True) + EPS) * sim_scr
        G1 = G.permute(0, 2, 1, 3).contiguous().view(Na * Ne, Ns, -1)
        G2 = G.permute(0, 2, 3, 1).contiguous().view(Na * Ne, -1, Ns)
        M = 1 - torch.bmm(G1, G2).view(Na, Ne, Ns, Ns)
        mask_vis = col_masked.view(Na, Ne, 1, 1).expand(Na, Ne, Ns, Ns).clone()
        eye = torch.eye(Ns, dtype=torch.bool).view(1, 1, Ns, Ns)
        mask_vis = mask_vis | (eye & (lens_t != 0).view(Na, 1, 1, 1) & (~col_masked).view(Na, Ne, 1, 1))
        M = M.masked_fill(mask_vis, 0)
        dem = int((M != 0).sum())
        vis_loss = M.sum() / dem
    S_att = (S - S.min(1, True)[0]) / (S.max(1, True)[0] - S.min(1, True)[0] + EPS)
    T = S * S_att
    Sf = T.view(Na, Ns, Na, Ne).sum(-1) / div_vec
    Sd = torch.diagonal(Sf, dim1=0, dim2=2).permute(1, 0).unsqueeze(2)
    fs = F.relu(Sf - Sd.permute(2, 1, 0) + Delta).mean(0).permute(1, 0) + F.relu(Sf - Sd + Delta).mean(2)
    rank = fs.mean()
    loss = (rank + vis_lam * vis_loss) * 10 if train else rank * 10
    return loss, rank, vis_loss, dem


class CpuKernels:
    """Same call signatures as nafae_amd.ops.{sim_max_fwd_frames, loss_fwd_bwd, sim_bwd_frames}."""

    @staticmethod
    def sim_max_fwd_frames(V, W, ent_len, Nb, Na, Ne, lens=None, exact_fp32=False):
        Q = Na * Ne
        masked = (torch.arange(Ne)[None, :] >= ent_len.long()[:, None]).view(1, Q)
        S_ = (V.detach() @ W.detach().t()).masked_fill(masked, 0)
        S_max, D_ind = S_.view(-1, Nb, Q).max(1)
        return S_max.contiguous(), D_ind.contiguous()

    @staticmethod
    def loss_fwd_bwd(S_max, D_ind, V0, ent_len, Na, Ns, Nb, Ne, Delta, vis_lam, train, need_grad=True, lens=None):
        lens = [int(x) for x in ent_len]
        with torch.enable_grad():        # (called from inside an autograd.Function.forward, where grad mode is off)
            S = S_max.detach().clone().requires_grad_(True)
            V0g = V0.detach().clone().requires_grad_(True)
            loss, rank, vis, dem = _tail(S, D_ind, V0g, lens, Na, Ns, Nb, Ne, Delta, vis_lam, train)
            loss.backward()
        out = torch.stack([loss.detach(), rank.detach(), vis.detach(), torch.tensor(float(dem))])
        ws = V0g.grad if V0g.grad is not None else torch.zeros_like(V0)      # "workspace": the clustering gradient rows
        return out, S.grad, ws

    @staticmethod
    def sim_bwd_frames(dS, D_ind, V, W, ent_len, Na, Ns, Nb, Ne, cluster_rows, ws, pre_scale=None, grad_scale=None):
        Fl, Q = dS.shape
        rows = (torch.arange(Fl)[:, None] * Nb + D_ind).reshape(-1)            # arg-max row of every (frame, query)
        dV = torch.zeros_like(V).index_add_(0, rows, (dS.reshape(-1, 1) * W.detach().repeat(Fl, 1)))
        if cluster_rows:
            dV[:Nb] += ws
        dW = (dS.unsqueeze(2) * V.detach()[rows].view(Fl, Q, -1)).sum(0)
        g = 1.0 if grad_scale is None else float(grad_scale)
        return dV * g, dW * g
