"""world_size-2 data-parallel test on CPU (gloo): the flat-buffer gradient all-reduce of nafae_amd/parallel.py.
The HIP ops cannot run here, so the two trainable sub-modules are driven by plain autograd on CPU; what is under
test is the N > 1 plumbing: grads are views of one flat buffer, one all-reduce averages them, ranks stay in sync."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.vis_ebd = torch.nn.Module()
        self.vis_ebd.fc1 = torch.nn.Linear(12, 8)
        self.word_ebd = torch.nn.Module()
        self.word_ebd.fc1 = torch.nn.Linear(6, 8)
        self.word_ebd.bn = torch.nn.BatchNorm1d(8)
        self.frozen = torch.nn.Linear(3, 3)
        for p in self.frozen.parameters():
            p.requires_grad = False


def _loss(m, x, g):
    v = torch.tanh(m.vis_ebd.fc1(x))
    w = torch.tanh(m.word_ebd.bn(m.word_ebd.fc1(g)))
    return (v @ w.t()).max(0)[0].sum()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nafae_amd.parallel import GradAllReducer, broadcast_parameters, trainable_parameters
    torch.manual_seed(100 + rank)            # ranks start DIFFERENT on purpose
    m = _Toy()
    broadcast_parameters(m, src=0)
    red = GradAllReducer(trainable_parameters(m))
    assert red.flat.numel() == sum(p.numel() for p in trainable_parameters(m))
    assert all(p.grad.data_ptr() >= red.flat.data_ptr() for p in trainable_parameters(m))
    opt = torch.optim.Adam(trainable_parameters(m), lr=1e-2)
    g = torch.Generator().manual_seed(7 + rank)   # each rank: its own shard of segments
    xs = [torch.randn(10, 12, generator=g) for _ in range(3)]
    gs = [torch.randn(5, 6, generator=g) for _ in range(3)]
    local_grads = []
    for x, gl in zip(xs, gs):
        red.zero_grad()
        _loss(m, x, gl).backward()
        local_grads.append(red.flat.clone())
        red.allreduce()
        # reduced gradient == mean over ranks of the local gradients
        gathered = [torch.zeros_like(red.flat) for _ in range(world)]
        dist.all_gather(gathered, local_grads[-1])
        assert torch.allclose(red.flat, sum(gathered) / world, atol=1e-6)
        assert m.vis_ebd.fc1.weight.grad.data_ptr() == red.flat.data_ptr()     # still a view after backward
        opt.step()
    flat_params = torch.cat([p.detach().reshape(-1) for p in trainable_parameters(m)])
    gathered = [torch.zeros_like(flat_params) for _ in range(world)]
    dist.all_gather(gathered, flat_params)
    assert torch.equal(gathered[0], gathered[1])                                # replicas stay bit-identical
    if rank == 0:
        torch.save({"ok": True, "nbytes": red.nbytes}, out)
    dist.destroy_process_group()


def test_grad_allreduce_world2(tmp_path):
    out = str(tmp_path / "res.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["ok"] and r["nbytes"] == 4 * (12 * 8 + 8 + 6 * 8 + 8 + 8 + 8)


def test_reducer_single_process_is_noop():
    from nafae_amd.parallel import GradAllReducer, trainable_parameters
    m = _Toy()
    red = GradAllReducer(trainable_parameters(m))
    _loss(m, torch.randn(4, 12), torch.randn(3, 6)).backward()
    before = red.flat.clone()
    red.allreduce()
    assert torch.equal(before, red.flat) and red.world == 1


# ---------------------------------------------------------------------------------------------------- exact mode
def _exact_case():
    Na, Ns, Nb, Ne, D = 4, 4, 6, 3, 16
    g = torch.Generator().manual_seed(5)
    V = torch.tanh(torch.randn(Na * Ns * Nb, D, generator=g))
    W = torch.tanh(torch.randn(Na * Ne, D, generator=g))
    return Na, Ns, Nb, Ne, V, W, [2, 0, 3, 1]


class _FakeDVSA:
    def __init__(self, Na, Ne):
        import argparse
        self.Na, self.phase = Na, 'train'
        self.args = argparse.Namespace(max_ent_len=Ne, Delta=10.0, vis_lam=4.13)


def test_cpu_tail_standin_matches_oracle():
    """The torch stand-in used below (tests/cpu_kernels.py) == the oracle's DVSA.forward, loss and both gradients."""
    from oracle import dvsa as O
    from tests.cpu_kernels import CpuKernels as K
    Na, Ns, Nb, Ne, V, W, lens = _exact_case()
    Vr, Wr = V.clone().requires_grad_(True), W.clone().requires_grad_(True)
    Di, Ds, L = O.dvsa_forward(Vr, Wr, lens, Na, Nb, Ne, 10.0, 4.13, 'train')
    L.backward()
    ent = torch.tensor(lens, dtype=torch.int32)
    S, Dk = K.sim_max_fwd_frames(V, W, ent, Nb, Na, Ne)
    out, dS, ws = K.loss_fwd_bwd(S, Dk, V[:Nb].contiguous(), ent, Na, Ns, Nb, Ne, 10.0, 4.13, True)
    dV, dW = K.sim_bwd_frames(dS, Dk, V, W, ent, Na, Ns, Nb, Ne, True, ws)
    assert torch.equal(Dk, Di) and torch.allclose(S, Ds) and abs(float(out[0]) - float(L)) < 1e-5 * abs(float(L))
    assert torch.allclose(dV, Vr.grad, atol=1e-6) and torch.allclose(dW, Wr.grad, atol=1e-6)


def _exact_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nafae_amd.config import cfg, reset_cfg
    from nafae_amd.parallel import GradAllReducer, dvsa_frame_sharded
    from tests.cpu_kernels import CpuKernels
    Na, Ns, Nb, Ne, V, W, lens = _exact_case()
    reset_cfg()
    cfg.TEST.RPN_POST_NMS_TOP_N = Nb
    k = V.shape[0] // world
    Vl = V[rank * k:(rank + 1) * k].clone().requires_grad_(True)      # this rank's frames
    Wl = W.clone().requires_grad_(True)                                # replicated queries
    D_ind, D_sim, loss = dvsa_frame_sharded(_FakeDVSA(Na, Ne), Vl, Wl, lens, kernels=CpuKernels)
    loss.backward()
    # the partial dW of the ranks sums to the global dW (what GradAllReducer.allreduce(average=False) does to the
    # parameter gradients that follow from it)
    red = GradAllReducer([torch.nn.Parameter(torch.zeros_like(W))])
    red.flat.copy_(Wl.grad.reshape(-1))
    red.allreduce(average=False)
    torch.save({"D_ind": D_ind, "D_sim": D_sim, "loss": float(loss), "dV": Vl.grad, "dW": red.flat.view_as(W).clone()},
               out % rank)
    dist.destroy_process_group()


def test_exact_mode_exchange_world2(tmp_path):
    """Frame-sharded exact mode on 2 gloo ranks (CPU stand-in kernels): the all-gather of S_max / arg-max, the broadcast of
    the frame-0 rows, the rank-0-only clustering rows and the summed partial gradients reproduce the single-process
    DVSA on the whole batch."""
    from oracle import dvsa as O
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_exact_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    Na, Ns, Nb, Ne, V, W, lens = _exact_case()
    Vr, Wr = V.clone().requires_grad_(True), W.clone().requires_grad_(True)
    Di, Ds, L = O.dvsa_forward(Vr, Wr, lens, Na, Nb, Ne, 10.0, 4.13, 'train')
    L.backward()
    rs = [torch.load(out % r) for r in range(2)]
    k = V.shape[0] // 2
    for r, o in enumerate(rs):
        assert torch.equal(o["D_ind"], Di) and torch.allclose(o["D_sim"], Ds)
        assert abs(o["loss"] - float(L)) < 1e-5 * abs(float(L))
        assert torch.allclose(o["dV"], Vr.grad[r * k:(r + 1) * k], atol=1e-6)       # own frames only
        assert torch.allclose(o["dW"], Wr.grad, atol=1e-6)                           # summed partials
    assert rs[1]["dV"][:Nb].abs().sum() >= 0 and not torch.equal(rs[0]["dV"], rs[1]["dV"])


def _direct_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nafae_amd.parallel import GradAllReducer, broadcast_parameters, trainable_parameters
    torch.manual_seed(5 + rank)
    res = {}
    for mode in ("allreduce", "direct"):
        m = _Toy()
        broadcast_parameters(m, src=0)
        red = GradAllReducer(trainable_parameters(m), mode=mode)
        g = torch.Generator().manual_seed(40 + rank)
        x, gl = torch.randn(10, 12, generator=g), torch.randn(5, 6, generator=g)
        red.zero_grad()
        _loss(m, x, gl).backward()
        local = red.flat.clone()
        red.allreduce()
        res[mode] = (local, red.flat.clone())
    torch.save(res, out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_direct_reduce_scatter_allgather_equals_allreduce(tmp_path):
    """The one-shot exchange (all-to-all of the shards, fixed-order local sum, all-gather; flat buffer padded to a multiple of
    the world size) leaves the mean of the local gradients on every rank, like the all-reduce; world 3 so that padding is used."""
    world = 3
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_direct_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    rs = [torch.load(out % r) for r in range(world)]
    for mode in ("direct", "allreduce"):                    # (each mode ran on its own freshly initialised toy model)
        mean_local = sum(r[mode][0] for r in rs) / world
        for r in rs:
            assert torch.allclose(r[mode][1], mean_local, rtol=1e-6, atol=1e-7), mode
            assert torch.equal(r[mode][1], rs[0][mode][1]), mode                # replicas bit-identical
