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
