"""Worker of tests/test_gpu_rccl_world1.py: ONE rank with backend nccl (= RCCL on ROCm) on the box's one GPU, so that every
collective of the N > 1 path executes at least once before the first 8-GPU run.
usage: RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/rccl_world1_worker.py OUT.pt
A fresh process: nothing here replaces a process that has touched the GPU."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    from nafae_amd import parallel as P
    P.FORCE_COLLECTIVES = True                    # a world-size-1 group would skip every collective otherwise
    from nafae_amd.config import cfg_from_file, cfg_from_list, reset_cfg
    from nafae_amd.model import default_args
    from nafae_amd.train import make_batch, setup_training, shard_frames, train_step, train_step_exact
    reset_cfg()
    cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
    cfg_from_list(["TEST.RPN_POST_NMS_TOP_N", "32"])
    Na, Ns, Ne = 2, 2, 8
    args = default_args(batch_size=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=10.0, vis_lam=4.13)
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}

    # -- the raw collectives of nafae_amd.parallel on device tensors (no host staging: that is the gloo test path)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(4096 + 3, device="cuda", generator=g)
    for mode in ("allreduce", "direct"):
        p = torch.nn.Parameter(torch.zeros_like(x))
        red = P.GradAllReducer([p], mode=mode)
        red.flat.copy_(x)
        red.allreduce(average=True)
        torch.cuda.synchronize()
        res["reduce_%s_equal" % mode] = bool(torch.equal(red.flat, x))      # one rank: the sum is the operand, bit for bit
    rows = torch.randn(7, 5, device="cuda", generator=g)
    res["all_gather_rows_equal"] = bool(torch.equal(P.all_gather_rows(rows), rows))
    ind = torch.arange(35, device="cuda", dtype=torch.int64).view(7, 5)
    res["all_gather_rows_i64_equal"] = bool(torch.equal(P.all_gather_rows(ind), ind))
    b = rows.clone()
    P.broadcast_rows(b, 0)
    res["broadcast_rows_equal"] = bool(torch.equal(b, rows))
    dist.barrier()

    # -- replicated DP step (broadcast_parameters inside setup_training, all-reduce of the flat gradient buffer) in both
    #    exchange modes, against the same step without a process group's collectives
    losses = {}
    for mode in ("allreduce", "direct"):
        model, opt, crit, red = setup_training(args, seed=21, distributed=True, grad_exchange=mode)
        batch = make_batch(Na, Ns, Ne, seed=100, lens=[3, 5])
        loss, D, D_sim, rois = train_step(model, opt, crit, batch, args, red)
        torch.cuda.synchronize()
        losses[mode] = (float(loss), opt.flat_params.clone().cpu())
    P.FORCE_COLLECTIVES = False
    model, opt, crit, red = setup_training(args, seed=21, distributed=False)
    batch = make_batch(Na, Ns, Ne, seed=100, lens=[3, 5])
    loss, D, D_sim, rois = train_step(model, opt, crit, batch, args, red)
    torch.cuda.synchronize()
    ref = (float(loss), opt.flat_params.clone().cpu())
    P.FORCE_COLLECTIVES = True
    for mode in ("allreduce", "direct"):
        res["step_%s_loss_equal" % mode] = losses[mode][0] == ref[0]
        res["step_%s_params_equal" % mode] = bool(torch.equal(losses[mode][1], ref[1]))

    # -- frame-sharded exact mode: all-gather of S_max / D_ind, broadcast of the clustering rows, summed partial gradients
    model, opt, crit, red = setup_training(args, seed=21, distributed=True)
    gb = make_batch(Na, Ns, Ne, seed=100, lens=[3, 5])
    loss, D, D_sim, rois = train_step_exact(model, opt, crit, shard_frames(gb, 0, 1), args, red)
    torch.cuda.synchronize()
    res["exact_loss_equal"] = float(loss) == ref[0]
    res["exact_params_maxdiff"] = float((opt.flat_params.cpu() - ref[1]).abs().max())
    res["exact_loss"] = float(loss)
    res["ref_loss"] = ref[0]
    dist.barrier()
    torch.save(res, out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
