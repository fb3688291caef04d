"""Data parallelism over the GPUs of one node: one process per GPU, replicated model, per-rank minibatch of whole
segments, ONE all-reduce (RCCL over xGMI; gloo on CPU in tests) per step on a flat fp32 buffer holding the gradients
of exactly the parameters that ever receive one -- vis_ebd.fc1, word_ebd.fc1, word_ebd.bn = 2,201,600 floats =
8.8 MB at the reference sizes.  The frozen detector has no gradients, so nothing else moves.

The reference has no distributed path at all (--mGPUs is parsed and never read); this module
adds one.  ``p.grad`` of every reduced parameter is a VIEW into the flat buffer, so backward accumulates straight
into it and the collective needs no pack/unpack copies.
"""
import torch
import torch.distributed as dist


def trainable_parameters(model):
    """The parameters the reference's optimiser can actually move (model.py:1077-1082 builds Adam over
    DVSA + word_ebd + vis_ebd, but DVSA's parameters never get a gradient)."""
    ps = []
    for m in (model.vis_ebd, model.word_ebd):
        ps += [p for p in m.parameters() if p.requires_grad]
    return ps


class GradAllReducer:
    def __init__(self, params, group=None):
        self.params = list(params)
        self.group = group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

    def zero_grad(self):
        self.flat.zero_()

    def allreduce(self):
        """Average gradients over ranks (sum then divide, identical on every rank)."""
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world)
        return self.flat

    @property
    def nbytes(self):
        return self.flat.numel() * 4


def broadcast_parameters(model, src=0, group=None):
    """Make every rank start from rank `src`'s weights and BatchNorm statistics."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)


class FusedClipAdam:
    """The reference's optimiser step -- clip_grad_norm_(all params, clip) then Adam(lr, weight_decay) over
    DVSA + word_ebd + vis_ebd (model.py:773-774, :1077-1082) -- as ONE pair of HIP launches over flat buffers.
    Only parameters that ever receive a gradient take part (DVSA's own parameters have grad None in the reference and
    are skipped by both clip_grad_norm_ and Adam).  Parameters become views of a flat buffer; gradients are the
    GradAllReducer's flat buffer, so DP needs no extra copies."""

    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=100.0):
        from . import ops
        self._ops = ops
        self.reducer = reducer
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        n = reducer.flat.numel()
        dev = reducer.flat.device
        self.flat_params = torch.empty(n, device=dev, dtype=torch.float32)
        o = 0
        with torch.no_grad():
            for p in reducer.params:
                k = p.numel()
                self.flat_params[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_params[o:o + k].view_as(p)
                o += k
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self.workspace = torch.zeros(256, device=dev, dtype=torch.float32)
        self.total_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.step_count = 0
        self.param_groups = [{"lr": lr}]          # the reference decays lr by editing param_groups (model.py:1084-1088)

    def zero_grad(self):
        self.reducer.zero_grad()

    def step(self):
        self.step_count += 1
        self._ops.adam_step(self.flat_params, self.reducer.flat, self.exp_avg, self.exp_avg_sq, self.param_groups[0]["lr"],
                            self.betas[0], self.betas[1], self.eps, self.weight_decay, self.max_norm, self.step_count,
                            self.workspace, self.total_norm)
