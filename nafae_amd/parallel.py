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
