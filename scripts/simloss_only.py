"""Similarity forward + loss tail + clustering + similarity backward alone, for rocprofv3 --kernel-trace runs:
    python scripts/simloss_only.py [c2|c4|c5] [hist|dense] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nafae_amd import ops, synthetic as syn
W = {"c2": (8, 8, 128, 16), "c4": (8, 8, 256, 32), "c5": (8, 8, 300, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
kind = sys.argv[2] if len(sys.argv) > 2 else "hist"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
Na, Ns, Nb, Ne = W[name]
lens = syn.entity_lengths(Na, Ne, seed=1234) if kind == "hist" else [Ne] * Na
V, Wt = syn.embeddings(Na * Ns * Nb, Na * Ne, 512, seed=1)
V, Wt = V.cuda(), Wt.cuda()
lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
# the operand planes of the many-live-column kernel, as the embedding modules hand them over (written by their tanh epilogue, attached
# to the tensors): produced once, outside the measured loop.  NAFAE_SIM_NO_PLANES=1: fp32 operands only (the entry point's pre-pass).
if not os.environ.get("NAFAE_SIM_NO_PLANES"):
    ops.attach_sim_planes(V, ops.sim_planes(V))
    ops.attach_sim_planes(Wt, ops.sim_planes(Wt))
ws = ops.loss_workspace(Na, Ns, Nb, Ne, 512, V.device)
for _ in range(iters):
    S, D = ops.sim_max_fwd(V, Wt, lt, Na, Ns, Nb, Ne, lens=lens)
    loss, dS, _ = ops.loss_fwd_bwd(S, D, V, lt, Na, Ns, Nb, Ne, 10.0, 4.13, True, workspace=ws, lens=lens)
    dV, dW = ops.sim_bwd(dS, D, V, Wt, lt, Na, Ns, Nb, Ne, True, ws)
torch.cuda.synchronize()
print(name, kind, "live", sum(lens), "loss", float(loss[0]), "dV", float(dV.abs().sum()), "dW", float(dW.abs().sum()))
