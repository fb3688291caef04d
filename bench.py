#!/usr/bin/env python3
"""Headline benchmark: training steps of the NAFAE grounding hot path on synthetic data (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4|c5] [--no-cpu-baseline]

One "step" = one pass of the hot path over one segment batch, exactly the body of the reference's train loop
(model.py:684-775): frozen detector forward (VGG16 conv -> RPN/NMS -> ROI-Align -> fc6/fc7), VisEbd / WordEbd,
similarity + contextual-similarity + clustering loss, backward, gradient all-reduce (N > 1), clip, Adam.
Inputs (frames, GloVe rows) are resident in HBM before the timed region.  At N = 1 the workload is BASELINE
config C2 (64 frames 224x224, 128 proposals/frame, 16 query slots, fp32 parity); each additional GPU processes its
own 64 frames (weak scaling, no data-path collective; one RCCL all-reduce of 8.8 MB of gradients per step).

Arithmetic (`dtype`, --precision): the default `bf16x3` evaluates every detector contraction as three bf16 MFMAs on
split-bf16 operands with fp32 accumulation -- it meets the fp32 parity bar of BASELINE.json (1e-4; measured ~1e-5,
tests/test_gpu_bf16.py) and is what SURVEY.md section 7(iv) names as the alternative to fp32 MFMA.  `f32` (exact fp32
MFMA) and `bf16` (BASELINE config C3) are timed in the same invocation and reported under `other_precisions`.

Prints ONE JSON line on rank 0.  `value` = frames/s over all GPUs; pairs/s through sim+loss is reported next to
it.  `roofline` prices the dominant kernel (the fc6 fp32-MFMA GEMM) from HIP-event timings taken inside the timed
region on the launch stream; `cpu_baseline` times the CPU oracle on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (Na, Ns, Nb, Ne)            BASELINE.json configs
    "c2": (8, 8, 128, 16),
    "c4": (8, 8, 256, 32),
    "c5": (8, 8, 300, 64),
    "c1": (2, 2, 32, 8),
}
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense (the 5 PF headline includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0


def flops_per_frame(Nb):
    """Algorithmic FLOPs of the detector forward per frame."""
    return 30.693e9 + 0.939e9 + Nb * (205.5e6 + 33.55e6 + 4.19e6)


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch from the latest committed PMC pass (profiles/rNN_pmc_counters.json: separate
    --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 correction).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_counters.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return d[kernel_key]["traffic_bytes_corrected"]
    except Exception:
        return None


def cpu_baseline(Na, Ns, Nb, Ne, seconds_budget=25.0):
    """CPU oracle (oracle/: plain PyTorch fp32 + C NMS/ROI-Align) on a bounded sample of the same workload:
    `nf` frames through the detector + embeddings, then sim+loss at the full (R, Q) shape, all host cores."""
    import torch
    from nafae_amd import synthetic as syn
    from oracle import detector as OD
    from oracle import dvsa as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    nf = 4
    sd = syn.detector_state(seed=1234, heads=False)
    im, im_info = syn.frames(nf, 224, 224, seed=1234)
    ocfg = dict(FEAT_STRIDE=16, ANCHOR_SCALES=[4, 8, 16, 32], ANCHOR_RATIOS=[0.5, 1, 2], RPN_PRE_NMS_TOP_N=6000,
                RPN_POST_NMS_TOP_N=Nb, RPN_NMS_THRESH=0.7, POOLING_SIZE=7)
    t0 = time.time()
    rois, rs, pooled, fc7 = OD.detector_forward(im, im_info, sd, ocfg)
    t_det = time.time() - t0
    # sim + loss (+ backward) at the full shape
    V, W = syn.embeddings(Na * Ns * Nb, Na * Ne, 512, seed=1)
    lens = syn.entity_lengths(Na, Ne, seed=1234)
    V.requires_grad_(); W.requires_grad_()
    t0 = time.time()
    Di, Ds, L = O.dvsa_forward(V, W, lens, Na, Nb, Ne, 10.0, 4.13, 'train')
    L.backward()
    t_sim = time.time() - t0
    frames = Na * Ns
    per_frame = t_det / nf + t_sim / frames
    return {"value": round(1.0 / per_frame, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "oracle detector forward on %d of %d frames (%.1f s) + sim+loss fwd+bwd at R=%d,Q=%d (%.2f s), "
                      "torch CPU fp32, %d threads" % (nf, frames, t_det, Na * Ns * Nb, Na * Ne, t_sim, cores),
            "pairs_per_s": round(Na * Ns * Nb * Na * Ne / t_sim, 1)}


def sim_loss_only(Na, Ns, Nb, Ne, dev, iters=50):
    """The similarity + loss part alone (SURVEY.md section 8d, C5 note): synthetic V, W = tanh(N(0,1)) of the workload's shape,
    sim+max forward, loss tail forward+backward, similarity backward; HIP-event time over `iters` back-to-back passes."""
    import torch
    from nafae_amd import ops
    from nafae_amd import synthetic as syn
    F, Q, R, D = Na * Ns, Na * Ne, Na * Ns * Nb, 512
    V, W = syn.embeddings(R, Q, D, seed=1)
    V, W = V.to(dev), W.to(dev)
    lens = torch.tensor(syn.entity_lengths(Na, Ne, seed=1234), dtype=torch.int32, device=dev)
    ws = ops.loss_workspace(Na, Ns, Nb, Ne, D, V.device)

    def fwd():
        return ops.sim_max_fwd(V, W, lens, Na, Ns, Nb, Ne)

    def full():
        S_max, D_ind = fwd()
        loss, dS, _ = ops.loss_fwd_bwd(S_max, D_ind, V, lens, Na, Ns, Nb, Ne, 10.0, 4.13, True, workspace=ws)
        return ops.sim_bwd(dS, D_ind, V, W, lens, Na, Ns, Nb, Ne, True, ws)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    t_f, t_fb = timeit(fwd), timeit(full)
    by_f = 4.0 * D * (R + Q) + 12.0 * F * Q                       # forward algorithmic bytes
    by_fb = by_f + 4.0 * D * (R + Q) + 4.0 * D * F * Q            # + dense dV, dW and the arg-max row re-reads
    return {"R": R, "Q": Q, "pairs": R * Q,
            "fwd_ms": round(t_f, 4), "fwd_pairs_per_s": round(R * Q / (t_f * 1e-3), 1),
            "fwd_hbm_frac": round(by_f / (t_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "fwd_fp32_mfma_frac": round(2.0 * R * Q * D / (t_f * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "fwd_bwd_ms": round(t_fb, 4), "fwd_bwd_pairs_per_s": round(R * Q / (t_fb * 1e-3), 1),
            "fwd_bwd_hbm_frac": round(by_fb / (t_fb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run detector and tail of a step back to back on one stream (default: the frozen detector of step "
                         "k+1 overlaps the embedding/loss/backward/optimiser tail of step k on a second HIP stream)")
    ap.add_argument("--no-other-precisions", action="store_true",
                    help="skip the short extra runs in the other two arithmetic modes (N = 1 only)")
    ap.add_argument("--dp-mode", default="replica", choices=["replica", "exact"],
                    help="N > 1: 'replica' = per-GPU minibatch of whole segments, local loss, averaged gradients (default, "
                         "BASELINE.json's DP); 'exact' = ONE global batch of N x the segments, frames sharded over the GPUs, "
                         "S_max all-gathered, summed partial gradients (equals a 1-GPU step on the global batch)")
    ap.add_argument("--precision", default=os.environ.get("NAFAE_PRECISION", "bf16x3"), choices=["f32", "bf16x3", "bf16"],
                    help="arithmetic of the detector contractions: exact fp32 MFMA | split-bf16 (fp32-accurate to ~1e-5) | bf16")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from nafae_amd import ops
    from nafae_amd.config import cfg, cfg_from_file, reset_cfg
    from nafae_amd.model import default_args
    from nafae_amd.train import PipelinedTrainer, make_batch, setup_training, shard_frames, train_step, train_step_exact

    Na, Ns, Nb, Ne = WORKLOADS[a.workload]
    reset_cfg()
    cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
    cfg.TEST.RPN_POST_NMS_TOP_N = Nb
    exact = a.dp_mode == "exact"
    Na_model = Na * world if exact else Na          # exact mode: ONE batch of world*Na segments, every rank sees all queries
    args = default_args(batch_size=Na_model, sample_num=Ns, max_ent_len=Ne, Delta=10.0, vis_lam=4.13)
    model, opt, crit, reducer = setup_training(args, device=dev, seed=1234, distributed=distributed)
    model.fasterRCNN.precision = a.precision
    if exact:
        if args.dropout_rate:
            torch.manual_seed(1234)                 # word-side dropout masks must agree across ranks
        batch = shard_frames(make_batch(Na_model, Ns, Ne, seed=1234, device=dev), rank, world)
    else:
        batch = make_batch(Na, Ns, Ne, seed=1234 + rank, device=dev)

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    pipe = None if (a.no_pipeline or exact) else PipelinedTrainer(model, opt, crit, args, reducer)

    def run_steps(n):
        """exactly n detector forwards and n tails; everything is enqueued inside the caller's timed region"""
        if exact:
            for _ in range(n):
                loss, _, _, _ = train_step_exact(model, opt, crit, batch, args, reducer)
            return loss
        if pipe is None:
            for _ in range(n):
                loss, _, _, _ = train_step(model, opt, crit, batch, args, reducer)
            return loss
        pipe.submit(batch)
        for i in range(n):
            loss, _, _, _ = pipe.step(batch if i + 1 < n else None)
        return loss

    if a.warmup:
        run_steps(a.warmup)
    sync()
    ops.profile_reset(enable=True)
    t0 = time.perf_counter()
    loss = run_steps(a.steps)
    sync()
    dt = time.perf_counter() - t0
    prof = ops.profile_summary()
    ops.profile_reset(enable=False)
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        F = Na * Ns
        R, Q = F * Nb, Na * Ne
        frames_per_s = world * F * a.steps / dt
        out = {
            "metric": "frames/sec + region-query-pairs/sec through sim+loss",
            "value": round(frames_per_s, 2), "unit": "frames/s",
            # replica DP: every rank pairs its own R regions with its own Q queries; exact mode: one global R x Q problem
            "pairs_per_s": round(world * R * Q * (world if exact else 1) * a.steps / dt, 1),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "%s: %d frames 224x224 per GPU (Na=%d,Ns=%d), %d proposals/frame, %d query slots/segment, "
                                   "VGG16 random-init, full train step" % (a.workload.upper(), F, Na, Ns, Nb, Ne),
                       "frames_per_gpu": F, "proposals_per_frame": Nb, "queries_per_segment": Ne,
                       "parallelism": ("dp%d" % world) + ("-exact-global-batch" if exact else ""), "grad_allreduce_bytes": reducer.nbytes if distributed else 0,
                       "step_pipeline": "detector(k+1) overlaps tail(k) on a second stream" if pipe else "sequential"},
            "loss": round(float(loss), 5),
        }
        # dominant kernel: fc6 = [R,25088] x [4096,25088]^T on fp32 MFMA
        fc6 = prof.get("fc6")
        if fc6:
            fl = 2.0 * R * 25088 * 4096
            # MFMA flops actually issued per algorithmic flop, and the dense peak of the pipe they run on
            nprod, peak, kname = {"f32": (1, FP32_MFMA_PEAK_TFLOPS, "gemm_nt_kernel<128,128,2,2> (fc6, fp32 MFMA)"),
                                  "bf16x3": (3, BF16_MFMA_PEAK_TFLOPS, "bf16_dma_kernel<256,256,2,4,split,gemm,2> (fc6, 3 bf16 MFMAs per product)"),
                                  "bf16": (1, BF16_MFMA_PEAK_TFLOPS, "bf16_dma_kernel<256,256,2,4,plain,gemm,3> (fc6, bf16 MFMA)")}[a.precision]
            ach = nprod * fl / (fc6["avg_ms"] * 1e-3) / 1e12
            out["roofline"] = {"kernel": kname, "bound": "mfma", "achieved": round(ach, 2),
                               "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "algorithmic_tflops": round(fl / (fc6["avg_ms"] * 1e-3) / 1e12, 2),
                               "traffic": (pmc_traffic({"f32": "gemm_nt_fc6", "bf16x3": "gemm_bf16x3_fc6_256x256_il",
                                                        "bf16": "gemm_bf16_plain_fc6_256x256"}.get(a.precision, ""))
                                           if a.workload == "c2" else None),
                               "algorithmic": (2.0 if a.precision == "bf16" else 4.0) * (R * 25088 + 4096 * 25088 + R * 4096),
                               "avg_ms": round(fc6["avg_ms"], 4), "launches": fc6["n"]}
        sim = prof.get("sim_max")
        if sim:
            by = 4.0 * 512 * (R + Q) + 12.0 * F * Q
            ach = by / (sim["avg_ms"] * 1e-3) / 1e9
            out["roofline_sim"] = {"kernel": "sim_max_kernel", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                   "traffic": pmc_traffic("sim_max") if a.workload == "c2" else None,
                                   "avg_ms": round(sim["avg_ms"], 4),
                                   "mfma_frac": round(2.0 * R * Q * 512 / (sim["avg_ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
        out["stage_ms"] = {k: round(v["avg_ms"], 4) for k, v in sorted(prof.items())}
        det_ms = sum(v["avg_ms"] for k, v in prof.items() if k in ("base", "rpn", "roi_align", "fc6", "fc7"))
        if det_ms > 0:
            out["detector_algorithmic_tflops"] = round(F * flops_per_frame(Nb) / (det_ms * 1e-3) / 1e12, 2)
            out["detector_fp32_mfma_frac"] = round(F * flops_per_frame(Nb) / (det_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
        if world == 1:
            out["sim_loss_only"] = sim_loss_only(Na, Ns, Nb, Ne, dev)
        if world == 1 and not a.no_other_precisions:
            # the same step in the other arithmetic modes, a few steps each (reported, never the headline)
            other = {}
            for prec in ("f32", "bf16x3", "bf16"):
                if prec == a.precision:
                    continue
                model.fasterRCNN.precision = prec
                for _ in range(2):
                    train_step(model, opt, crit, batch, args, reducer)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n_o = max(3, min(a.steps, 5))
                for _ in range(n_o):
                    train_step(model, opt, crit, batch, args, reducer)
                torch.cuda.synchronize()
                d_o = (time.perf_counter() - t1) / n_o
                other[prec] = {"frames_per_s": round(F / d_o, 2), "ms_per_step": round(1e3 * d_o, 3), "steps": n_o}
            model.fasterRCNN.precision = a.precision
            out["other_precisions"] = other
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(Na, Ns, Nb, Ne)
            except Exception as e:      # the baseline is reporting, never a reason to lose the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
