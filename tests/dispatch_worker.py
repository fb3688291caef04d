"""Child process of tests/test_gpu_dispatch.py: runs every BASELINE layer / GEMM / similarity shape against the EXPERIMENTS build
of the library (NAFAE_LIB=.../libnafae_hip_exp.so: the only build that leaves dispatch tags, csrc/hip_util.h NAFAE_TAG) and
writes {"<op> <shape signature>": "<kernel tag>"} as JSON.

    python tests/dispatch_worker.py <out.json>

The wrappers below sit on nafae_amd.ops (what detector.py / model.py call), so the table records the dispatch of the REAL call
sequence -- the detector forward in the three arithmetic modes at C2, the fc / embedding / similarity / loss / backward calls at
C2, C4 and C5 with entity lengths from the histogram and with every slot live."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    from nafae_amd import _lib, ops
    assert _lib.LIB_PATH.endswith("_exp.so"), _lib.LIB_PATH
    L = _lib.lib()
    L.nafae_last_kernel_id.restype = ctypes.c_char_p
    L.nafae_last_kernel_id.argtypes = []
    table = {}
    ctx = {"label": ""}

    def shape_of(a):
        if isinstance(a, torch.Tensor):
            return "x".join(str(d) for d in a.shape)
        if isinstance(a, ops.Planes):
            return "P" + "x".join(str(d) for d in a.shape) + ("s" if a.lo is not None else "p")
        return None

    def wrap(name, nargs):
        fn = getattr(ops, name)

        def wrapped(*a, **kw):
            r = fn(*a, **kw)
            tag = L.nafae_last_kernel_id().decode()
            sig = " ".join(s for s in (shape_of(x) for x in a[:nargs]) if s)
            if name == "conv3x3_wino":                 # (x, U, bias, Cout): the transformed weights are a flat tensor
                sig += " Cout=%d" % a[3]
            extra = ""
            if name.startswith("conv3x3") and kw.get("pool"):
                extra = " pool"
            key = "%s%s [%s]%s" % (ctx["label"], name, sig, extra)
            if key in table and table[key] != tag:
                raise SystemExit("dispatch is not a function of the shapes: %s -> %s and %s" % (key, table[key], tag))
            table[key] = tag
            return r
        setattr(ops, name, wrapped)

    for name, nargs in (("conv3x3_relu", 2), ("conv3x3_wino", 1), ("conv3x3_bf16", 2), ("gemm_nt", 2), ("gemm_nt_bf16", 2), ("sim_max_fwd_frames", 2),
                        ("loss_fwd_bwd", 1), ("sim_bwd", 1), ("sim_bwd_frames", 1)):
        wrap(name, nargs)

    from nafae_amd import synthetic as syn
    from nafae_amd.config import cfg, cfg_from_file, reset_cfg
    from nafae_amd.model import default_args
    from nafae_amd.train import make_batch, setup_training, train_step
    for wl, (Na, Ns, Nb, Ne) in (("c2", (8, 8, 128, 16)), ("c4", (8, 8, 256, 32)), ("c5", (8, 8, 300, 64))):
        reset_cfg()
        cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
        cfg.TEST.RPN_POST_NMS_TOP_N = Nb
        args = default_args(batch_size=Na, sample_num=Ns, max_ent_len=Ne, Delta=10.0, vis_lam=4.13)
        model, opt, crit, reducer = setup_training(args, device="cuda", seed=1234)
        for lens_kind in ("hist", "all-live"):
            lens = syn.entity_lengths(Na, Ne, seed=1234) if lens_kind == "hist" else [Ne] * Na
            batch = make_batch(Na, Ns, Ne, seed=1234, device="cuda", lens=lens)
            # "f32" = the default exact-fp32 route (Winograd convs); "f32-direct" keeps the implicit-GEMM convs pinned as well
            for prec in (("f32", "f32-direct", "bf16x3", "bf16") if lens_kind == "hist" else ("bf16x3",)):
                model.fasterRCNN.precision = prec.split("-")[0]
                model.fasterRCNN.conv_algo = "direct" if prec.endswith("-direct") else "winograd"
                ctx["label"] = "%s/%s/%s " % (wl, prec, lens_kind)
                train_step(model, opt, crit, batch, args, reducer)
        torch.cuda.synchronize()
        del model, opt, reducer
        torch.cuda.empty_cache()
    with open(out_path, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
