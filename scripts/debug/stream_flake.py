"""Stress loop around tests/test_gpu_widened.py::test_frame_streamer_feeds_the_pipeline_identically (one rare failure in a full-suite
run, round 5): the three arms many times in one process, printing every mismatch (arm, step, values)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from nafae_amd.config import cfg, cfg_from_file
from nafae_amd.model import default_args
from nafae_amd.train import Batch, FrameStreamer, PipelinedTrainer, make_batch, setup_training, train_step
cfg_from_file(os.path.join(ROOT, 'cfgs', 'vgg16.yml'))
Na, Ns, Ne, Nb = 2, 3, 4, 16
cfg.TEST.RPN_POST_NMS_TOP_N = Nb
args = default_args(batch_size=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=10.0, vis_lam=4.13)
tmpl = make_batch(Na, Ns, Ne, H=96, W=96, seed=21, lens=[2, 3])
rs = np.random.RandomState(1)
host = [torch.from_numpy(rs.randint(0, 255, (Na * Ns, 96, 96, 3)).astype(np.uint8)).pin_memory() for _ in range(3)]
n = 5
bad = 0
junk = []
# timing perturbation (argv[2] = 1): a side stream kept busy with matmuls of random size, and random host-side pauses between the
# feeder and the trainer -- so that the copy stream, the detector stream and the tail interleave differently every step
import random, time
PERTURB = len(sys.argv) > 2 and sys.argv[2] == "1"
side = torch.cuda.Stream()
noise_a = torch.randn(2048, 2048, device='cuda')
rnd = random.Random(7)
def perturb():
    if not PERTURB:
        return
    with torch.cuda.stream(side):
        for _ in range(rnd.randint(0, 3)):
            k = rnd.choice([256, 512, 1024, 2048])
            (noise_a[:k, :k] @ noise_a[:k, :k]).sum()
    if rnd.random() < 0.3:
        time.sleep(rnd.random() * 0.003)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    if it % 3 == 1:      # perturb the allocator / the GPU's state between iterations like other tests would
        junk = [torch.randn(1 << (18 + (it % 5)), device='cuda') for _ in range(4)]
        (junk[0][:1 << 18] @ junk[1][:1 << 18]).item()
    elif it % 3 == 2:
        junk = []
        torch.cuda.empty_cache()
    model, opt, crit, red = setup_training(args, seed=5)
    ref = []
    for k in range(n):
        b = Batch(host[k % 3].cuda(), tmpl.im_info, tmpl.glove_feats, tmpl.entities_length)
        ref.append(float(train_step(model, opt, crit, b, args, red)[0]))
    want = torch.cat([p.detach().reshape(-1) for p in red.params]).clone()
    for pipelined in (False, True):
        model2, opt2, crit2, red2 = setup_training(args, seed=5)
        feeder = FrameStreamer(host, tmpl, "cuda")
        got = []
        if pipelined:
            pipe = PipelinedTrainer(model2, opt2, crit2, args, red2)
            pipe.submit(feeder.next())
            for i in range(n):
                perturb()
                nb = feeder.next() if i + 1 < n else None
                perturb()
                got.append(float(pipe.step(nb)[0]))
        else:
            for i in range(n):
                perturb()
                nb = feeder.next()
                perturb()
                got.append(float(train_step(model2, opt2, crit2, nb, args, red2)[0]))
        torch.cuda.synchronize()
        peq = torch.equal(torch.cat([p.detach().reshape(-1) for p in red2.params]), want)
        if got != ref or not peq:
            bad += 1
            print("iteration %d pipelined=%s: losses %s vs %s | params equal %s" % (it, pipelined, got, ref, peq), flush=True)
print("done: %d mismatching arms" % bad)
