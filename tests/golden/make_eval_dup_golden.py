"""tests/golden/eval_dup.npz: the reference's phrase_accuracy / box_accuracy (lib/datasets/youcook_eval.py:135-336) on frames
that carry a REPEATED entity label interleaved with other labels -- the case in which the reference books a match on the class
index left over from the last label inserted.      python tests/golden/make_eval_dup_golden.py"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness as H  # noqa: E402


def main():
    H.load_reference()
    from datasets.youcook_eval import box_accuracy, phrase_accuracy
    classes = ['bowl', 'egg', 'pan', 'oil']
    box = lambda x: np.array([x, x, x + 50.0, x + 50.0])
    # frame 0: detections (bowl: miss), (egg: miss), (bowl again: hit) -> the hit is booked on 'egg' by the reference
    # frame 1: (pan: hit), (oil: miss), (pan again) ; frame 2: (egg: miss), (oil: hit first try), (egg: hit)
    recs = [{'label': ['bowl', 'egg'], 'bbox': [box(10), box(200)], 'thr': [0.5, 0.5], 'img_ids': [0, 0]},
            {'label': ['pan', 'oil'], 'bbox': [box(30), box(300)], 'thr': [0.5, 0.5], 'img_ids': [1, 1]},
            {'label': ['egg', 'oil'], 'bbox': [box(60), box(150)], 'thr': [0.5, 0.5], 'img_ids': [2, 2]}]
    det_img = [0, 0, 0, 1, 1, 1, 2, 2, 2]
    det_lab = ['bowl', 'egg', 'bowl', 'pan', 'oil', 'pan', 'egg', 'oil', 'egg']
    det_box = [box(400), box(500), box(12), box(31), box(0), box(600), box(400), box(151), box(61)]
    det_conf = [0.9, 0.8, 0.7, 0.9, 0.8, 0.7, 0.9, 0.8, 0.7]
    dets = [det_img, det_lab, det_box, det_conf]
    with contextlib.redirect_stdout(io.StringIO()):
        pa = phrase_accuracy(recs, dets, classes)
        ba = box_accuracy(recs, dets, classes)
    np.savez_compressed(os.path.join(HERE, "eval_dup.npz"), classes=np.array(classes),
                        rec_lab=np.array(['|'.join(r['label']) for r in recs]), rec_box=np.array([r['bbox'] for r in recs]),
                        det_img=np.array(det_img), det_lab=np.array(det_lab), det_box=np.array(det_box), det_conf=np.array(det_conf),
                        phrase_acc=np.float64(pa), box_acc=np.float64(ba))
    print("phrase", pa, "box", ba)


if __name__ == "__main__":
    main()
