"""Grounding evaluation: host-side mirror of the reference's detection recording and box / query accuracy
(SURVEY.md section 8f.1 -- the callers right after the hot path).

  record_det       <- model.py:477-487
  phrase_accuracy  <- lib/datasets/youcook_eval.py:135-237   ("macro / micro query accuracy")
  box_accuracy     <- lib/datasets/youcook_eval.py:241-336   ("macro / micro box accuracy")
  evaluate_box     <- lib/datasets/youcook_eval.py:408-413

The reference does this in Python/numpy on the host; so does this module (it is O(#detections x #gt boxes of one
frame), not a device hot spot).  Semantics are kept bug-for-bug, because the published accuracy numbers depend on
them: detections are grouped per frame after a stable sort by frame id, and inside a frame they are ordered by a
confidence array that the reference permutes by that sort order TWICE (youcook_eval.py:155,160 and :261,266); the IoU
uses +1 widths and `ov >= thr`; a query class is counted once per frame (phrase) / every gt box is counted (box).
Pinned against the imported reference by tests/golden/eval.npz.
"""
import numpy as np


def record_det(img_inds, obj_labels, obj_bboxes, obj_confs, Nb, vid_entities, D, D_sim, img_ids, infer_boxes):
    """Append (frame id, query label, grounded box, similarity) for every real entity of every sampled frame.
    D [Na,Ns,Ne]: global box indices from `postprocess`; infer_boxes [F*Nb,4]; img_ids [F]."""
    Na, Ns, Ne = D.shape
    for a, entities in enumerate(vid_entities):
        for s in range(Ns):
            for e, entity in enumerate(entities):
                box_id = int(D[a][s][e])
                img_inds.append(img_ids[box_id // Nb])
                obj_labels.append(entity)
                obj_bboxes.append(infer_boxes[box_id])
                obj_confs.append(D_sim[a][s][e])


def _group_by_frame(dets):
    img_ids = np.array(dets[0])
    labels = np.array(dets[1])
    boxes = np.array(dets[2])
    confs = np.array(dets[3])
    order = np.argsort(img_ids)
    img_ids, labels, boxes = img_ids[order], labels[order], boxes[order]
    confs = confs[order][order]                        # the reference applies `order` twice to the confidences
    num_imgs = int(np.max(img_ids)) + 1
    cells = [None] * num_imgs
    bounds = np.flatnonzero(np.diff(img_ids)) + 1
    for lo, hi in zip(np.r_[0, bounds], np.r_[bounds, len(img_ids)]):
        by_conf = np.argsort(-confs[lo:hi])
        cells[int(img_ids[lo])] = (labels[lo:hi][by_conf], boxes[lo:hi][by_conf])
    return cells, num_imgs


def _iou_ge(obj, gt, thr):
    ix1, iy1 = max(obj[0], gt[0]), max(obj[1], gt[1])
    ix2, iy2 = min(obj[2], gt[2]), min(obj[3], gt[3])
    iw, ih = ix2 - ix1 + 1, iy2 - iy1 + 1
    if iw > 0 and ih > 0:
        ua = (obj[2] - obj[0] + 1.) * (obj[3] - obj[1] + 1.) + (gt[2] - gt[0] + 1.) * (gt[3] - gt[1] + 1.) - iw * ih
        return iw * ih / ua >= thr
    return False


def _summary(match, count):
    cls_acc = match / (count + 1e-6)
    return float(np.mean(cls_acc)), float(np.sum(match) / np.sum(count))


def phrase_accuracy(recs, dets, class_list, verbose=False, both=False):
    """Per (frame, query class): counted once if the class has a gt box in the frame; matched if any of its gt boxes
    overlaps the grounded box with IoU >= that box's threshold.  Returns the macro (class-mean) accuracy.

    Bug-for-bug with youcook_eval.py:185-227: the reference increments `class_match_count[class_ind]` with the `class_ind`
    left over from the label most recently INSERTED into its per-frame dict, not the index of the label that matched -- so
    when a frame carries a repeated entity label interleaved with other labels (A, B, A) a later match of A is booked on B.
    The total (micro accuracy) is unaffected, the printed macro accuracy is; tests/golden/eval_dup.npz pins that case."""
    cells, num_imgs = _group_by_frame(dets)
    match = np.zeros(len(class_list), dtype=int)
    count = np.zeros(len(class_list), dtype=int)
    class_ind = 0
    for img_id in range(num_imgs):
        if cells[img_id] is None:
            continue
        rec = recs[img_id]
        state = {}                                       # label -> 0 (seen, unmatched) / 1 (matched)
        for obj_label, obj_box in zip(*cells[img_id]):
            for gt_label, gt_box, thr in zip(rec['label'], rec['bbox'], rec['thr']):
                if obj_label != gt_label:
                    continue
                if obj_label not in state:
                    state[obj_label] = 0
                    class_ind = class_list.index(gt_label)
                    count[class_ind] += 1
                elif state[obj_label] == 1:
                    continue
                if _iou_ge(obj_box, gt_box, thr):
                    match[class_ind] += 1                # (stale index when the label was inserted earlier: see docstring)
                    state[obj_label] = 1
    macro, micro = _summary(match, count)
    if verbose:
        print('macro query accuracy: {:0.2%}'.format(macro))
        print('micro query accuracy: {:0.2%}'.format(micro))
    return (macro, micro) if both else macro        # (the reference returns the macro figure; `both` adds the printed micro one)


def box_accuracy(recs, dets, class_list, verbose=False, both=False):
    """Per gt box: counted always; matched if some detection of the same class in that frame has IoU >= its threshold."""
    cells, num_imgs = _group_by_frame(dets)
    match = np.zeros(len(class_list), dtype=int)
    count = np.zeros(len(class_list), dtype=int)
    for img_id in range(num_imgs):
        rec = recs[img_id]
        for gt_label, gt_box, thr in zip(rec['label'], rec['bbox'], rec['thr']):
            ci = class_list.index(gt_label)
            count[ci] += 1
            if cells[img_id] is None:
                continue
            for obj_label, obj_box in zip(*cells[img_id]):
                if obj_label == gt_label and _iou_ge(obj_box, gt_box, thr):
                    match[ci] += 1
                    break
    macro, micro = _summary(match, count)
    if verbose:
        print('macro box accuracy: {:0.2%}'.format(macro))
        print('micro box accuracy: {:0.2%}'.format(micro))
    return (macro, micro) if both else macro


def evaluate_box(recs, dets, class_list, verbose=False):
    phrase_accuracy(recs, dets, class_list, verbose)
    return box_accuracy(recs, dets, class_list, verbose)
