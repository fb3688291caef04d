"""Generates tests/golden/loader.npz by calling the REFERENCE's own MPrpDataSet methods (lib/datasets/youcook2.py) on a
synthetic in-memory "dataset": frame-path lists in the genframes.py naming, per-segment entity lists, random frames.

    python tests/golden/make_loader_golden.py

The methods are called unbound on a stand-in `self` carrying exactly the attributes they read (args, phase, entity_type,
vid_ids, img_id_dict ...), so no dataset files are needed; cv2 is the harness stub (nothing here decodes a JPEG: frames are
handed to __getitem__'s arithmetic as arrays).  Only input / output arrays are stored.
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import ref_harness as H  # noqa: E402


def main():
    m = H.load_reference()
    from datasets.youcook2 import MPrpDataSet
    from model.utils.blob import im_list_to_blob
    rs = np.random.RandomState(11)
    vid_ids = ['vidA', 'vidB', 'vidC']
    seg_frames = {'vidA': [7, 12, 3], 'vidB': [20], 'vidC': [5, 5, 9, 16]}
    entities = {'vidA': [['bowl', 'egg'], [], ['pan']], 'vidB': [['oil', 'salt', 'egg']],
                'vidC': [['water'], ['bowl', 'pan'], [], ['egg', 'oil', 'salt', 'pan']]}
    paths = {v: ['data/x/%s/%04d%06d.jpg' % (v, s, f) for s, n in enumerate(seg_frames[v]) for f in range(n)] for v in vid_ids}
    out = dict(vid_ids=np.array(vid_ids))
    for phase, kw in (('train', dict(fix_seg_len=True, sample_num=5, sample_rate=1, fix_seg_len_val=False, sample_num_val=0,
                                     sample_rate_val=16)),
                      ('val', dict(fix_seg_len=False, sample_num=5, sample_rate=1, fix_seg_len_val=False, sample_num_val=0,
                                   sample_rate_val=4))):
        args = types.SimpleNamespace(act_trunc=20, img_h=24, img_w=32, **kw)
        self = types.SimpleNamespace(args=args, phase=phase, entity_type=['category'], vid_ids=vid_ids, img_id_dict={})
        for name in ('parse_img_path', 'div_imglst_by_name', 'get_frm_inds', 'img_id_mapping', 'combine_batches'):
            setattr(self, name, types.MethodType(getattr(MPrpDataSet, name), self))
        # the bookkeeping loop of get_segment_num (youcook2.py:66-121), driven with the reference's own helpers
        img_id = -1
        actions_length = {}
        for v in vid_ids:
            shuffled = list(rs.permutation(paths[v]))            # glob order is arbitrary; div_imglst_by_name sorts
            cells = self.div_imglst_by_name(shuffled)
            holder = []
            for cell in cells:
                d = {}
                for f_ind in self.get_frm_inds(cell):
                    img_id += 1
                    d[self.parse_img_path(cell[f_ind])[3]] = img_id
                holder.append(d)
            self.img_id_dict[v] = holder
            actions_length[v] = [len(a) for a in entities[v]]
            out['cells_%s_%s' % (phase, v)] = np.array(['|'.join(c) for c in cells])
        # __getitem__'s image arithmetic (youcook2.py:208-227) on synthetic decoded frames, then the collate function
        datas, raws = [], []
        for (v, seg) in (('vidC', 3), ('vidA', 0), ('vidB', 0), ('vidA', 1)):
            cell = self.div_imglst_by_name(list(paths[v]))[seg]
            f_inds = self.get_frm_inds(cell)
            img_list = [cell[i] for i in f_inds]
            frames = rs.randint(0, 256, (len(img_list), 24, 32, 3)).astype(np.uint8)
            imgs = []
            for fr in frames:
                img = fr.astype(np.float32, copy=True)
                img -= 127.5
                imgs.append(img)
            blob = im_list_to_blob(imgs)
            datas.append((blob, entities[v][seg], img_list, False, actions_length[v], seg))
            raws.append(frames)
            out['finds_%s_%s_%d' % (phase, v, seg)] = np.asarray(f_inds)
        blobs, ents, ent_len, frm_length, rl, seg_nums, img_paths, img_ids = self.combine_batches(datas)
        out.update({'raw_' + phase: np.concatenate(raws, 0), 'blobs_' + phase: blobs, 'entities_' + phase: np.array(ents),
                    'entities_length_' + phase: np.array(ent_len), 'frm_length_' + phase: np.array(frm_length),
                    'rl_seg_inds_' + phase: np.array(rl), 'seg_nums_' + phase: np.array(seg_nums),
                    'img_paths_' + phase: np.array(img_paths), 'img_ids_' + phase: np.array(img_ids)})
    out['paths'] = np.array(['|'.join(paths[v]) for v in vid_ids])
    out['seg_frames'] = np.array([','.join(map(str, seg_frames[v])) for v in vid_ids])
    out['entities'] = np.array([';'.join(','.join(a) for a in entities[v]) for v in vid_ids])
    # im_list_to_blob on ragged shapes
    ims = [rs.rand(5, 7, 3).astype(np.float32), rs.rand(6, 4, 3).astype(np.float32)]
    out.update(rag0=ims[0], rag1=ims[1], rag_blob=im_list_to_blob(ims))
    path = os.path.join(HERE, "loader.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
