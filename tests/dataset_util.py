"""A small dataset in the reference's on-disk formats (tests of the data layer and of the evaluation drivers): per-image
fc `.npy` and att `.npz['feat']` files, labels / label_start_ix / label_end_ix arrays, a json with ix_to_word and splits."""
import argparse
import json
import os

import numpy as np


def make_dataset(tmp, n=11, D=12, seq_length=16, ragged=True, seed=0, vocab=29, ncap_range=(1, 4), val_from=None):
    rs = np.random.RandomState(seed)
    fc_dir, att_dir = os.path.join(tmp, 'fc'), os.path.join(tmp, 'att')
    os.makedirs(fc_dir), os.makedirs(att_dir)
    images, labels, start, end = [], [], [], []
    feats = {}
    for i in range(n):
        iid = 1000 + 7 * i
        K = int(rs.randint(3, 8)) if ragged else 5
        fc = rs.rand(D).astype('float32')
        att = rs.rand(K, D).astype('float32')
        np.save(os.path.join(fc_dir, f'{iid}.npy'), fc)
        np.savez(os.path.join(att_dir, f'{iid}.npz'), feat=att.reshape(1, K, D))      # prepro_feats writes [h, w, C]
        feats[iid] = (fc, att)
        ncap = int(rs.randint(*ncap_range))
        start.append(len(labels) + 1)                                                # 1-based, as prepro_labels.py
        for _ in range(ncap):
            ln = int(rs.randint(3, seq_length + 1))
            row = np.zeros(seq_length, np.int64)
            row[:ln] = rs.randint(1, vocab + 1, size=ln)
            labels.append(row)
        end.append(len(labels))
        split = ('train' if i < val_from else 'val') if val_from is not None else \
            ('train' if i < n - 3 else ('val' if i < n - 1 else 'restval'))
        images.append({'id': iid, 'split': split,
                       'file_path': f'x/{iid}.jpg'})
    np.savez(os.path.join(tmp, 'labels.npz'), labels=np.stack(labels), label_start_ix=np.array(start), label_end_ix=np.array(end))
    with open(os.path.join(tmp, 'data.json'), 'w') as f:
        json.dump({'ix_to_word': {str(i): f'w{i}' for i in range(1, vocab + 1)}, 'images': images}, f)
    opt = argparse.Namespace(input_json=os.path.join(tmp, 'data.json'), input_label_h5=os.path.join(tmp, 'labels.npz'),
                             input_fc_dir=fc_dir, input_att_dir=att_dir, batch_size=4, seq_per_img=2, train_only=0,
                             use_att=True, use_fc=True, pin_memory=0)
    return opt, images, feats, np.stack(labels), np.array(start), np.array(end)
