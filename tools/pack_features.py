#!/usr/bin/env python3
"""Packs a directory of per-image feature files (the reference's layout: <id>.npy or <id>.npz['feat']) into ONE `.npy`
array [N, ...] + `<out>.index.json` {image key: row}, the memory-mappable store cooperativeimagecaptioning_amd.dataloader
reads when --input_fc_dir / --input_att_dir name such a file.  Images with fewer regions than the largest are zero-padded
(the loader recovers nothing from the padding: use this for fixed-size region sets such as 36 bottom-up boxes).
usage: pack_features.py <feature_dir> <out.npy> [--dtype float32]"""
import json
import os
import sys

import numpy as np


def main():
    src, out = sys.argv[1], sys.argv[2]
    dtype = np.dtype(sys.argv[sys.argv.index('--dtype') + 1]) if '--dtype' in sys.argv else np.dtype('float32')
    files = sorted(f for f in os.listdir(src) if f.endswith(('.npy', '.npz')))
    assert files, f'no .npy / .npz files under {src}'

    def read(f):
        a = np.load(os.path.join(src, f))
        return a if f.endswith('.npy') else a['feat']
    first = read(files[0])
    shapes = [first.shape]
    if first.ndim >= 2:                                   # region features: find the largest region count
        first = first.reshape(-1, first.shape[-1])
        kmax = max(read(f).reshape(-1, first.shape[-1]).shape[0] for f in files)
        row_shape = (kmax, first.shape[-1])
    else:
        row_shape = first.shape
    arr = np.lib.format.open_memmap(out, mode='w+', dtype=dtype, shape=(len(files),) + tuple(row_shape))
    index = {}
    for i, f in enumerate(files):
        a = read(f)
        if len(row_shape) == 2:
            a = a.reshape(-1, row_shape[1])
            assert a.shape[0] == row_shape[0], (f'{f}: {a.shape[0]} regions, the store holds {row_shape[0]} per image: '
                                                'ragged region counts need the per-image files')
        arr[i] = a
        index[os.path.splitext(f)[0]] = i
    arr.flush()
    with open(out + '.index.json', 'w') as fo:
        json.dump(index, fo)
    print(f'{len(files)} images -> {out} {arr.shape} {dtype}, {arr.nbytes / 2**20:.1f} MiB')


if __name__ == '__main__':
    main()
