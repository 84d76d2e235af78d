"""CPU: the data layer (cooperativeimagecaptioning_amd/dataloader.py) against the reference's get_batch contract
(dataloader.py:171-245) on a small dataset in the reference's on-disk formats: per-image fc `.npy` and att
`.npz['feat']` files with ragged region counts, labels / label_start_ix / label_end_ix arrays, a json with ix_to_word
and splits.  The h5 label container itself cannot be read here (h5py is not in the image): the same three arrays come
from an .npz."""
import argparse
import json
import os
import random

import numpy as np
import pytest


from dataset_util import make_dataset as _dataset


@pytest.mark.parametrize('ragged', [True, False])
def test_get_batch_contract(tmp_path, ragged):
    from cooperativeimagecaptioning_amd.dataloader import DataLoader
    opt, images, feats, labels, start, end = _dataset(str(tmp_path), ragged=ragged)
    random.seed(3)
    dl = DataLoader(opt, workers=2)
    assert dl.vocab_size == 29 and dl.seq_length == 16
    assert len(dl.split_ix['train']) == 9 and len(dl.split_ix['val']) == 2       # restval joins train (train_only 0)
    order = list(dl.split_ix['train'])
    seen, wraps = [], 0
    for it in range(3):
        d = dl.get_batch('train')
        B, spi = 4, 2
        ixs = [inf['ix'] for inf in d['infos']]
        seen += ixs
        assert d['fc_feats'].shape == (B * spi, 12) and d['labels'].shape == (B * spi, 18) and d['masks'].shape == (B * spi, 18)
        kmax = max(feats[images[ix]['id']][1].shape[0] for ix in ixs)
        assert d['att_feats'].shape == (B * spi, kmax, 12)
        for i, ix in enumerate(ixs):
            fc, att = feats[images[ix]['id']]
            for q in range(spi):
                np.testing.assert_array_equal(d['fc_feats'][i * spi + q], fc)
                np.testing.assert_array_equal(d['att_feats'][i * spi + q, :att.shape[0]], att)
                assert not d['att_feats'][i * spi + q, att.shape[0]:].any()            # zero padding
                if d['att_masks'] is not None:
                    np.testing.assert_array_equal(d['att_masks'][i * spi + q], (np.arange(kmax) < att.shape[0]).astype('float32'))
                lab = d['labels'][i * spi + q]
                assert lab[0] == 0 and lab[17] == 0                                     # <bos> / <eos> columns
                cand = labels[start[ix] - 1:end[ix]]
                assert any(np.array_equal(lab[1:17], c) for c in cand)                  # one of the image's own captions
                nnz = int((lab != 0).sum())
                np.testing.assert_array_equal(d['masks'][i * spi + q], (np.arange(18) < nnz + 2).astype('float32'))
            np.testing.assert_array_equal(d['gts'][i], cand)
        if not ragged:
            assert d['att_masks'] is None                                               # all images the same length (:228-229)
        elif len({feats[images[ix]['id']][1].shape[0] for ix in ixs}) > 1:
            assert d['att_masks'] is not None
        wraps += int(d['bounds']['wrapped'])
        assert d['bounds']['it_max'] == 9
    assert seen[:9] == order and wraps == 1                 # first epoch in split order, wrapped once in 12 images
    assert set(seen[9:]) <= set(order)
    dl.close()


def test_state_dict_resumes_the_stream(tmp_path):
    from cooperativeimagecaptioning_amd.dataloader import DataLoader
    opt, *_ = _dataset(str(tmp_path))
    random.seed(5)
    a = DataLoader(opt, workers=2)
    a.get_batch('train')
    st = a.state_dict()
    nxt = [inf['ix'] for inf in a.get_batch('train')['infos']]
    b = DataLoader(opt, workers=2)
    b.load_state_dict(json.loads(json.dumps(st)))           # survives the JSON infos record
    assert [inf['ix'] for inf in b.get_batch('train')['infos']] == nxt
    assert a.state_dict(rewind=1)['iterators']['train'] == st['iterators']['train']
    a.close(), b.close()


def test_val_split_and_unsupported_stores(tmp_path):
    from cooperativeimagecaptioning_amd.dataloader import DataLoader, HybridLoader
    opt, images, feats, *_ = _dataset(str(tmp_path))
    dl = DataLoader(opt, workers=1)
    d = dl.get_batch('val', batch_size=2)
    assert [inf['id'] for inf in d['infos']] == [images[8]['id'], images[9]['id']] and d['bounds']['wrapped']
    dl.close()
    with pytest.raises(NotImplementedError):
        HybridLoader('features.lmdb', '.npy')


def test_packed_store_hands_out_the_same_batches(tmp_path):
    """tools/pack_features.py -> one memory-mapped .npy per feature kind: the loader reads rows of the mapping instead of
    per-image files and must hand out, batch for batch, what the per-image loader hands out."""
    import subprocess
    import sys
    from cooperativeimagecaptioning_amd.dataloader import DataLoader
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    opt, images, feats, labels, start, end = _dataset(str(tmp_path), ragged=False)
    packed = {}
    for kind, src in (('fc', opt.input_fc_dir), ('att', opt.input_att_dir)):
        out = os.path.join(str(tmp_path), kind + '_packed.npy')
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'pack_features.py'), src, out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        packed[kind] = out
    opt_p = argparse.Namespace(**vars(opt))
    opt_p.input_fc_dir, opt_p.input_att_dir = packed['fc'], packed['att']
    random.seed(5)
    a = DataLoader(opt, workers=2)
    random.seed(5)
    b = DataLoader(opt_p, workers=2)
    assert b.att_loader.packed is not None and b.fc_loader.packed is not None
    for _ in range(4):                                   # over an epoch boundary (9 training images, 4 per batch)
        random.seed(11)
        da = a.get_batch('train')
        random.seed(11)
        db = b.get_batch('train')
        for k in ('fc_feats', 'att_feats', 'labels', 'masks'):
            np.testing.assert_array_equal(da[k], db[k], err_msg=k)
        assert da['att_masks'] is None and db['att_masks'] is None
        assert [i['ix'] for i in da['infos']] == [i['ix'] for i in db['infos']]
    a.close()
    b.close()


def test_rank_sharded_train_split_is_disjoint_and_covers_every_epoch(tmp_path):
    """Data-parallel runs: rank r of `world` reads its own images of every global batch.  Over an epoch the
    ranks' images are disjoint and together cover the train split (padded with the order's head to a multiple of world, so
    that all ranks wrap in the same iteration); the shuffle at the wrap is the same permutation on every rank and differs
    from epoch to epoch; val stays whole on every rank."""
    from cooperativeimagecaptioning_amd.dataloader import DataLoader
    opt, images, *_ = _dataset(str(tmp_path), n=26, val_from=23)          # 23 train images, 3 val
    opt.batch_size, opt.seq_per_img = 3, 1
    world = 2
    random.seed(1)
    dls = [DataLoader(opt, workers=1, rank=r, world=world) for r in range(world)]
    train = [ix for ix, im in enumerate(images) if im['split'] == 'train']
    per_rank = (len(train) + world - 1) // world                            # 12: the order is padded by one image
    orders = []
    for epoch in range(3):
        seen = [[] for _ in range(world)]
        wrapped = [False] * world
        while not all(wrapped):
            for r, dl in enumerate(dls):
                random.seed(100 + r)                                        # process-local state must not matter
                d = dl.get_batch('train')
                assert d['bounds']['it_max'] == per_rank
                seen[r] += [inf['ix'] for inf in d['infos']]
                wrapped[r] = wrapped[r] or d['bounds']['wrapped']
            assert len(set(wrapped)) == 1, 'the ranks must wrap in the same iteration'
        # the batch that wraps continues into the next epoch: cut at the epoch's length
        epoch_seen = [s[:per_rank] for s in seen]
        carried = [s[per_rank:] for s in seen]
        both = epoch_seen[0] + epoch_seen[1]
        assert len(set(epoch_seen[0]) & set(epoch_seen[1])) <= (-len(train)) % world, 'disjoint shards (but for the padding)'
        assert set(both) == set(train), f'epoch {epoch}: the shards together cover the split'
        assert len(both) - len(set(both)) == (-len(train)) % world, 'only the padding repeats'
        orders.append(tuple(x for pair in zip(*epoch_seen) for x in pair))
        for dl, c in zip(dls, carried):                                      # put the carried images back: restart the epoch view
            dl.load_state_dict(dict(dl.state_dict(), iterators=dict(dl.iterators, train=0)))
        assert dls[0].split_ix['train'] == dls[1].split_ix['train'], 'every rank holds the same epoch order'
    assert orders[0][:len(train)] == tuple(train), 'first epoch in split order (dealt round robin)'
    assert orders[1] != orders[0] and orders[2] != orders[1], 'reshuffled at every wrap'
    va = [[inf['ix'] for inf in dl.get_batch('val', batch_size=3)['infos']] for dl in dls]
    assert va[0] == va[1] and len(va[0]) == 3                               # evaluation splits are not sharded
    for dl in dls:
        dl.close()


def test_snapshot_resume_replays_the_prefetched_batch_across_an_epoch_wrap(tmp_path):
    """ADVICE (round 2): the state saved while a prefetcher holds batches ahead of the trainer is the snapshot taken when
    the OLDEST unconsumed batch was drawn - not arithmetic on the current position, which loses the tail of an epoch
    once the read-ahead has crossed the wrap (iterator reset + reshuffle)."""
    from cooperativeimagecaptioning_amd.dataloader import DataLoader
    opt, *_ = _dataset(str(tmp_path))                                       # 9 training images, batches of 4
    random.seed(7)
    a = DataLoader(opt, workers=1)
    a.get_batch('train')
    h2 = a.begin_batch('train')                                             # images 4..7
    h3 = a.begin_batch('train')                                             # image 8 + wrap + 3 of the reshuffled epoch
    want2 = [inf['ix'] for inf in a.end_batch(h2)['infos']]
    want3 = [inf['ix'] for inf in a.end_batch(h3)['infos']]
    assert h3['data']['bounds']['wrapped']
    for handle, want in ((h2, want2), (h3, want3)):
        st = json.loads(json.dumps(a.state_dict(snapshot=handle['state'])))
        b = DataLoader(opt, workers=1)
        b.load_state_dict(st)
        got = [inf['ix'] for inf in b.get_batch('train')['infos']]
        if handle is h2:
            assert got == want
        else:
            assert got[0] == want[0], 'the tail of the old epoch is replayed, not dropped'
        b.close()
    # a val batch assembled while a train batch is held does not touch the train batch's pinned buffers
    held = a.begin_batch('train')
    before = a.end_batch(held)['att_feats'].copy()
    a.get_batch('val', batch_size=2)
    np.testing.assert_array_equal(held['data']['att_feats'], before)
    a.close()
