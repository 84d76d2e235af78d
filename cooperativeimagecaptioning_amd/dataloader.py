"""The reference's data layer (dataloader.py:18-369) for the MI355X trainer: the same on-disk formats, the same
``get_batch(split)`` output contract, built for a step that takes a few milliseconds.

What the reference does                                  what happens here
  per-image fc `.npy` / att `.npz['feat']` files          the same files, read with numpy alone (HybridLoader, :18-53)
  labels / label_start_ix / label_end_ix in an h5 file    the same three arrays from an `.npz` (or `.h5` when h5py exists)
  4 worker PROCESSES, one image per message (:332-338)    a pool of reader THREADS (file reads release the GIL) that
                                                          runs a window of images ahead of the trainer
  np.stack / np.zeros batch assembly, then .cuda()        images are written straight into PINNED batch buffers (three
                                                          rotating sets): the assembly copy is the staging copy, and
                                                          prefetch.PrefetchLoader's upload is a true asynchronous DMA
lmdb feature stores are not supported (the library is not in the image).

Per-image files cannot feed a 4 ms step (128 images x 295 KB through zip / np.load per batch): a PACKED store - one
`.npy` array [N, ...] per feature kind next to `<file>.index.json` {image key: row}, written once by
tools/pack_features.py from the per-image files - is memory-mapped instead, and a batch is then 128 row copies out of the
page cache, done by the reader threads in parallel straight into the pinned batch buffer."""
import json
import os
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


class HybridLoader:
    """dataloader.py:18-53: `.npy` files hold the array, anything else is an `.npz` with the array under 'feat'."""

    def __init__(self, db_path, ext):
        self.db_path, self.ext = db_path, ext
        self.packed = None
        if db_path.endswith('.npy') and os.path.isfile(db_path):
            # packed store (this repository's extension): one memory-mapped array, rows named by the index file
            self._mmap = np.load(db_path, mmap_mode='r')
            self.packed = self._mmap.view(np.ndarray)      # plain views of the mapping: np.memmap's per-slice bookkeeping costs ~10 us
            with open(db_path + '.index.json') as f:
                self.index = {str(k): int(v) for k, v in json.load(f).items()}
            return
        if db_path.endswith('.lmdb'):
            raise NotImplementedError('lmdb feature stores need the lmdb package, which is not available here: '
                                      'use a directory of per-image .npy / .npz files')
        self.loader = (lambda x: np.load(x)) if ext == '.npy' else (lambda x: np.load(x)['feat'])

    def get(self, key):
        if self.packed is not None:
            return self.packed[self.index[key]]            # a view of the mapping: copied when the batch is assembled
        return self.loader(os.path.join(self.db_path, key + self.ext))


def _load_labels(path):
    """-> (labels int[N, seq_length], label_start_ix, label_end_ix), the three datasets of prepro_labels.py's h5 file."""
    if path.endswith('.npz'):
        z = np.load(path, allow_pickle=False)
        return z['labels'], z['label_start_ix'], z['label_end_ix']
    try:
        import h5py
    except ImportError as e:
        raise NotImplementedError(f'{path}: reading an .h5 label file needs h5py (not installed); convert it once with '
                                  f'np.savez(labels=..., label_start_ix=..., label_end_ix=...)') from e
    with h5py.File(path, 'r') as f:
        return f['labels'][:], f['label_start_ix'][:], f['label_end_ix'][:]


def pinned_empty(shape, dtype, pin=True):
    """A numpy array over page-locked host memory (a view of a pinned torch tensor) when a GPU is there, else plain."""
    t = torch.empty(tuple(shape), dtype=dtype)
    if pin and torch.cuda.is_available():
        t = t.pin_memory()
    return t.numpy()


class DataLoader:
    """dataloader.py:56-294.  opt: input_json, input_label_h5 ('none' = no labels), input_fc_dir, input_att_dir,
    batch_size, seq_per_img, train_only, use_att / use_fc / norm_att_feat as in the reference."""

    N_BUFFERS = 3        # batch i is consumed by the step, i+1 is uploading, i+2 is being filled

    def __init__(self, opt, workers=None, window=None, rank=None, world=None):
        self.opt = opt
        # data-parallel runs (rank r takes its own images of every global batch): the TRAIN split is dealt
        # round robin over the ranks - rank r reads positions r, r + world, ... of the epoch's order, which is the same
        # list on every rank (the shuffle at an epoch wrap is seeded by (loader_seed, epoch), not by process state).  The
        # order is padded to a multiple of `world` with its own head, so every rank wraps in the same iteration.
        # val / test stay whole: evaluation runs on one rank, as in the reference.
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        assert 0 <= self.rank < self.world, (self.rank, self.world)
        self.loader_seed = int(getattr(opt, 'loader_seed', getattr(opt, 'seed', 0)) or 0)
        self.sharded_splits = ('train',) if self.world > 1 else ()
        self.epochs = {'train': 0, 'val': 0, 'test': 0}
        self._shard_cache = {}
        self.batch_size = opt.batch_size
        self.seq_per_img = opt.seq_per_img
        self.dataset = getattr(opt, 'dataset', 'coco')
        self.use_fc = getattr(opt, 'use_fc', True)
        self.use_att = getattr(opt, 'use_att', True)
        self.norm_att_feat = getattr(opt, 'norm_att_feat', 0)
        print('DataLoader loading json file: ', opt.input_json)
        with open(opt.input_json) as f:
            self.info = json.load(f)
        if 'ix_to_word' in self.info:
            self.ix_to_word = self.info['ix_to_word']
            self.vocab_size = len(self.ix_to_word)
            print('vocab size is ', self.vocab_size)
        self.has_labels = opt.input_label_h5 != 'none'
        if self.has_labels:
            self.label, self.label_start_ix, self.label_end_ix = _load_labels(opt.input_label_h5)
            self.seq_length = self.label.shape[1]
            print('max sequence length in data is', self.seq_length)
        else:
            self.seq_length = 1
        self.fc_loader = HybridLoader(opt.input_fc_dir, '.npy')
        self.att_loader = HybridLoader(opt.input_att_dir, '.npz')
        self.num_images = len(self.info['images'])
        print('read %d image features' % self.num_images)
        self.split_ix = {'train': [], 'val': [], 'test': []}                       # :118-134
        for ix, img in enumerate(self.info['images']):
            if 'split' not in img:
                for s in self.split_ix:
                    self.split_ix[s].append(ix)
            elif img['split'] in self.split_ix:
                self.split_ix[img['split']].append(ix)
            elif getattr(opt, 'train_only', 0) == 0:                                # restval
                self.split_ix['train'].append(ix)
        for s in ('train', 'val', 'test'):
            print('assigned %d images to split %s' % (len(self.split_ix[s]), s))
        self.iterators = {'train': 0, 'val': 0, 'test': 0}
        if workers is None:                                 # the reference runs 4 loader processes (:337); threads are cheaper
            workers = max(2, min(8, (os.cpu_count() or 4) // 2))
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix='cic-reader')
        self._workers = workers
        self._packed = self.att_loader.packed is not None and self.fc_loader.packed is not None
        self._window = window or 2 * self.batch_size
        self._queue = {s: [] for s in self.split_ix}      # [(ix, wrapped, future)] read ahead, in hand-out order
        self._cursor = dict(self.iterators)               # position of the read-ahead (>= iterators)
        self._buffers = {}
        self._turn = {s: 0 for s in self.split_ix}        # per split: a val batch never lands in a train batch's buffers

    # ---- reference accessors (:58-70)
    def get_vocab_size(self):
        return self.vocab_size

    def get_vocab(self):
        return self.ix_to_word

    def get_seq_length(self):
        return self.seq_length

    def reset_iterator(self, split):
        for _, _, fut in self._queue[split]:
            if fut is not None:
                fut.cancel()
        self._queue[split] = []
        self.iterators[split] = 0
        self._cursor[split] = 0

    # ---- the rank's view of a split
    def _order(self, split):
        """Positions this rank reads, in hand-out order: the split's order itself, or every world-th entry of it."""
        order = self.split_ix[split]
        if split not in self.sharded_splits:
            return order
        cached = self._shard_cache.get(split)
        if cached is not None and cached[0] is order:
            return cached[1]
        pad = (-len(order)) % self.world
        mine = (list(order) + list(order[:pad]))[self.rank::self.world]
        self._shard_cache[split] = (order, mine)
        return mine

    def _shuffle(self, split):
        """dataloader.py:352-353: the train split is reshuffled when an epoch wraps.  One process: the module-level
        `random`, as the reference.  Data-parallel: a generator seeded by (loader_seed, epoch) - every rank draws the
        same permutation without talking to the others.  The list is REPLACED, never shuffled in place: snapshots taken
        for a resume keep pointing at the order their batch was drawn from."""
        order = list(self.split_ix[split])
        self.epochs[split] += 1
        if self.world > 1:
            random.Random(self.loader_seed * 1000003 + self.epochs[split]).shuffle(order)
        else:
            random.shuffle(order)
        self.split_ix[split] = order

    # ---- resume (the reference keeps iterators / split_ix in infos, train.py:312-313,363-364)
    def _snapshot(self):
        """The stream's position NOW (cheap: the orders are shared, not copied - _shuffle replaces them)."""
        return dict(iterators=dict(self.iterators), split_ix=dict(self.split_ix), epochs=dict(self.epochs))

    def state_dict(self, rewind=0, snapshot=None):
        """snapshot: the handle['state'] of the oldest batch handed out by begin_batch() that the trainer has not consumed
        (a prefetcher's): resuming from it replays that batch, across an epoch wrap too.  rewind (batches, kept for
        loaders driven without handles) steps the train iterator back inside the current epoch."""
        snap = snapshot if snapshot is not None else self._snapshot()
        it = dict(snap['iterators'])
        if snapshot is None and rewind:
            it['train'] = max(0, it['train'] - int(rewind) * self.batch_size)
        return dict(iterators=it, split_ix={k: list(v) for k, v in snap['split_ix'].items()}, epochs=dict(snap['epochs']),
                    rank=self.rank, world=self.world)

    def load_state_dict(self, st):
        for s in self.split_ix:
            for _, _, fut in self._queue[s]:
                if fut is not None:
                    fut.cancel()
            self._queue[s] = []
        if st.get('split_ix'):
            self.split_ix = {k: list(v) for k, v in st['split_ix'].items()}
        self.epochs.update({k: int(v) for k, v in (st.get('epochs') or {}).items()})
        its = dict(st.get('iterators', {}))
        if int(st.get('world', self.world)) != self.world:
            # iterators count positions of a rank's own view: under another world size the epoch restarts
            print(f"loader state was saved with world={st.get('world')}, resuming with world={self.world}: "
                  f"the train iterator restarts at the beginning of the saved epoch's order")
            its['train'] = 0
        self.iterators.update(its)
        self._cursor = dict(self.iterators)

    # ---- one image (:250-294)
    def _image_key(self, ix):
        img = self.info['images'][ix]
        if self.dataset in ('flickr8k', 'flickr30k'):
            return str(img['file_path'].split('/')[1].split('.')[0])
        return str(img['id'])

    def __getitem__(self, ix):
        if self.use_att:
            att = self.att_loader.get(self._image_key(ix))
            att = att.reshape(-1, att.shape[-1])                                   # K x C
            if self.norm_att_feat:
                att = att / np.linalg.norm(att, 2, 1, keepdims=True)
        else:
            att = np.zeros((1, 1), dtype='float32')
        fc = self.fc_loader.get(self._image_key(ix)) if self.use_fc else np.zeros((1,), dtype='float32')
        return fc, att, ix

    def __len__(self):
        return len(self.info['images'])

    def get_captions(self, ix, seq_per_img):
        """:154-172 (sampling with the module-level `random`, as the reference)."""
        ix1 = self.label_start_ix[ix] - 1                  # label_start_ix starts from 1
        ix2 = self.label_end_ix[ix] - 1
        ncap = ix2 - ix1 + 1
        assert ncap > 0, 'an image does not have any label. this can be handled but right now isn\'t'
        if ncap < seq_per_img:                             # subsample with replacement
            seq = np.zeros([seq_per_img, self.seq_length], dtype='int')
            for q in range(seq_per_img):
                seq[q, :] = self.label[random.randint(ix1, ix2), :self.seq_length]
            return seq
        ixl = random.randint(ix1, ix2 - seq_per_img + 1)
        return self.label[ixl: ixl + seq_per_img, :self.seq_length]

    # ---- read-ahead in hand-out order (BlobFetcher._get_next_minibatch_inds, :343-358)
    def _advance(self, split):
        order = self._order(split)
        max_index = len(order)
        ri = self._cursor[split]
        ix = order[ri]
        ri_next, wrapped = ri + 1, False
        if ri_next >= max_index:
            ri_next, wrapped = 0, True
        self._cursor[split] = ri_next
        return ix, wrapped

    def _fill(self, split, need):
        q = self._queue[split]
        while len(q) < need:
            if q and q[-1][1]:
                break                                       # the epoch ended: the next one is scheduled after its shuffle
            ix, wrapped = self._advance(split)
            # per-image files are read ahead by the pool; rows of a packed store are views of the mapping: nothing to read yet
            q.append((ix, wrapped, None if self._packed else self._pool.submit(self.__getitem__, ix)))

    def _next(self, split):
        self._fill(split, 1)
        ix, wrapped, fut = self._queue[split].pop(0)
        if wrapped and split == 'train':
            self._shuffle(split)                            # :352-353: shuffled when the epoch wraps
        self.iterators[split] = 0 if wrapped else self.iterators[split] + 1
        fc, att, ix2 = fut.result() if fut is not None else self.__getitem__(ix)
        assert ix2 == ix, 'ix not equal'
        return fc, att, ix, wrapped

    def _buffer(self, name, split, shape, dtype):
        key = (name, split, self._turn[split] % self.N_BUFFERS)
        b = self._buffers.get(key)
        if b is None or b.shape != tuple(shape):
            b = self._buffers[key] = pinned_empty(shape, dtype, pin=getattr(self.opt, 'pin_memory', 1))
        return b

    # ---- the batch (:174-248), in two phases so that a prefetcher can keep the large copies off the training thread:
    # begin_batch() does the bookkeeping (which images, captions, labels, masks) and hands the feature copies into the
    # pinned buffers to the reader threads; end_batch() waits for them.  get_batch() = both, the reference's call.
    def get_batch(self, split, batch_size=None):
        return self.end_batch(self.begin_batch(split, batch_size))

    def end_batch(self, handle):
        for f in handle['copies']:
            f.result()
        return handle['data']

    def begin_batch(self, split, batch_size=None):
        batch_size = batch_size or self.batch_size
        spi = self.seq_per_img
        state = self._snapshot()                            # where a resume must start to hand out THIS batch again
        n_mine = len(self._order(split))
        self._fill(split, min(self._window + batch_size, n_mine))
        fcs, atts, label_batch, gts, infos = [], [], [], [], []
        wrapped = False
        for _ in range(batch_size):
            fc, att, ix, w = self._next(split)
            wrapped = wrapped or w
            fcs.append(fc)
            atts.append(att)
            lab = np.zeros([spi, self.seq_length + 2], dtype='int')
            if self.has_labels:
                lab[:, 1:self.seq_length + 1] = self.get_captions(ix, spi)
                gts.append(self.label[self.label_start_ix[ix] - 1: self.label_end_ix[ix]])
            else:
                gts.append([])
            label_batch.append(lab)
            img = self.info['images'][ix]
            infos.append({'ix': ix, 'id': img['id'], 'file_path': img.get('file_path', '')})
        self._fill(split, min(self._window, n_mine))       # keep the readers busy under the step
        self._turn[split] += 1
        data = {}
        fcb = self._buffer('fc', split, (batch_size * spi,) + tuple(fcs[0].shape), torch.float32)
        data['fc_feats'] = fcb
        max_att_len = max(a.shape[0] for a in atts)
        attb = self._buffer('att', split, (batch_size * spi, max_att_len, atts[0].shape[1]), torch.float32)
        masks_att = np.zeros(attb.shape[:2], dtype='float32')
        for i, a in enumerate(atts):
            masks_att[i * spi:(i + 1) * spi, :a.shape[0]] = 1

        def place(i):                                       # numpy releases the GIL inside these copies
            a = atts[i]
            fcb[i * spi:(i + 1) * spi] = fcs[i]
            attb[i * spi:(i + 1) * spi, :a.shape[0]] = a
            if a.shape[0] < max_att_len:
                attb[i * spi:(i + 1) * spi, a.shape[0]:] = 0
        # the feature rows go into the pinned buffers on the reader threads, a slice of the batch per worker (rows of a
        # packed store come out of the page cache here; per-image files were read ahead and are only copied)
        data['att_feats'] = attb
        data['att_masks'] = None if masks_att.sum() == masks_att.size else masks_att      # :228-229
        labels = self._buffer('labels', split, (batch_size * spi, self.seq_length + 2), torch.int64)
        labels[:] = np.vstack(label_batch)
        data['labels'] = labels
        nonzeros = (labels != 0).sum(1) + 2
        mask_batch = self._buffer('masks', split, (batch_size * spi, self.seq_length + 2), torch.float32)
        mask_batch[:] = (np.arange(self.seq_length + 2)[None, :] < nonzeros[:, None]).astype('float32')
        data['masks'] = mask_batch
        data['gts'] = gts
        data['bounds'] = {'it_pos_now': self.iterators[split], 'it_max': n_mine, 'wrapped': wrapped}
        data['infos'] = infos
        # submitted last: the bookkeeping above does not then share the interpreter with the copy threads
        n, w = len(atts), self._workers
        copies = [self._pool.submit(lambda c=c: [place(i) for i in range(c, n, w)]) for c in range(min(w, n))]
        return {'data': data, 'copies': copies, 'state': state}

    def close(self):
        self._pool.shutdown(wait=False, cancel_futures=True)
