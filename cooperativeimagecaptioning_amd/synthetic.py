"""Synthetic COCO-shaped batches and reference-default options: what bench.py,
smoke() and the scaling tests feed the step when no dataset is present."""
import argparse

import numpy as np
import torch


def default_opt(**kw):
    """The fields the model constructors read, at the reference's run_joint.sh defaults
    (bash_scripts/run_joint.sh:36-61,285-326; opts.py)."""
    d = dict(vocab_size=9487, input_encoding_size=512, rnn_size=512, num_layers=1, drop_prob_lm=0.5,
             seq_length=16, fc_feat_size=2048, att_feat_size=2048, att_hid_size=512,
             retrieval_reward='gumbel', gumbel_temp=1.0, multinomial_temp=1.0,
             prob_gumbel_softmax=0.25, prob_multinomial_soft=0.25, use_bn=0, decoding_constraint=0,
             rnn_type='lstm', caption_model='att2in2', vse_model='fc', share_embed=0, phase=None,
             vse_embed_size=1024, vse_no_imgnorm=0, vse_use_abs=0, vse_num_layers=1,
             vse_rnn_type='gru', vse_pool_type='last', vse_margin=0.2, vse_measure='cosine',
             vse_max_violation=1, vse_loss_type='contrastive', batch_size=128, vse_loss_weight=0,
             caption_loss_weight=0, alternating_turn=['speaker', 'listener'],
             retrieval_reward_weight=0.01, reinforce_baseline_type='gt', only_one_retrieval='off',
             cider_optimization=0.99, use_gen_cider_scores=0, is_alternating=1, start_from=None,
             initialize_retrieval=None, df='corpus', continue_from_existing_models=False,
             learning_rate=5e-4, weight_decay=0.0, grad_clip=0.1, seq_per_img=1, cached_tokens='corpus')
    d.update(kw)
    return argparse.Namespace(**d)


def make_batch(opt, K=36, seed=1234, ncap=5, device='cpu'):
    """att_feats ~ 0.5*|N(0,1)| (ReLU-like pooled CNN features), fc = mean over regions, labels with
    Zipf(1.1) tokens and lengths U{6..16}, 5 ground-truth captions per image (dataloader.py:171-245
    output contract)."""
    g = torch.Generator().manual_seed(seed)
    B, V, SL = opt.batch_size, opt.vocab_size, opt.seq_length
    att = torch.randn(B, K, opt.att_feat_size, generator=g).abs() * 0.5
    fc = att.mean(1)
    rs = np.random.RandomState(seed)

    def cap():
        ln = rs.randint(6, SL + 1)
        row = np.zeros(SL, np.int64)
        row[:ln] = np.minimum(rs.zipf(1.1, size=ln), V)
        return row
    gts = [np.stack([cap() for _ in range(ncap)]) for _ in range(B)]
    labels = np.zeros((B, SL + 2), np.int64)
    masks = np.zeros((B, SL + 2), np.float32)
    for i in range(B):
        labels[i, 1:SL + 1] = gts[i][0]
        masks[i, :int((gts[i][0] > 0).sum()) + 2] = 1
    return dict(fc_feats=fc.to(device), att_feats=att.to(device), att_masks=None,
                labels=torch.from_numpy(labels).to(device), masks=torch.from_numpy(masks).to(device), gts=gts)


class SyntheticLoader:
    """get_batch('train') with the reference loader's output contract (dataloader.py:171-245): numpy
    arrays on the host, `bounds.wrapped` every `iters_per_epoch` calls.  The synthetic "dataset" is a pool of `pool`
    distinct random batches generated on first use and served round robin (a real dataset also repeats every epoch;
    drawing 9.4 M normals per call costs 50-150 ms of host time, an order of magnitude more than the step);
    pool = 0 draws a fresh batch on every call."""

    def __init__(self, opt, seed=1234, K=36, iters_per_epoch=100, pool=8, pin=True, val_batches=2):
        self.opt, self.seed, self.K, self.n, self.ipe = opt, seed, K, 0, iters_per_epoch
        self.vocab_size, self.seq_length = opt.vocab_size, opt.seq_length
        self.pool = int(pool)
        self.pin = bool(pin) and torch.cuda.is_available()      # pool batches live in page-locked memory: asynchronous uploads
        self._cache = {}
        # what the evaluation drivers read from a loader (eval_utils.eval_split / encode_data): a small 'val' / 'test' split
        # of `val_batches` batches of its own, one caption row per image
        self.batch_size, self.seq_per_img, self.dataset = opt.batch_size, 1, 'coco'
        self.val_batches = int(val_batches)
        self._eval_pos = {'val': 0, 'test': 0}
        self.ix_to_word = None

    def get_vocab(self):
        if self.ix_to_word is None:
            self.ix_to_word = {str(i): f'w{i}' for i in range(1, self.vocab_size + 1)}
        return self.ix_to_word

    def reset_iterator(self, split):
        if split in self._eval_pos:
            self._eval_pos[split] = 0
        else:
            self.n = 0

    def _eval_batch(self, split):
        i = self._eval_pos[split]
        key = (split, i)
        if key not in self._cache:
            self._cache[key] = self._make(1000003 * (1 + list(self._eval_pos).index(split)) + i)
        b = dict(self._cache[key])
        B = self.batch_size
        if self.seq_per_img > 1:        # encode_data asks for 5 caption rows per image (eval_utils.py:300-304)
            spi = self.seq_per_img
            rep = lambda a: np.repeat(a, spi, axis=0)        # noqa: E731
            lab = np.zeros((B * spi, self.seq_length + 2), np.int64)
            msk = np.zeros((B * spi, self.seq_length + 2), np.float32)
            for k in range(B):
                for q in range(spi):
                    cap = b['gts'][k][q % len(b['gts'][k])]
                    lab[k * spi + q, 1:self.seq_length + 1] = cap
                    msk[k * spi + q, :int((cap > 0).sum()) + 2] = 1
            b.update(fc_feats=rep(b['fc_feats']), att_feats=rep(b['att_feats']), labels=lab, masks=msk)
        self._eval_pos[split] = (i + 1) % self.val_batches
        b['bounds'] = dict(it_pos_now=(i + 1) * B if i + 1 < self.val_batches else 0, it_max=self.val_batches * B,
                           wrapped=(i + 1 == self.val_batches))
        b['infos'] = [{'ix': i * B + k, 'id': 900000 * (1 + list(self._eval_pos).index(split)) + i * B + k, 'file_path': ''}
                      for k in range(B)]
        return b

    def state_dict(self, rewind=0):
        """Position of the stream (the reference keeps loader.iterators in infos, train.py:312); rewind: batches
        already handed out that the caller has not consumed."""
        return dict(n=max(0, int(self.n) - int(rewind)))

    def load_state_dict(self, st):
        self.n = int(st.get('n', 0))

    def _make(self, idx):
        b = make_batch(self.opt, K=self.K, seed=self.seed + idx)
        host = (lambda t: t.pin_memory().numpy()) if self.pin and self.pool > 0 else (lambda t: t.numpy())
        return dict(fc_feats=host(b['fc_feats']), att_feats=host(b['att_feats']), att_masks=None,
                    labels=host(b['labels']), masks=host(b['masks']), gts=b['gts'])

    def get_batch(self, split):
        if split in self._eval_pos:
            return self._eval_batch(split)
        if self.pool > 0:
            idx = self.n % self.pool
            if idx not in self._cache:
                self._cache[idx] = self._make(idx)
            b = dict(self._cache[idx])          # shallow copy: per-call keys (bounds, loader attachments) stay per call
        else:
            b = self._make(self.n)
        self.n += 1
        b['bounds'] = dict(it_pos_now=self.n, it_max=self.ipe, wrapped=(self.n % self.ipe == 0))
        b['infos'] = []
        return b
