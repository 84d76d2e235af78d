#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

Imports the reference (vgilad/CooperativeImageCaptioning) UNMODIFIED through the
compat harness of SURVEY.md Appendix B, drives it with seeded inputs, records every
random draw it makes (dropout keep masks, Gumbel uniforms, multinomial picks,
partial-sampling row uniforms) and writes inputs + weights + noise + outputs as small
.npz fixtures under tests/golden/.  Nothing of the reference's source travels: the
fixtures are data only.  tests/test_oracle_golden.py replays them against oracle/.

Usage:  python tools/gen_golden.py            (rewrites tests/golden/*.npz)
        python tools/gen_golden.py --only-masks   (the att_masks cases only)
        python tools/gen_golden.py --only-fullwidth   (the BASELINE-width joint step only)
"""
import argparse
import os
import pickle
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, os.path.join(REPO, 'tests'))
import golden_util as GU  # noqa: E402


# ----------------------------------------------------------------------------
# compat harness (torch 0.4.1 code on torch 2.x, no heavy deps) — SURVEY.md Appendix B
# ----------------------------------------------------------------------------
def install_harness():
    sys.dont_write_bytecode = True
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    scratch = tempfile.mkdtemp(prefix='cic_golden_')
    os.makedirs(os.path.join(scratch, 'cider', 'data'))
    with open(os.path.join(scratch, 'cider', 'data', 'coco-val.p'), 'wb') as f:
        pickle.dump({}, f)                      # cider_diff unpickles this at import; never used
    os.chdir(scratch)
    for m in ['skimage', 'skimage.io', 'skimage.transform', 'h5py', 'lmdb']:
        sys.modules[m] = types.ModuleType(m)
    import warnings
    warnings.filterwarnings('ignore')
    import scipy.misc
    scipy.misc.imresize = lambda *a, **k: None
    sys.path.insert(0, REF)
    import torch
    orig_getitem = torch.Tensor.__getitem__

    def getitem(self, idx):                     # 0.4.1: zero_dim_tensor[0] is the scalar
        if self.dim() == 0 and isinstance(idx, int) and idx == 0:
            return self
        return orig_getitem(self, idx)
    torch.Tensor.__getitem__ = getitem
    import models                                # noqa: F401
    import misc.rewards as rewards               # noqa: F401
    vm = sys.modules['models.VSEFCModel']
    orig_pack = vm.pack_padded_sequence

    def pack(x, lengths, batch_first=False):
        return orig_pack(x, [int(l) for l in lengths], batch_first=batch_first)
    vm.pack_padded_sequence = pack
    return torch, models, rewards


# ----------------------------------------------------------------------------
# noise recorder: wraps the torch RNG entry points the reference uses
# ----------------------------------------------------------------------------
class Recorder:
    def __init__(self, torch):
        self.torch = torch
        self.events = []
        self.on = False
        self.inject = None                  # torch.Generator: torch.rand draws from it instead of the global generator
        F = torch.nn.functional
        self._dropout = F.dropout
        self._rand = torch.rand
        self._multinomial = torch.multinomial
        self._uniform_ = torch.Tensor.uniform_
        rec = self

        def dropout(input, p=0.5, training=True, inplace=False):
            if not training:
                return input
            if p == 0.0:                       # keep the event stream shape-stable
                if rec.on:
                    rec.events.append(('dropout', torch.ones_like(input)))
                return input
            keep = torch.bernoulli(torch.full_like(input, 1.0 - p))
            if rec.on:
                rec.events.append(('dropout', keep.clone()))
            return input * (keep / (1.0 - p))

        def rand(*a, **k):
            if rec.inject is not None:      # a stream of its own, so that a test can regenerate it call by call
                k = dict(k, generator=rec.inject)
            u = rec._rand(*a, **k)
            if rec.on:
                rec.events.append(('rand', u.clone()))
            return u

        def multinomial(p, n, *a, **k):
            r = rec._multinomial(p, n, *a, **k)
            if rec.on:
                rec.events.append(('multinomial', r.view(-1).clone()))
            return r

        def uniform_(self_, *a, **k):
            r = rec._uniform_(self_, *a, **k)
            if rec.on and self_.dim() == 1:
                rec.events.append(('uniform', r.clone()))
            return r
        F.dropout = dropout
        torch.rand = rand
        torch.multinomial = multinomial
        torch.Tensor.uniform_ = uniform_

    def start(self):
        self.events = []
        self.on = True

    def stop(self):
        self.on = False
        ev, self.events = self.events, []
        return ev


def unpack_rows(torch, packed, att_masks):
    """[N, H] rows in pack_padded_sequence order (AttModel.py:30-36: images sorted by region count, descending,
    then region-major) -> [B, K, H] with ones at the padded rows."""
    lens = att_masks.long().sum(1)
    sorted_lengths, indices = torch.sort(lens, descending=True)
    B, K = att_masks.shape
    out = np.ones((B, K, packed.shape[1]), np.float32)
    n = 0
    for k in range(int(sorted_lengths[0])):
        for i in range(B):
            if int(sorted_lengths[i]) > k:
                out[int(indices[i]), k] = packed[n].numpy()
                n += 1
    assert n == packed.shape[0]
    return out


def split_decodes(events, T, B, E, H, Vp1, att_masks=None, want_steps=False):
    """Split a flat event list into per-decode noise dicts (see oracle/speaker.py).
    A 3-D dropout event (att_embed) opens a new decode; 2-D dropout events then
    alternate x_keep[t], out_keep[t]; rand / multinomial / uniform events belong to
    the step whose x_keep comes next.  With att_masks the att_embed dropout runs on the packed
    valid region rows (pack_wrapper, AttModel.py:44-51): a 2-D event of sum(att_masks) rows."""
    decs = []
    cur = None
    n_packed = int(att_masks.sum()) if att_masks is not None else -1
    assert n_packed != B
    for kind, t in events:
        if kind == 'dropout' and t.dim() == 2 and t.shape[0] == n_packed:
            import torch as _torch
            kind, t = 'dropout', _torch.from_numpy(unpack_rows(_torch, t, att_masks))
        if kind == 'dropout' and t.dim() == 3:
            cur = dict(att_keep=t.numpy(), x_keep=np.ones((T, B, E), np.float32),
                       out_keep=np.ones((T, B, H), np.float32),
                       gumbel_u=np.full((T, B, Vp1), 0.5, np.float32),
                       pick=np.zeros((T, B), np.int64), ps_u=np.ones((T, B), np.float32),
                       _nx=0, _no=0, _usteps=[], has_u=False, has_pick=False, has_ps=False)
            decs.append(cur)
        elif kind == 'dropout':
            if cur['_nx'] == cur['_no']:
                cur['x_keep'][cur['_nx']] = t.numpy()
                cur['_nx'] += 1
            else:
                cur['out_keep'][cur['_no']] = t.numpy()
                cur['_no'] += 1
        elif kind == 'rand':
            cur['gumbel_u'][cur['_nx']] = t.numpy()
            cur['_usteps'].append(cur['_nx'])
            cur['has_u'] = True
        elif kind == 'multinomial':
            cur['pick'][cur['_nx']] = t.numpy()
            cur['has_pick'] = True
        elif kind == 'uniform':
            cur['ps_u'][cur['_nx']] = t.numpy()
            cur['has_ps'] = True
    out = []
    for d in decs:
        o = dict(att_keep=d['att_keep'], x_keep=d['x_keep'], out_keep=d['out_keep'])
        if d['has_u']:
            o['gumbel_u'] = d['gumbel_u']
            if want_steps:
                o['gumbel_u_steps'] = np.array(d['_usteps'], np.int64)
        if d['has_pick']:
            o['pick'] = d['pick']
        if d['has_ps']:
            o['ps_u'] = d['ps_u']
        out.append(o)
    return out


# ----------------------------------------------------------------------------
# configs / inputs
# ----------------------------------------------------------------------------
def make_opt(**kw):
    d = dict(vocab_size=97, input_encoding_size=64, rnn_size=64, num_layers=1, drop_prob_lm=0.0,
             seq_length=16, fc_feat_size=96, att_feat_size=96, att_hid_size=64,
             retrieval_reward='gumbel', gumbel_temp=1.0, multinomial_temp=1.0,
             prob_gumbel_softmax=0.5, prob_multinomial_soft=0.5, use_bn=0, decoding_constraint=0,
             rnn_type='lstm', caption_model='att2in2', vse_model='fc', share_embed=0, phase=None,
             vse_embed_size=128, vse_no_imgnorm=0, vse_use_abs=0, vse_num_layers=1,
             vse_rnn_type='gru', vse_pool_type='last', vse_margin=0.2, vse_measure='cosine',
             vse_max_violation=1, vse_loss_type='contrastive', batch_size=6, vse_loss_weight=0,
             caption_loss_weight=0, alternating_turn=['speaker', 'listener'],
             retrieval_reward_weight=0.01, reinforce_baseline_type='gt', only_one_retrieval='off',
             cider_optimization=0.99, use_gen_cider_scores=0, is_alternating=0, start_from=None,
             initialize_retrieval=None, df='corpus')
    d.update(kw)
    return argparse.Namespace(**d)


def make_batch(torch, opt, K=7, seed=0, ncap=5):
    g = torch.Generator().manual_seed(1234 + seed)
    B, V, SL = opt.batch_size, opt.vocab_size, opt.seq_length
    att = (torch.randn(B, K, opt.att_feat_size, generator=g).abs() * 0.5)
    fc = att.mean(1)
    rs = np.random.RandomState(seed)

    def cap():
        ln = rs.randint(3, SL + 1)
        toks = np.minimum(rs.zipf(1.3, size=ln), V).astype(np.int64)
        row = np.zeros(SL, np.int64)
        row[:ln] = toks
        return row
    gts = [np.stack([cap() for _ in range(ncap)]) for _ in range(B)]
    labels = np.zeros((B, SL + 2), np.int64)
    masks = np.zeros((B, SL + 2), np.float32)
    for i in range(B):
        labels[i, 1:SL + 1] = gts[i][0]
        nnz = int((gts[i][0] > 0).sum())
        masks[i, :nnz + 2] = 1
    return dict(fc_feats=fc, att_feats=att, att_masks=None, labels=torch.from_numpy(labels),
                masks=torch.from_numpy(masks), gts=gts)


MASK_LENS = (7, 3, 5, 4, 7, 6, 3, 5)


def mask_batch(torch, batch):
    """Ragged region counts as dataloader.get_batch builds them (dataloader.py:218-229): features zero-padded to
    the longest image, att_masks = 1 on an image's own rows."""
    B, K, _ = batch['att_feats'].shape
    am = torch.zeros(B, K)
    for i in range(B):
        am[i, :MASK_LENS[i % len(MASK_LENS)]] = 1
    assert int(am.sum(1).max()) == K
    batch['att_feats'] = batch['att_feats'] * am.unsqueeze(2)
    batch['fc_feats'] = batch['att_feats'].sum(1) / am.sum(1, keepdim=True)
    batch['att_masks'] = am
    return batch


def widen(cg, batch):
    for w in (cg.core.i2h.weight, cg.core.h2h.weight, cg.core.a2c.weight, cg.embed[0].weight,
              cg.core.attention.h2att.weight, cg.core.attention.alpha_net.weight):
        w.data.mul_(3.0)
    cg.logit.weight.data.mul_(6.0)
    B = batch['att_feats'].shape[0]
    sc = (0.3 + 0.5 * (np.arange(B) % 6)).astype(np.float32)
    import torch
    batch['att_feats'] = batch['att_feats'] * torch.from_numpy(sc).view(B, 1, 1)
    if batch.get('att_masks') is not None:
        batch['fc_feats'] = batch['att_feats'].sum(1) / batch['att_masks'].sum(1, keepdim=True)
    else:
        batch['fc_feats'] = batch['att_feats'].mean(1)
    return sc


def sd_np(module, prefix=''):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def flat_noise(prefix, nd):
    return {f'{prefix}.{k}': v for k, v in nd.items()}


_BASES = {}


def base_weights(seed, module, store=True):
    """Store the seeded state dict once; later cases reference it by name.  store=False (full-width case): the weights
    stay in memory, the fixture carries the seed and per-parameter digests instead (golden_util.load_case redraws them)."""
    key = f'weights_s{seed}' if store else f'weights_regen_s{seed}'
    sd = sd_np(module)
    if key not in _BASES:
        _BASES[key] = sd
        if not store:
            return key
        os.makedirs(OUT, exist_ok=True)
        path = os.path.join(OUT, key + '.npz')
        if os.path.exists(path):            # same seed, same draws: leave an identical file alone
            old = np.load(path, allow_pickle=False)
            if sorted(old.files) == sorted(sd) and all(np.array_equal(old[k], sd[k]) for k in sd):
                return key
        np.savez_compressed(path, **sd)
        print('wrote', key, sum(a.nbytes for a in sd.values()) // 1024, 'KiB')
    return key


def weights_of(key, module, strip=''):
    base = _BASES[key]
    cur = {k: v for k, v in sd_np(module).items() if not k.startswith('prev_')}
    if strip:
        base = {k[len(strip):]: v for k, v in base.items() if k.startswith(strip)}
    enc = GU.encode_weights(base, cur)
    enc['weights_ref'] = np.array(key)
    if strip:
        enc['weights_strip'] = np.array(strip)
    return enc


def digests(named_grads):
    return {'gdig.' + k: GU.digest(g.detach().numpy()) for k, g in named_grads}


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    clean = {}
    for k, v in arrs.items():
        if v is None:
            continue
        if hasattr(v, 'detach'):
            v = v.detach().numpy()
        v = np.asarray(v)
        if k.endswith('_keep'):
            v = v.astype(np.uint8)
        clean[k] = v
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **clean)
    print('wrote', name, sum(a.nbytes for a in clean.values()) // 1024, 'KiB')


WIDTH_KEYS = ['input_encoding_size', 'rnn_size', 'fc_feat_size', 'att_feat_size', 'att_hid_size', 'vse_embed_size']


def opt_np(opt, widths=False):
    keys = ['vocab_size', 'seq_length', 'drop_prob_lm', 'gumbel_temp', 'multinomial_temp',
            'prob_gumbel_softmax', 'prob_multinomial_soft', 'decoding_constraint', 'vse_margin',
            'vse_max_violation', 'vse_no_imgnorm', 'vse_use_abs', 'retrieval_reward_weight',
            'cider_optimization', 'caption_loss_weight', 'vse_loss_weight', 'use_gen_cider_scores']
    d = {'cfg.' + k: np.float64(getattr(opt, k)) for k in keys + (WIDTH_KEYS if widths else [])}
    d['cfg.retrieval_reward'] = np.array(opt.retrieval_reward)
    d['cfg.reinforce_baseline_type'] = np.array(opt.reinforce_baseline_type)
    d['cfg.only_one_retrieval'] = np.array(opt.only_one_retrieval)
    d['cfg.vse_pool_type'] = np.array(opt.vse_pool_type)
    return d


# ----------------------------------------------------------------------------
def fc_noise(events, n, B, H, prefix='noise'):
    """The draws of one FCModel pass: one dropout mask per loop iteration (the LSTM output), multinomial picks."""
    keep = np.ones((n, B, H), np.float32)
    pick = np.zeros((n, B), np.int64)
    has_pick, i = False, 0
    for kind, t in events:
        if kind == 'dropout':
            keep[i] = t.numpy()
            i += 1
        elif kind == 'multinomial':
            pick[i] = t.numpy()          # drawn at iteration i (before that iteration's core call)
            has_pick = True
    o = {prefix + '.out_keep': keep}
    if has_pick:
        o[prefix + '.pick'] = pick
    return o


FC_JOINT = (  # the fc-feature speaker under the terms its two-value sample() can drive (reinforce_disc, CIDEr)
    ('fc_joint_reinforce_gt', dict(retrieval_reward='reinforce', reinforce_baseline_type='gt', drop_prob_lm=0.5), 0.8),
    ('fc_joint_reinforce_greedy', dict(retrieval_reward='reinforce', reinforce_baseline_type='greedy',
                                       cider_optimization=0), 0.6),
)


def gen_fc_joint(torch, models, rec):
    for name, kw, bias0 in FC_JOINT:
        opt = make_opt(caption_model='fc', **kw)
        torch.manual_seed(21)
        m = models.AlternatingJointModel(opt)
        m.train()
        cg = m.caption_generator
        for w in (cg.core.i2h.weight, cg.core.h2h.weight, cg.embed.weight, cg.img_embed.weight):
            w.data.mul_(3.0)
        cg.logit.weight.data.mul_(6.0)
        cg.logit.bias.data[0] = bias0
        batch = make_batch(torch, opt, K=7, seed=21)
        marks, tokens = [], []
        orig_sample = cg.sample

        def spy(*a, **k):
            marks.append(len(rec.events))           # where this decode's draws begin
            r = orig_sample(*a, **k)
            tokens.append(r[0].detach().numpy().astype(np.int64).copy())
            return r
        cg.sample = spy
        torch.manual_seed(13)
        rec.start()
        loss = m(batch['fc_feats'], batch['labels'], batch['masks'], {'gts': batch['gts']}, batch['att_feats'], None,
                 is_alternating=True, alternating_turn='speaker')
        ev = rec.stop()
        m.zero_grad()
        loss.backward()
        bounds = marks + [len(ev)]
        nz = {}
        for i in range(len(marks)):
            nz.update(fc_noise(ev[bounds[i]:bounds[i + 1]], opt.seq_length + 2, opt.batch_size, opt.rnn_size, f'noise{i}'))
        grads = digests((k, p.grad) for k, p in m.named_parameters() if not k.startswith('prev_') and p.grad is not None)
        aux = {}
        for k, v in m.loss().items():
            try:
                aux['aux.' + k] = np.float64(float(v))
            except Exception:
                pass
        cfg = opt_np(opt)
        cfg['cfg.caption_model'] = np.array('fc')
        print(name, 'loss', float(loss), 'decodes', [(t.shape[1], sorted(set((t > 0).sum(1).tolist()))) for t in tokens],
              'ngrads', len(grads))
        save(name, **{'w.' + k: v for k, v in sd_np(m).items() if not k.startswith('prev_')}, **cfg, **nz, **grads, **aux,
             **{f'tokens{i}': t for i, t in enumerate(tokens)}, loss=loss, n_decodes=np.int64(len(marks)),
             turn=np.array('speaker'), fc=batch['fc_feats'], att_raw=batch['att_feats'], labels=batch['labels'],
             masks=batch['masks'], gts_flat=np.concatenate(batch['gts'], 0),
             gts_count=np.array([len(x) for x in batch['gts']]))


# ----------------------------------------------------------------------------
# share_embed = 1 (AlternatingJointModel.py:83-88, train.py:390-391, optimizer.py:224-242): ONE embedding table is a parameter
# of both agents and of both Adam instances.  Multi-step cases through the reference's own zeroing_optimizer / update_optimizer:
# what pins the semantics is the weight trajectory (the table is updated by each optimizer that steps, with that optimizer's
# moments, from the one clamped gradient both agents accumulated into).
SHARE_CASES = [
    # gumbel: every iteration is a "speaker" turn that steps BOTH optimizers (optimizer.py:90-95,233-239): the table moves twice
    ('share_joint_gumbel', dict(retrieval_reward='gumbel', drop_prob_lm=0.5), ['speaker', 'speaker']),
    # reinforce: real alternation.  listener turn: the table is frozen (the caption model's requires_grad loop runs last,
    # :592-645) but the listener's Adam still steps it by its momentum; speaker turn: only the speaker's Adam steps it
    ('share_reinforce', dict(retrieval_reward='reinforce', reinforce_baseline_type='gt', vse_loss_weight=1.0, drop_prob_lm=0.5),
     ['listener', 'speaker', 'listener', 'speaker']),
]


def gen_share_embed(torch, models, rec):
    import contextlib
    import io
    import optimizer as ref_optim
    # torch 0.4.1's Optimizer.zero_grad() zero-FILLS the gradients (p.grad.detach_(); p.grad.zero_()); torch 2.x sets them to
    # None by default, and Adam then SKIPS such a parameter.  The difference is visible exactly here: in a reinforce listener
    # turn the shared table is frozen, its gradient stays the zero tensor, and 0.4.1's Adam still moves it by its momentum.
    # Same kind of shim as the zero-dim indexing one: the reference's own call, with the semantics of the torch it was written for.
    _zg = torch.optim.Optimizer.zero_grad
    torch.optim.Optimizer.zero_grad = lambda self, set_to_none=False: _zg(self, set_to_none=False)
    for name, kw, turns in SHARE_CASES:
        opt = make_opt(share_embed=1, learning_rate=2e-3, weight_decay=0.0, grad_clip=0.1, **kw)
        torch.manual_seed(31)
        m = models.AlternatingJointModel(opt)
        assert m.caption_generator.embed[0].weight is m.vse.txt_enc.embed.weight
        m.train()
        cg = m.caption_generator
        batch0 = make_batch(torch, opt, K=7, seed=31)
        widen(cg, batch0)                    # (scales the SHARED table too)
        cg.logit.bias.data[0] = 1.0
        init = {k: v for k, v in sd_np(m).items()}
        opt.is_alternating = 1               # what zeroing_optimizer / update_optimizer branch on
        o_s = ref_optim.define_optimizer(m.caption_generator, opt)
        o_l = ref_optim.define_optimizer(m.vse, opt)
        if opt.retrieval_reward == 'reinforce':
            od = {'speaker': o_s, 'listener': o_l}
        else:
            od = {'speaker': {'speaker': o_s, 'listener': o_l}}      # optimizer.py:90-93
        out = {}
        T = opt.seq_length + 1
        for s_i, turn in enumerate(turns):
            batch = make_batch(torch, opt, K=7, seed=31 + s_i)
            B = opt.batch_size
            sc = (0.3 + 0.5 * (np.arange(B) % 6)).astype(np.float32)
            batch['att_feats'] = batch['att_feats'] * torch.from_numpy(sc).view(B, 1, 1)
            batch['fc_feats'] = batch['att_feats'].mean(1)
            optimizer = od[turn] if turn in od else od['speaker']
            ref_optim.zeroing_optimizer(opt, od, optimizer)
            torch.manual_seed(100 + s_i)
            tokens = []
            orig_sample = cg.sample

            def spy(*a, **k):
                r = orig_sample(*a, **k)
                tokens.append(r[0].detach().numpy().astype(np.int64).copy())
                return r
            cg.sample = spy
            rec.start()
            with contextlib.redirect_stdout(io.StringIO()):
                loss = m(batch['fc_feats'], batch['labels'], batch['masks'], {'gts': batch['gts']}, batch['att_feats'], None,
                         is_alternating=True, alternating_turn=turn)
            ev = rec.stop()
            cg.sample = orig_sample
            loss.backward()
            decs = split_decodes(ev, T, B, opt.input_encoding_size, opt.rnn_size, opt.vocab_size + 1, None)
            pre = f's{s_i}.'
            for i, d in enumerate(decs):
                out.update({pre + k: v for k, v in flat_noise(f'noise{i}', d).items()})
            for i, t in enumerate(tokens):
                out[pre + f'tokens{i}'] = t
            out[pre + 'n_decodes'] = np.int64(len(decs))
            out[pre + 'loss'] = np.float64(float(loss))
            out[pre + 'turn'] = np.array(turn)
            for k, p_ in m.named_parameters():          # (shared table: listed once, under the caption generator's name)
                if k.startswith('prev_') or p_.grad is None:
                    continue
                out[pre + 'gdig.' + k] = GU.digest(p_.grad.detach().numpy())
            out[pre + 'embed_requires_grad'] = np.int64(int(m.vse.txt_enc.embed.weight.requires_grad))
            for k in ('fc_feats', 'att_feats', 'labels', 'masks'):
                out[pre + k] = batch[k].numpy()
            out[pre + 'gts_flat'] = np.concatenate(batch['gts'], 0)
            out[pre + 'gts_count'] = np.array([len(x) for x in batch['gts']])
            with contextlib.redirect_stdout(io.StringIO()):
                ref_optim.update_optimizer(od, optimizer, opt)
            assert cg.embed[0].weight is m.vse.txt_enc.embed.weight
            for k, v in sd_np(m).items():
                if not k.startswith('prev_'):
                    out[pre + 'wdig.' + k] = GU.digest(v)
            print(name, 'step', s_i, turn, 'loss', float(loss), 'decodes', [(t.shape[1], sorted(set((t > 0).sum(1).tolist()))) for t in tokens])
        cfg = opt_np(opt)
        cfg['cfg.share_embed'] = np.float64(1)
        cfg['cfg.learning_rate'] = np.float64(opt.learning_rate)
        cfg['cfg.grad_clip'] = np.float64(opt.grad_clip)
        save(name, **{'w.' + k: v for k, v in init.items()}, **cfg, **out, n_steps=np.int64(len(turns)))
    torch.optim.Optimizer.zero_grad = _zg


def gen_fc(torch, models, rec):
    """FCModel (the fc-feature speaker of BASELINE configs[0]): MLE forward/backward and greedy / multinomial decodes."""

    for name, pdrop, seed in (('fc_mle', 0.0, 11), ('fc_mle_dropout', 0.5, 12)):
        opt = make_opt(caption_model='fc', drop_prob_lm=pdrop)
        torch.manual_seed(seed)
        fm = models.setup(opt, 'fc', 'caption_model')
        fm.train()
        for w in (fm.core.i2h.weight, fm.core.h2h.weight, fm.embed.weight, fm.img_embed.weight):
            w.data.mul_(3.0)
        fm.logit.weight.data.mul_(6.0)
        fm.logit.bias.data.uniform_(-0.2, 0.2)
        batch = make_batch(torch, opt, K=7, seed=seed)
        rec.start()
        loss = fm(batch['fc_feats'], None, None, batch['labels'], batch['masks'])
        ev = rec.stop()
        loss.backward()
        T2 = batch['labels'].shape[1]
        save(name, **{'w.' + k: v for k, v in sd_np(fm).items()}, **opt_np(opt), fc=batch['fc_feats'],
             labels=batch['labels'], masks=batch['masks'], loss=loss.detach().reshape(1),
             **fc_noise(ev, T2, opt.batch_size, opt.rnn_size), **digests((k, p.grad) for k, p in fm.named_parameters()))

    for name, smax, temp, pdrop, bias0, seed in (('fc_sample_greedy', 1, 1.0, 0.0, 0.6, 13), ('fc_sample_greedy_dropout', 1, 1.0, 0.5, 0.4, 14),
                                                 ('fc_sample_multinomial', 0, 1.0, 0.5, 0.8, 15), ('fc_sample_multinomial_temp', 0, 0.7, 0.0, 1.2, 16)):
        opt = make_opt(caption_model='fc', drop_prob_lm=pdrop)
        torch.manual_seed(seed)
        fm = models.setup(opt, 'fc', 'caption_model')
        fm.train()
        for w in (fm.core.i2h.weight, fm.core.h2h.weight, fm.embed.weight, fm.img_embed.weight):
            w.data.mul_(3.0)
        fm.logit.weight.data.mul_(6.0)
        fm.logit.bias.data[0] = bias0
        batch = make_batch(torch, opt, K=7, seed=seed)
        rec.start()
        with torch.no_grad():
            seq, slp = fm.sample(batch['fc_feats'], None, None, {'sample_max': smax, 'temperature': temp})
        ev = rec.stop()
        print(name, 'L =', seq.shape[1], 'lens', (seq > 0).sum(1).tolist())
        save(name, **{'w.' + k: v for k, v in sd_np(fm).items()}, **opt_np(opt), fc=batch['fc_feats'], res0=seq, res1=slp,
             **{'opt.sample_max': np.float64(smax), 'opt.temperature': np.float64(temp)},
             **fc_noise(ev, opt.seq_length + 2, opt.batch_size, opt.rnn_size))


BEAM_CASES = (('beam2', 2, 0, 2.0, 22, False), ('beam3_early', 3, 0, 1.8, 22, False),
              ('beam5_constraint', 5, 1, 1.86, 22, False))
BEAM_CASES_MASKED = (('masked_beam3', 3, 0, 1.8, 22, True),)


def gen_beam(torch, models, rec, cases=BEAM_CASES):
    """AttModel.sample_beam (evaluation decode, AttModel.py:150-289): beam 2 (eval.py's setting), 3 and 5 with the
    decoding constraint; `masked`: ragged region counts + att_masks."""
    for name, beam, dc, bias0, seed, masked in cases:
        opt = make_opt(batch_size=5, decoding_constraint=dc)
        torch.manual_seed(seed)
        m = models.AlternatingJointModel(opt)
        cg = m.caption_generator
        batch = make_batch(torch, opt, K=7, seed=seed)
        if masked:
            mask_batch(torch, batch)
        widen(cg, batch)
        cg.logit.bias.data[0] = bias0
        cg.eval()
        with torch.no_grad():
            seq, lps = cg.sample(batch['fc_feats'], batch['att_feats'], batch['att_masks'],
                                 {'beam_size': beam, 'decoding_constraint': dc})
        score = np.array([float(cg.done_beams[k][0]['p']) for k in range(opt.batch_size)], np.float32)
        nd = np.array([len(cg.done_beams[k]) for k in range(opt.batch_size)])
        print(name, 'lens', (seq > 0).sum(1).tolist(), 'done beams per image', nd.tolist())
        save(name, **{'w.' + k: v for k, v in sd_np(cg).items()}, **opt_np(opt), att_raw=batch['att_feats'], fc=batch['fc_feats'],
             res0=seq, res1=lps, score=score, beam=np.int64(beam), att_masks=batch['att_masks'])


def gen_retrieval(torch):
    """eval_utils.i2t / t2i (retrieval-rank evaluation of the listener, eval_utils.py:545-720) on random unit-norm
    embeddings with a planted image-caption correlation."""
    import eval_utils
    for name, N, K, cpi, seed in (('retrieval_5cap', 60, 48, 5, 41), ('retrieval_gen_1cap', 80, 32, 1, 42)):
        rs = np.random.RandomState(seed)
        im = rs.randn(N, K).astype(np.float32)
        im /= np.linalg.norm(im, axis=1, keepdims=True)
        cap = (0.8 * np.repeat(im, cpi, 0) + rs.randn(N * cpi, K).astype(np.float32))
        cap /= np.linalg.norm(cap, axis=1, keepdims=True)
        images = np.repeat(im, cpi, 0)
        data = [{'id': i, 'file_path': str(i)} for i in range(N)]
        out = {}
        if cpi == 5:
            r, (ranks, top1) = eval_utils.i2t(images, cap, measure='cosine', return_ranks=True)
            out.update(i2t_r=np.array(r), i2t_ranks=ranks, i2t_top1=top1)
        ri, (ranks_i, top1_i), _ = eval_utils.t2i(images, cap, data, measure='cosine', return_ranks=True, useGenSent=(cpi == 1))
        out.update(t2i_r=np.array(ri), t2i_ranks=ranks_i, t2i_top1=top1_i)
        save(name, images=images, captions=cap.astype(np.float32), cpi=np.int64(cpi), **out)


def gen_state_dict_layout(torch, models):
    """Checkpoint compatibility: the KEYS and SHAPES of the reference's AlternatingJointModel.state_dict()
    - what train.py's save_model writes into alternatingModel.pth (train.py:119-129) - at BASELINE's widths: a fresh model,
    and the model after a REINFORCE speaker turn, when changeModelUpdateStatus (AlternatingJointModel.py:571-586) has
    deep-copied both agents into the sub-modules prev_vse / prev_caption_generator, whose tensors the reference then saves
    as well.  Names and shapes only (plus a digest of the seed-0 initial values per tensor): no weights are stored."""
    import json as _json
    opt = make_opt(vocab_size=9487, input_encoding_size=512, rnn_size=512, att_hid_size=512, fc_feat_size=2048,
                   att_feat_size=2048, vse_embed_size=1024, retrieval_reward='reinforce', batch_size=4)
    torch.manual_seed(0)
    m = models.AlternatingJointModel(opt)
    fresh = {k: [int(d) for d in v.shape] for k, v in m.state_dict().items()}
    dig = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in m.state_dict().items()}
    m.changeModelUpdateStatus({'vseModel': False, 'captionModel': True})          # what a reinforce speaker turn does first (:508-511)
    after = {k: [int(d) for d in v.shape] for k, v in m.state_dict().items()}
    assert set(fresh) < set(after)
    np.savez_compressed(os.path.join(OUT, 'state_dict_layout.npz'),
                        fresh=np.array(_json.dumps(fresh)), after_reinforce_speaker_turn=np.array(_json.dumps(after)),
                        init_digest_seed0=np.array(_json.dumps(dig)), opt=np.array(_json.dumps(
                            {k: v for k, v in vars(opt).items() if isinstance(v, (int, float, str, list, type(None)))})))
    print('state_dict_layout:', len(fresh), 'keys fresh,', len(after), 'after a reinforce speaker turn')


def main():
    torch, models, rewards = install_harness()
    rec = Recorder(torch)
    torch.set_num_threads(4)
    if '--only-statedict' in sys.argv:
        gen_state_dict_layout(torch, models)
        return
    only_masks = '--only-masks' in sys.argv or '--only-bn' in sys.argv     # the att_masks cases only (other fixtures untouched); --only-bn: the two use_bn cases only
    only_full = '--only-fullwidth' in sys.argv  # the BASELINE-width joint step only
    if '--only-retrieval' in sys.argv:
        gen_retrieval(torch)
        return
    if '--only-fc' in sys.argv:
        gen_fc(torch, models, rec)
        return
    if '--only-fc-joint' in sys.argv:
        rewards.init_scorer('corpus')
        gen_fc_joint(torch, models, rec)
        return
    if '--only-share-embed' in sys.argv:
        rewards.init_scorer('corpus')
        gen_share_embed(torch, models, rec)
        return
    if '--only-beam' in sys.argv:
        gen_beam(torch, models, rec)
        return

    def build(opt, seed=0, eos_bias=None, store=True):
        torch.manual_seed(seed)
        m = models.AlternatingJointModel(opt)
        m._wkey = base_weights(seed, m, store)
        if eos_bias is not None:
            m.caption_generator.logit.bias.data[0] = eos_bias
        return m
    rewards.init_scorer('corpus')

    # ------------------------------------------------------------------ S3 / S4 / S5 kernels
    def kernel_cases():
        opt = make_opt()
        m = build(opt, 1)
        cg = m.caption_generator
        batch = make_batch(torch, opt, K=7, seed=1)
        B, H = opt.batch_size, opt.rnn_size
        g = torch.Generator().manual_seed(7)
        h = torch.randn(B, H, generator=g) * 0.5
        c = torch.randn(B, H, generator=g) * 0.5
        att = cg.att_embed(batch['att_feats'])
        p_att = cg.ctx2att(att)
        att_res = cg.core.attention(h, att, p_att, None)
        am = (torch.rand(B, 7, generator=g) > 0.3).float()
        am[:, 0] = 1
        att_res_m = cg.core.attention(h, att, p_att, am)
        xt = torch.randn(B, opt.input_encoding_size, generator=g)
        out, st = cg.core(xt, None, att, p_att, None, (h.unsqueeze(0), c.unsqueeze(0)))
        logp = torch.nn.functional.log_softmax(cg.logit(out), dim=1)
        save('kernels_speaker', **weights_of(m._wkey, cg, 'caption_generator.'), **opt_np(opt), att_raw=batch['att_feats'], h=h, c=c, xt=xt,
             att=att, p_att=p_att, att_res=att_res, att_masks=am, att_res_masked=att_res_m,
             out=out, h2=st[0][0], c2=st[1][0], logp=logp)

    # ------------------------------------------------------------------ S1 decodes
    def sample_case(name, rr, kw, opts_, eos, masked=False):
        opt = make_opt(retrieval_reward=rr, **kw)
        m = build(opt, 2, None if eos == 'search' else eos)
        cg = m.caption_generator
        cg.train()
        batch = make_batch(torch, opt, K=7, seed=2)
        if masked:
            mask_batch(torch, batch)
        am = batch['att_masks']
        if eos == 'search':
            # greedy decodes finish all-at-once for a large EOS bias and never for a small
            # one: widen the logits, then scan for mixed lengths
            cg.logit.weight.data.mul_(4.0)
            found = None
            for bias in np.linspace(0.0, 3.0, 61):
                cg.logit.bias.data[0] = float(bias)
                torch.manual_seed(11)
                try:
                    r = cg.sample(batch['fc_feats'], batch['att_feats'], am, dict(opts_))
                except ValueError:
                    break
                lens = (r[0] > 0).sum(1)
                if 2 <= r[0].shape[1] < opt.seq_length and len(set(lens.tolist())) >= 3:
                    found = float(bias)
            assert found is not None, name
            cg.logit.bias.data[0] = found
        torch.manual_seed(11)
        rec.start()
        res = cg.sample(batch['fc_feats'], batch['att_feats'], am, dict(opts_))
        ev = rec.stop()
        T = opt.seq_length + 1
        nd = split_decodes(ev, T, opt.batch_size, opt.input_encoding_size, opt.rnn_size, opt.vocab_size + 1, am)
        nz = flat_noise('noise', nd[0]) if nd else {}
        outs = {f'res{i}': r for i, r in enumerate(res)}
        print(name, 'L =', res[0].shape[1])
        save(name, **weights_of(m._wkey, cg, 'caption_generator.'), **opt_np(opt), **nz, **outs, att_raw=batch['att_feats'],
             fc=batch['fc_feats'], att_masks=am, **{'opt.' + k: np.float64(v) for k, v in opts_.items()})

    SAMPLE_MASKED = [
        ('masked_sample_greedy', 'gumbel', {'drop_prob_lm': 0.5}, {'sample_max': 1}, 'search'),
        ('masked_sample_gumbel_st', 'gumbel', {'drop_prob_lm': 0.5}, {'sample_max': 0, 'temperature': 1, 'use_one_hot': 1}, 2.5),
    ]
    for name, rr, kw, opts_, eos in [
        ('sample_greedy_full', 'gumbel', {}, {'sample_max': 1}, None),
        ('sample_greedy_early', 'gumbel', {}, {'sample_max': 1}, 'search'),
        ('sample_greedy_dropout', 'gumbel', {'drop_prob_lm': 0.5}, {'sample_max': 1}, 'search'),
        ('sample_multinomial_plain', 'reinforce', {'drop_prob_lm': 0.5}, {'sample_max': 0, 'temperature': 1}, 2.5),
        ('sample_multinomial_temp', 'reinforce', {}, {'sample_max': 0, 'temperature': 0.7}, 2.5),
        ('sample_gumbel_st', 'gumbel', {'drop_prob_lm': 0.5}, {'sample_max': 0, 'temperature': 1, 'use_one_hot': 1}, 2.5),
        ('sample_gumbel_st_tau', 'gumbel', {'gumbel_temp': 0.5}, {'sample_max': 0, 'temperature': 1, 'use_one_hot': 1}, 3.0),
        ('sample_multinomial_st', 'multinomial', {'drop_prob_lm': 0.5}, {'sample_max': 0, 'temperature': 1, 'use_one_hot': 1}, 2.5),
        ('sample_gumbel_ps', 'gumbel_softmax', {'drop_prob_lm': 0.5}, {'sample_max': 0, 'temperature': 1, 'use_one_hot': 1}, 2.5),
        ('sample_multinomial_ps', 'multinomial_soft', {'multinomial_temp': 1.0}, {'sample_max': 0, 'temperature': 1, 'use_one_hot': 1}, 2.5),
        ('sample_multinomial_ps_tau', 'multinomial_soft', {'multinomial_temp': 2.0}, {'sample_max': 0, 'temperature': 1, 'use_one_hot': 1}, 2.5),
        ('sample_constraint', 'reinforce', {'decoding_constraint': 1}, {'sample_max': 1}, None),
    ]:
        if not only_masks:
            sample_case(name, rr, kw, opts_, eos)

    # ------------------------------------------------------------------ S2 MLE
    def mle_case(name, kw, ss, masked=False, seed=3):
        # (seed: a model with other state-dict keys - use_bn = 1 - takes a base-weight file of its own)
        opt = make_opt(**kw)
        m = build(opt, seed)
        cg = m.caption_generator
        cg.train()
        cg.ss_prob = ss
        batch = make_batch(torch, opt, K=7, seed=3)
        if masked:
            mask_batch(torch, batch)
        am = batch['att_masks']
        torch.manual_seed(12)
        rec.start()
        loss = cg(batch['fc_feats'], batch['att_feats'], am, batch['labels'], batch['masks'])
        ev = rec.stop()
        loss.backward()
        T = opt.seq_length + 1
        # scheduled sampling draws: uniform_ [B] then (maybe) multinomial per step i>=1
        nd = split_decodes(ev, T, opt.batch_size, opt.input_encoding_size, opt.rnn_size, opt.vocab_size + 1, am)[0]
        if 'ps_u' in nd:
            nd['ss_u'] = nd.pop('ps_u')
        grads = digests((k, p.grad) for k, p in cg.named_parameters() if p.grad is not None)
        # use_bn = 1: the running statistics AFTER this one training-mode forward (momentum 0.1, unbiased variance)
        after = {'after.' + k: v.detach().numpy().copy() for k, v in cg.state_dict().items() if 'running_' in k or 'num_batches' in k}
        cfg = opt_np(opt)
        if getattr(opt, 'use_bn', 0):
            cfg['cfg.use_bn'] = np.float64(1)
        save(name, **weights_of(m._wkey, cg, 'caption_generator.'), **cfg, **flat_noise('noise', nd), **grads, **after, loss=loss,
             ss_prob=np.float64(ss), att_raw=batch['att_feats'], fc=batch['fc_feats'], att_masks=am,
             labels=batch['labels'], masks=batch['masks'])

    for name, kw, ss in [('mle_plain', {}, 0.0), ('mle_dropout', {'drop_prob_lm': 0.5}, 0.0),
                         ('mle_ss', {'drop_prob_lm': 0.5}, 0.25)]:
        if not only_masks:
            mle_case(name, kw, ss)

    # ------------------------------------------------------------------ V1-V4 listener
    def listener_cases():
        opt = make_opt()
        m = build(opt, 4)
        vse = m.vse
        batch = make_batch(torch, opt, K=7, seed=4)
        B, V = opt.batch_size, opt.vocab_size
        img = vse.img_enc(batch['fc_feats'])
        cap = vse.txt_enc(batch['labels'], batch['masks'])
        onehot = torch.zeros(B, batch['labels'].shape[1], V + 2)
        onehot.scatter_(2, batch['labels'].unsqueeze(2), 1.0)
        g = torch.Generator().manual_seed(5)
        soft = torch.softmax(torch.randn(B, batch['labels'].shape[1], V + 2, generator=g), 2)
        soft.requires_grad_(True)
        cap_oh = vse.txt_enc(onehot, batch['masks'])
        res = {}
        for wb in (False, True):
            for oor in ('off', 'image', 'caption'):
                res[f'loss_wb{int(wb)}_{oor}'] = vse(batch['fc_feats'], None, batch['labels'], batch['masks'], wb, oor)
        vse.zero_grad()
        loss_soft = vse(batch['fc_feats'], None, soft, batch['masks'])
        loss_soft.backward()
        grads = digests((k, p.grad) for k, p in vse.named_parameters())
        save('listener', **weights_of(m._wkey, vse, 'vse.'), **opt_np(opt), fc=batch['fc_feats'], labels=batch['labels'],
             masks=batch['masks'], img_emb=img, cap_emb=cap, cap_emb_onehot=cap_oh, soft=soft,
             loss_soft=loss_soft, grad_soft=soft.grad, **grads, **res)
        for name, kw in [('listener_mean', dict(vse_pool_type='mean', vse_max_violation=0)),
                         ('listener_max', dict(vse_pool_type='max', vse_use_abs=1, vse_no_imgnorm=1))]:
            opt2 = make_opt(**kw)
            m2 = build(opt2, 4)
            l = m2.vse(batch['fc_feats'], None, batch['labels'], batch['masks'])
            save(name, **weights_of(m2._wkey, m2.vse, 'vse.'), **opt_np(opt2), fc=batch['fc_feats'], labels=batch['labels'],
                 masks=batch['masks'], loss=l)

    # ------------------------------------------------------------------ R1-R4 CIDEr-D
    def cider_cases():
        rewards.init_scorer('corpus')
        rs = np.random.RandomState(9)
        B, L = 12, 16
        V = 23

        def rnd_rows(n, Lr, p0):
            a = rs.randint(1, V, size=(n, Lr))
            for i in range(n):
                if rs.rand() < p0:
                    a[i, rs.randint(0, Lr):] = 0
            return a
        gen = rnd_rows(B, L, 0.7)
        gen[0, :] = 0            # EOS first
        gen[1] = 3               # repeated token, no EOS
        gen[2, :5] = [4, 5, 4, 5, 4]
        greedy = rnd_rows(B, 11, 0.7)
        gts = [rnd_rows(rs.randint(1, 6), 16, 0.9) for _ in range(B)]
        gts[3][0] = gen[3]       # exact match
        sc, cg_ = rewards.get_self_critical_reward({'gts': gts}, torch.from_numpy(gen), torch.from_numpy(greedy))
        cgen, sc2, cg2 = rewards.get_self_critical_reward({'gts': gts}, torch.from_numpy(gen),
                                                          torch.from_numpy(greedy), True)
        ngts = np.array([len(x) for x in gts])
        gts_flat = np.concatenate(gts, 0)
        save('ciderd', gen=gen, greedy=greedy, gts_flat=gts_flat, gts_count=ngts, reward=sc,
             cider_greedy=np.float64(cg_), cider_gen=cgen)
        # seq_per_img = 2 variant
        gts2 = gts[:6]
        sc3, cg3 = rewards.get_self_critical_reward({'gts': gts2}, torch.from_numpy(gen), torch.from_numpy(greedy))
        save('ciderd_spi2', gen=gen, greedy=greedy, gts_flat=np.concatenate(gts2, 0),
             gts_count=np.array([len(x) for x in gts2]), reward=sc3, cider_greedy=np.float64(cg3))

    # ------------------------------------------------------------------ A1 full joint steps
    cases = [
        ('joint_gumbel', dict(retrieval_reward='gumbel'), 'speaker', 2.5),
        ('joint_gumbel_dropout', dict(retrieval_reward='gumbel', drop_prob_lm=0.5), 'speaker', 2.5),
        ('joint_gumbel_tau', dict(retrieval_reward='gumbel', gumbel_temp=0.5, only_one_retrieval='image'), 'speaker', 2.5),
        ('joint_multinomial', dict(retrieval_reward='multinomial', drop_prob_lm=0.5), 'speaker', 2.5),
        ('joint_gumbel_ps', dict(retrieval_reward='gumbel_softmax', drop_prob_lm=0.5), 'speaker', 2.5),
        ('joint_multinomial_ps', dict(retrieval_reward='multinomial_soft'), 'speaker', 2.5),
        ('joint_reinforce_gt', dict(retrieval_reward='reinforce', reinforce_baseline_type='gt', drop_prob_lm=0.5), 'speaker', 2.5),
        ('joint_reinforce_greedy', dict(retrieval_reward='reinforce', reinforce_baseline_type='greedy'), 'speaker', 2.5),
        ('joint_reinforce_no', dict(retrieval_reward='reinforce', reinforce_baseline_type='no', cider_optimization=0), 'speaker', 2.5),
        ('joint_reinforce_listener', dict(retrieval_reward='reinforce', vse_loss_weight=1.0, drop_prob_lm=0.5), 'listener', 2.5),
        ('joint_gumbel_mle', dict(retrieval_reward='gumbel', caption_loss_weight=0.5, use_gen_cider_scores=1), 'speaker', 2.5),
        ('joint_plain_all', dict(retrieval_reward='gumbel', caption_loss_weight=1.0, vse_loss_weight=1.0), None, 2.5),
    ]
    U_SEED = 4242

    def joint_case(name, kw, turn, eos, masked=False, regen=False, K=7, seed=5):
        # regen (the full-width case): nothing large is stored.  Weights are the seeded draw (seed + digests), features
        # a seeded draw, the Gumbel uniforms come from a generator of their own (Recorder.inject); golden_util.load_case
        # redraws all three and checks the digests stored here.
        opt = make_opt(**kw)
        # (seed: a model with other state-dict keys - use_bn = 1 - takes a base-weight file of its own)
        m = build(opt, seed, None, store=not regen)
        m.train()
        batch = make_batch(torch, opt, K=K, seed=5)
        if masked:
            mask_batch(torch, batch)
        am = batch['att_masks']
        cg = m.caption_generator
        # random-init greedy decodes are knife-edge (never EOS / all EOS at t=1, SURVEY.md
        # Appendix A.16): widen the dynamics so the state, and with it EOS, varies per row
        rowscale = widen(cg, batch)
        shapes = []
        orig_sample = cg.sample

        tokens = []

        def spy(*a, **k):
            r = orig_sample(*a, **k)
            shapes.append((r[0].shape[1], sorted(set((r[0] > 0).sum(1).tolist()))))
            tokens.append(r[0].detach().numpy().astype(np.int64).copy())
            return r
        cg.sample = spy

        def run():
            if turn is None:
                return m(batch['fc_feats'], batch['labels'], batch['masks'], {'gts': batch['gts']},
                         batch['att_feats'], am)
            return m(batch['fc_feats'], batch['labels'], batch['masks'], {'gts': batch['gts']},
                     batch['att_feats'], am, is_alternating=True, alternating_turn=turn)
        # scan the EOS bias for a run whose decodes have mixed lengths and (if any
        # greedy decode happens) at least one decode that stops early (L < 16)
        found = None
        import contextlib, io
        for bias in (np.linspace(3.0, 13.0, 41) if regen else np.linspace(-1.0, 3.0, 81)):
            cg.logit.bias.data[0] = float(bias)
            torch.manual_seed(13)
            rec.inject = torch.Generator().manual_seed(U_SEED) if regen else None
            del shapes[:]
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    run()
            except ValueError:
                continue
            ok = all(len(sh[1]) >= 2 for sh in shapes) and any(len(sh[1]) >= 3 for sh in shapes)
            early = any(sh[0] < opt.seq_length for sh in shapes)
            if not shapes:                      # a step without sampled decodes (MLE only): nothing to scan for
                ok = early = True
            if ok and (early or found is None):
                found = float(bias)
                if early:
                    break
        assert found is not None, name
        cg.logit.bias.data[0] = found
        if hasattr(m, 'prev_vse'):      # drop the reinforce bookkeeping copies made by the scan
            del m.prev_vse, m.prev_caption_generator, m.prev_gradDic
        del shapes[:]
        del tokens[:]
        # use_bn = 1: the scan above has ticked the running statistics; the recorded step starts from a fresh BatchNorm1d
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.reset_running_stats()
        torch.manual_seed(13)
        rec.inject = torch.Generator().manual_seed(U_SEED) if regen else None
        rec.start()
        loss = run()
        print(name, 'eos bias', found, 'decodes (L, lens):', shapes)
        ev = rec.stop()
        rec.inject = None
        m.zero_grad()
        loss.backward()
        T = opt.seq_length + 1
        decs = split_decodes(ev, T, opt.batch_size, opt.input_encoding_size, opt.rnn_size, opt.vocab_size + 1, am,
                             want_steps=regen)
        nz = {}
        calls = 0
        for i, d in enumerate(decs):
            if regen and 'gumbel_u' in d:       # [seed, calls made before this decode, calls of this decode] + digest
                u = d.pop('gumbel_u')
                steps = d['gumbel_u_steps']
                d['gumbel_u_regen'] = np.array([U_SEED, calls, len(steps)], np.int64)
                d['gumbel_u_digest'] = GU.digest(u)
                calls += len(steps)
            nz.update(flat_noise(f'noise{i}', d))
        grads = digests((k, p.grad) for k, p in m.named_parameters()
                        if not k.startswith('prev_') and p.grad is not None)
        sd = weights_of(m._wkey, m)
        aux = {}
        for k, v in m.loss().items():
            try:
                aux['aux.' + k] = np.float64(float(v))
            except Exception:
                pass
        print(name, 'loss', float(loss), 'ndecodes', len(decs), 'ngrads', len(grads))
        # use_bn = 1: the running statistics AFTER the step (one update per decode that ran att_embed), and the flag
        after = {'after.' + k: v.detach().numpy().copy() for k, v in m.state_dict().items()
                 if not k.startswith('prev_') and ('running_' in k or 'num_batches' in k)}
        if getattr(opt, 'use_bn', 0):
            after['cfg.use_bn'] = np.float64(1)
        feats = dict(fc=batch['fc_feats'], att_raw=batch['att_feats'], **after)
        if regen:
            B_, K_, D_ = batch['att_feats'].shape
            feats = {'regen.att': np.array([5, B_, K_, D_], np.int64), 'regen.att_rowscale': rowscale,
                     'dig.att_raw': GU.digest(batch['att_feats'].numpy()), 'dig.fc': GU.digest(batch['fc_feats'].numpy())}
            feats.update({'wdig.' + k: GU.digest(v) for k, v in _BASES[m._wkey].items()})
            feats.update({f'tokens{i}': t for i, t in enumerate(tokens)})
        save(name, **sd, **opt_np(opt, widths=regen), **nz, **grads, **aux, **feats, loss=loss, n_decodes=np.int64(len(decs)),
             turn=np.array(str(turn)), att_masks=am, labels=batch['labels'], masks=batch['masks'],
             gts_flat=np.concatenate(batch['gts'], 0), gts_count=np.array([len(x) for x in batch['gts']]))

    # BASELINE.json's widths (configs[2]: 36 x 2048 regions, hidden 512, vocabulary 9487, listener 1024), 32 images
    FULLWIDTH = dict(retrieval_reward='gumbel', drop_prob_lm=0.5, vocab_size=9487, input_encoding_size=512, rnn_size=512,
                     fc_feat_size=2048, att_feat_size=2048, att_hid_size=512, vse_embed_size=1024, batch_size=32)
    # ... and one step with every term on (MLE + VSE on the labels + ST-Gumbel + CIDEr-D, both agents), not alternating
    FULLWIDTH_ALL = dict(FULLWIDTH, caption_loss_weight=1.0, vse_loss_weight=1.0)
    # ... and the listener's turn of the REINFORCE configuration (BASELINE configs[3]): multinomial captions -> VSE loss
    FULLWIDTH_LST = dict(FULLWIDTH, retrieval_reward='reinforce', vse_loss_weight=1.0)
    # ... and its speaker's turn: REINFORCE with the ground-truth baseline + self-critical CIDEr-D
    FULLWIDTH_RF = dict(FULLWIDTH, retrieval_reward='reinforce', reinforce_baseline_type='gt')
    # ... and BASELINE configs[2] itself: the headline step at its own batch size
    FULLSIZE = dict(FULLWIDTH, batch_size=128)
    # ... and BASELINE configs[1]: the MLE step at B = 64
    # ... and BASELINE configs[3]: the speaker's turn of the REINFORCE configuration at B = 256
    FULLSIZE_RF = dict(FULLWIDTH_RF, batch_size=256)
    FULLSIZE_MLE = dict(FULLWIDTH, batch_size=64, caption_loss_weight=1.0, retrieval_reward_weight=0.0, cider_optimization=0)
    if only_full:
        joint_case('fullwidth_joint_gumbel', FULLWIDTH, 'speaker', 2.5, regen=True, K=36)
        joint_case('fullwidth_plain_all', FULLWIDTH_ALL, None, 2.5, regen=True, K=36)
        joint_case('fullwidth_reinforce_listener', FULLWIDTH_LST, 'listener', 2.5, regen=True, K=36)
        joint_case('fullwidth_reinforce_speaker', FULLWIDTH_RF, 'speaker', 2.5, regen=True, K=36)
        joint_case('fullsize_joint_gumbel', FULLSIZE, 'speaker', 2.5, regen=True, K=36)
        joint_case('fullsize_mle', FULLSIZE_MLE, None, 2.5, regen=True, K=36)
        joint_case('fullsize_reinforce_speaker', FULLSIZE_RF, 'speaker', 2.5, regen=True, K=36)
        return
    for name, kw, turn, eos in cases:
        if not only_masks:
            joint_case(name, kw, turn, eos)
    if not only_masks:
        joint_case('fullwidth_joint_gumbel', FULLWIDTH, 'speaker', 2.5, regen=True, K=36)
        joint_case('fullwidth_plain_all', FULLWIDTH_ALL, None, 2.5, regen=True, K=36)
        joint_case('fullwidth_reinforce_listener', FULLWIDTH_LST, 'listener', 2.5, regen=True, K=36)
        joint_case('fullwidth_reinforce_speaker', FULLWIDTH_RF, 'speaker', 2.5, regen=True, K=36)
        joint_case('fullsize_joint_gumbel', FULLSIZE, 'speaker', 2.5, regen=True, K=36)
        joint_case('fullsize_mle', FULLSIZE_MLE, None, 2.5, regen=True, K=36)
        joint_case('fullsize_reinforce_speaker', FULLSIZE_RF, 'speaker', 2.5, regen=True, K=36)

    # ------------------------------------------------------------------ O1 clamp + Adam
    def clamp_adam_case():
        import misc.utils as rutils
        torch.manual_seed(3)
        p0 = torch.randn(37, 5)
        p = torch.nn.Parameter(p0.clone())
        optim = torch.optim.Adam([p], lr=5e-4, weight_decay=0)
        traj = []
        gs = []
        for it in range(3):
            gr = torch.randn(37, 5) * (0.3 if it != 1 else 0.01)
            gs.append(gr.clone())
            optim.zero_grad()
            p.grad = gr.clone()
            rutils.clip_gradient(optim, 0.1)
            optim.step()
            traj.append(p.detach().clone())
        save('clamp_adam', p0=p0, grads=torch.stack(gs), traj=torch.stack(traj), lr=np.float64(5e-4),
             grad_clip=np.float64(0.1))
    JOINT_MASKED = [('masked_joint_gumbel', dict(retrieval_reward='gumbel', drop_prob_lm=0.5), 'speaker', 2.5)]
    if '--only-bn' in sys.argv:      # use_bn = 1 on ragged region counts (the only input the option can run on, AttModel.py:44-51,82-85)
        mle_case('bn_masked_mle', {'drop_prob_lm': 0.5, 'use_bn': 1}, 0.0, masked=True, seed=33)
        joint_case('bn_masked_joint_gumbel', dict(retrieval_reward='gumbel', drop_prob_lm=0.5, use_bn=1), 'speaker', 2.5, masked=True, seed=55)
        return
    if not only_masks:
        kernel_cases()
        listener_cases()
        cider_cases()
        clamp_adam_case()
        gen_fc(torch, models, rec)
        gen_fc_joint(torch, models, rec)
        gen_share_embed(torch, models, rec)
        gen_beam(torch, models, rec)
        gen_retrieval(torch)
        gen_state_dict_layout(torch, models)
    # ------------------------------------------------------------------ att_masks (ragged region counts)
    for name, rr, kw, opts_, eos in SAMPLE_MASKED:
        sample_case(name, rr, kw, opts_, eos, masked=True)
    mle_case('masked_mle', {'drop_prob_lm': 0.5}, 0.0, masked=True)
    mle_case('bn_masked_mle', {'drop_prob_lm': 0.5, 'use_bn': 1}, 0.0, masked=True, seed=33)
    joint_case('bn_masked_joint_gumbel', dict(retrieval_reward='gumbel', drop_prob_lm=0.5, use_bn=1), 'speaker', 2.5, masked=True, seed=55)
    for name, kw, turn, eos in JOINT_MASKED:
        joint_case(name, kw, turn, eos, masked=True)
    gen_beam(torch, models, rec, BEAM_CASES_MASKED)


if __name__ == '__main__':
    main()
