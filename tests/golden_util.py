"""Helpers shared by tools/gen_golden.py (writer) and the tests (reader).

Fixtures are kept small: weights are stored once per seed (``weights_s<seed>.npz``)
and each case stores only a per-parameter scalar multiplier (or the full array
where a parameter is not a scalar multiple of the seeded one); gradients are
stored as a digest (sum, abs-sum and 256 fixed sample entries per parameter).
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NSAMP = 256


def sample_index(n):
    rs = np.random.RandomState(12345)
    if n <= NSAMP:
        return np.arange(n)
    return np.sort(rs.choice(n, NSAMP, replace=False))


def digest(arr):
    """-> f64[2 + nsamp]: sum, abs-sum, samples."""
    a = np.asarray(arr, dtype=np.float64).reshape(-1)
    return np.concatenate([[a.sum(), np.abs(a).sum()], a[sample_index(a.size)]])


def encode_weights(base, cur):
    """Per-parameter: scalar c with cur == base * f32(c), else the full array."""
    out = {}
    for k, v in cur.items():
        b = base.get(k)
        if b is not None and b.shape == v.shape:
            if np.array_equal(b, v):
                continue
            nz = np.flatnonzero(b)
            if nz.size:
                c = np.float32(v.reshape(-1)[nz[0]] / b.reshape(-1)[nz[0]])
                hit = [x for x in (c, np.float32(np.round(c))) if np.array_equal(b * x, v)]
                if hit:
                    out['wscale.' + k] = hit[0]
                    continue
        out['w.' + k] = v
    return out


WIDTH_KEYS = ('input_encoding_size', 'rnn_size', 'fc_feat_size', 'att_feat_size', 'att_hid_size', 'vse_embed_size')


def _check_digest(arr, want, what):
    got = digest(arr)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0, err_msg=f'{what}: the redrawn array is not the recorded one')


def _redraw(z):
    """The full-width case stores no large array (tools/gen_golden.py, joint_case(regen=True)): the seeded weights,
    the seeded region features and the Gumbel uniforms are drawn again here, exactly as the generator drew them, and
    checked against the digests recorded from the reference run (sum, abs-sum, 256 samples per array)."""
    import torch
    from cooperativeimagecaptioning_amd import models
    seed = int(str(z['weights_ref']).rsplit('_s', 1)[1])
    cfg = cfg_dict(z)
    B = int(z['regen.att'][1])
    torch.manual_seed(seed)                              # same constructors, same seed -> the reference's initial weights
    m = models.AlternatingJointModel(make_opt(cfg, B))
    base = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    for k, v in base.items():
        _check_digest(v, z['wdig.' + k], 'weight ' + k)
    a_seed, B, K, D = (int(x) for x in z['regen.att'])
    g = torch.Generator().manual_seed(1234 + a_seed)     # gen_golden.make_batch
    att = torch.randn(B, K, D, generator=g).abs() * 0.5
    att = att * torch.from_numpy(z['regen.att_rowscale']).view(B, 1, 1)      # gen_golden.widen
    z['att_raw'], z['fc'] = att.numpy(), att.mean(1).numpy()
    _check_digest(z['att_raw'], z['dig.att_raw'], 'att_raw')
    _check_digest(z['fc'], z['dig.fc'], 'fc')
    V1 = int(cfg['vocab_size']) + 1
    T = int(cfg['seq_length']) + 1
    for key in [k for k in z if k.endswith('.gumbel_u_regen')]:
        pre = key[:-len('gumbel_u_regen')]
        u_seed, skip, n = (int(x) for x in z[key])
        gu = torch.Generator().manual_seed(u_seed)       # one generator for the whole step: Recorder.inject
        for _ in range(skip):
            torch.rand(B, V1, generator=gu)
        u = np.full((T, B, V1), 0.5, np.float32)
        for step in z[pre + 'gumbel_u_steps']:
            u[int(step)] = torch.rand(B, V1, generator=gu).numpy()
        _check_digest(u, z[pre + 'gumbel_u_digest'], pre + 'gumbel_u')
        z[pre + 'gumbel_u'] = u
    return base


def load_case(name):
    """-> dict with arrays; 'weights' = reconstructed {param name: f32 array}."""
    z = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    w = {}
    if 'weights_ref' in z and str(z['weights_ref']).startswith('weights_regen_s'):
        w = _redraw(z)
    elif 'weights_ref' in z:
        base = np.load(os.path.join(GOLDEN, str(z['weights_ref']) + '.npz'), allow_pickle=False)
        strip = str(z['weights_strip']) if 'weights_strip' in z else ''
        w = {k[len(strip):]: base[k].copy() for k in base.files if k.startswith(strip)}
    for k in list(z.keys()):
        if k.startswith('wscale.'):
            w[k[7:]] = w[k[7:]] * np.float32(z[k])
        elif k.startswith('w.'):
            w[k[2:]] = z[k]
    z['weights'] = w
    for k in list(z.keys()):
        if k.endswith('_keep'):
            z[k] = z[k].astype(np.float32)
    return z


def noise_dict(z, prefix):
    """Collect '<prefix>.<key>' entries into a noise dict of numpy arrays."""
    p = prefix + '.'
    return {k[len(p):]: v for k, v in z.items() if k.startswith(p)}


def cfg_dict(z):
    out = {}
    for k, v in z.items():
        if k.startswith('cfg.'):
            v = v.item() if v.dtype.kind in 'fiu' else str(v)
            if isinstance(v, float) and v == int(v) and k[4:] in (
                    'vocab_size', 'seq_length', 'decoding_constraint', 'vse_max_violation',
                    'vse_no_imgnorm', 'vse_use_abs', 'use_gen_cider_scores') + WIDTH_KEYS:
                v = int(v)
            out[k[4:]] = v
    return out


def gts_list(z):
    out, o = [], 0
    for n in z['gts_count']:
        out.append(z['gts_flat'][o:o + int(n)])
        o += int(n)
    return out


def make_opt(cfg, B, **kw):
    """argparse.Namespace with the fields the reference constructors read (golden-fixture sizes)."""
    import argparse
    d = dict(vocab_size=97, input_encoding_size=64, rnn_size=64, num_layers=1, drop_prob_lm=0.0,
             seq_length=16, fc_feat_size=96, att_feat_size=96, att_hid_size=64,
             retrieval_reward='gumbel', gumbel_temp=1.0, multinomial_temp=1.0,
             prob_gumbel_softmax=0.5, prob_multinomial_soft=0.5, use_bn=0, decoding_constraint=0,
             rnn_type='lstm', caption_model='att2in2', vse_model='fc', share_embed=0, phase=None,
             vse_embed_size=128, vse_no_imgnorm=0, vse_use_abs=0, vse_num_layers=1,
             vse_rnn_type='gru', vse_pool_type='last', vse_margin=0.2, vse_measure='cosine',
             vse_max_violation=1, vse_loss_type='contrastive', batch_size=B, vse_loss_weight=0,
             caption_loss_weight=0, alternating_turn=['speaker', 'listener'],
             retrieval_reward_weight=0.01, reinforce_baseline_type='gt', only_one_retrieval='off',
             cider_optimization=0.99, use_gen_cider_scores=0, is_alternating=0, start_from=None,
             initialize_retrieval=None, df='corpus')
    for k, v in cfg.items():
        if k in d:
            d[k] = v
    d.update(kw)
    return argparse.Namespace(**d)
