"""share_embed = 1 fixtures (tests/golden/share_*.npz: several iterations of the reference's trainer recorded through
tools/gen_golden.py --only-share-embed): loading, and the replay through the oracle that both the CPU test and the GPU test
compare against."""
import numpy as np
import torch

import golden_util as GU

SHARE_CASES = ['share_joint_gumbel', 'share_reinforce']
SPK_EMBED, LST_EMBED = 'caption_generator.embed.0.weight', 'vse.txt_enc.embed.weight'


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load(name):
    z = dict(np.load(GU.GOLDEN + '/' + name + '.npz', allow_pickle=False))
    for k in list(z):
        if k.endswith('_keep'):
            z[k] = z[k].astype(np.float32)
    cfg = GU.cfg_dict(z)
    cfg['share_embed'] = 1
    cfg['learning_rate'] = float(z['cfg.learning_rate'])
    cfg['grad_clip'] = float(z['cfg.grad_clip'])
    weights = {k[2:]: v for k, v in z.items() if k.startswith('w.')}
    assert np.array_equal(weights[SPK_EMBED], weights[LST_EMBED])
    return z, cfg, weights


def step_view(z, s):
    """The entries of iteration s as a fixture dict of their own (the names the single-step helpers expect)."""
    pre = f's{s}.'
    d = {k[len(pre):]: v for k, v in z.items() if k.startswith(pre)}
    d['turn'] = str(d['turn'])
    return d


def step_noise(d, cfg, turn):
    """Recorded decodes of one iteration -> the oracle's / the product's named decodes (tests/test_oracle_golden.joint_noise)."""
    n = int(d['n_decodes'])
    decs = [{k: v for k, v in GU.noise_dict(d, f'noise{i}').items()} for i in range(n)]
    if turn == 'listener':
        names = ['sample']
    else:
        names = []
        if cfg['retrieval_reward_weight'] > 0:
            names.append('sample')
        if cfg['cider_optimization']:
            names.append('greedy')
    assert len(names) == n, (names, n)
    return dict(zip(names, decs))


def step_batch(d):
    o, gts = 0, []
    for c in d['gts_count']:
        gts.append(d['gts_flat'][o:o + int(c)])
        o += int(c)
    return dict(fc_feats=d['fc_feats'], att_feats=d['att_feats'], att_masks=None, labels=d['labels'], masks=d['masks'], gts=gts)
