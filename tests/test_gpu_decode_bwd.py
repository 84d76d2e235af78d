"""GPU: backward of the speaker decode engine against the oracle's autograd, driven with
the same weights and noise.  Objective = <one_hot, G> + <sampled logprobs, w> for random G, w,
which exercises both gradient entries (straight-through one-hot and log-prob gather)."""
import numpy as np
import pytest
import torch

import golden_util as GU

pytestmark = pytest.mark.gpu


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _compare(grads, P, rtol=5e-4):
    # d/d(alpha_net.bias) is exactly 0 in exact arithmetic (softmax shift invariance): both sides hold
    # rounding noise only, so the absolute floor is tied to the overall gradient scale
    glob = max(float(p.grad.abs().mean()) for p in P.values() if p.grad is not None)
    for k, p in P.items():
        ref = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
        got = grads[k].cpu().numpy()
        scale = np.abs(ref).mean() + 1e-12
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=rtol * scale + 1e-5 * glob, err_msg=k)


def _run(W, cfg, att_raw, noise, mode_name, seed, att_masks=None):
    from cooperativeimagecaptioning_amd import engine, _lib
    from oracle import speaker as S
    B, K, D = att_raw.shape
    H = W['core.h2h.weight'].shape[1]
    E = W['core.i2h.weight'].shape[1]
    A = W['ctx2att.weight'].shape[0]
    V, T = cfg['vocab_size'], cfg['seq_length']
    p = cfg['drop_prob_lm']
    rr = {'gumbel_st': 'gumbel', 'multinomial_st': 'multinomial', 'multinomial': 'reinforce',
          'gumbel_ps': 'gumbel_softmax', 'multinomial_ps': 'multinomial_soft'}[mode_name]
    ps = mode_name.endswith('_ps')
    # ---- oracle with autograd
    P = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    opt = {'sample_max': 0, 'temperature': 1, 'use_one_hot': 0 if rr == 'reinforce' else 1}
    res = S.sample(P, cfg, att_raw.mean(1), att_raw, att_masks, opt, noise, rr)
    g = torch.Generator().manual_seed(seed)
    if rr == 'reinforce':
        seq, slp = res
        oh = None
    else:
        seq, oh, slp = res
    L = seq.shape[1]
    w2 = torch.randn(B, T, generator=g)
    G = torch.randn(T, B, V + 1, generator=g)
    obj = (slp * w2[:, :L]).sum()
    if oh is not None:
        obj = obj + (oh[:, :, :V + 1] * G[:L].permute(1, 0, 2)).sum()
    obj.backward()
    # ---- HIP engine
    Wd = {k: v.cuda().contiguous() for k, v in W.items()}
    dims = engine.speaker_dims(B, K, D, H, E, A, V, T, p)
    params = engine.speaker_params(Wd)
    raw_d = att_raw.cuda().contiguous()
    att_pre = engine.speaker_att_embed_fwd(dims, params, raw_d)

    def nz(key, dt=None):
        if noise is None or key not in noise or (p == 0.0 and key.endswith('_keep')):
            return None
        t = noise[key]
        return (t.to(dt) if dt is not None else t).cuda().contiguous()
    mode = dict(multinomial=_lib.SAMPLE_MULTINOMIAL, gumbel_st=_lib.SAMPLE_GUMBEL_ST,
                multinomial_st=_lib.SAMPLE_MULTINOMIAL_ST, gumbel_ps=_lib.SAMPLE_GUMBEL_PS,
                multinomial_ps=_lib.SAMPLE_MULTINOMIAL_PS)[mode_name]
    temp = (cfg['gumbel_temp'] if mode_name.startswith('gumbel') else
            cfg['multinomial_temp'] if mode_name in ('multinomial_st', 'multinomial_ps') else 1.0)
    ps_prob = cfg['prob_gumbel_softmax'] if mode_name == 'gumbel_ps' else cfg['prob_multinomial_soft']
    am_d = att_masks.cuda().contiguous() if att_masks is not None else None
    f = engine.speaker_decode_fwd(dims, params, att_pre, mode, temp, am_d, nz('att_keep', torch.uint8),
                                  nz('x_keep', torch.uint8), nz('out_keep', torch.uint8), nz('gumbel_u'), nz('pick'),
                                  int(cfg.get('decoding_constraint', 0)), want_stv=(oh is not None), ps_u=nz('ps_u') if ps else None,
                                  ps_prob=ps_prob if ps else 0.0)
    assert int(f['L']) == L
    if ps:   # the soft rows themselves (time-major here, [B,L,V+2] in the oracle)
        np.testing.assert_allclose(f['soft'][:L].transpose(0, 1).cpu().numpy(), oh[:, :, :V + 1].detach().numpy(),
                                   rtol=2e-4, atol=2e-7)
    np.testing.assert_array_equal(f['seq'][:, :L].cpu().numpy(), seq.numpy())
    grads = {k: torch.zeros_like(v) for k, v in Wd.items()}
    engine.speaker_decode_bwd(dims, params, f, grads, raw_d, d_onehot=G.cuda().contiguous() if oh is not None else None,
                              dslp=w2.cuda().contiguous())
    torch.cuda.synchronize()
    _compare(grads, P)


@pytest.mark.parametrize('name,mode', [('sample_gumbel_st', 'gumbel_st'), ('sample_gumbel_st_tau', 'gumbel_st'),
                                       ('sample_multinomial_st', 'multinomial_st'),
                                       ('sample_multinomial_plain', 'multinomial'),
                                       ('sample_gumbel_ps', 'gumbel_ps'), ('sample_multinomial_ps', 'multinomial_ps'),
                                       ('sample_multinomial_ps_tau', 'multinomial_ps'),
                                       ('masked_sample_gumbel_st', 'gumbel_st')])
def test_decode_bwd_golden_inputs(name, mode):
    z = GU.load_case(name)
    cfg = GU.cfg_dict(z)
    W = {k: T_(v) for k, v in z['weights'].items()}
    noise = {k: T_(v) for k, v in GU.noise_dict(z, 'noise').items()}
    _run(W, cfg, T_(z['att_raw']), noise, mode, 3, T_(z['att_masks']) if 'att_masks' in z else None)


@pytest.mark.parametrize('masked', [False, True])
def test_decode_bwd_flagship_dims(masked):
    """H=E=A=512, K=36, V=9487 (the shapes the fast kernel paths are built for), small batch; `masked`: ragged region
    counts (10..36 regions per image, zero-padded features, att_masks as dataloader.py:224-229 builds them)."""
    g = torch.Generator().manual_seed(21)
    B, K, D, H, V, T = 8, 36, 256, 512, 9487, 16

    def lin(o, i, s=1.0):
        r = s / np.sqrt(i)
        return (torch.rand(o, i, generator=g) * 2 - 1) * r, (torch.rand(o, generator=g) * 2 - 1) * r
    W = {'embed.0.weight': torch.randn(V + 2, H, generator=g)}
    for nm, (o, i, s) in {'att_embed.0': (H, D, 1), 'logit': (V + 1, H, 6), 'ctx2att': (H, H, 1), 'core.a2c': (2 * H, H, 1),
                          'core.i2h': (5 * H, H, 1), 'core.h2h': (5 * H, H, 1), 'core.attention.h2att': (H, H, 1),
                          'core.attention.alpha_net': (1, H, 3)}.items():
        W[nm + '.weight'], W[nm + '.bias'] = lin(o, i, s)
    W['logit.bias'][0] = 3.0
    cfg = dict(vocab_size=V, seq_length=T, drop_prob_lm=0.5, gumbel_temp=1.0, multinomial_temp=1.0,
               prob_gumbel_softmax=1, prob_multinomial_soft=1, decoding_constraint=0)
    att_raw = torch.randn(B, K, D, generator=g).abs() * 0.5
    noise = dict(att_keep=(torch.rand(B, K, H, generator=g) >= 0.5).float(),
                 x_keep=(torch.rand(T + 1, B, H, generator=g) >= 0.5).float(),
                 out_keep=(torch.rand(T + 1, B, H, generator=g) >= 0.5).float(),
                 gumbel_u=torch.rand(T + 1, B, V + 1, generator=g))
    am = None
    if masked:
        lens = torch.tensor([36, 10, 25, 36, 17, 30, 12, 33])
        am = (torch.arange(K).unsqueeze(0) < lens.unsqueeze(1)).float()
        att_raw = att_raw * am.unsqueeze(2)
    _run(W, cfg, att_raw, noise, 'gumbel_st', 4, am)


def test_decode_bwd_partial_sampling_flagship_dims():
    """Partial-sampling Gumbel (gumbel_softmax.py) at the flagship widths: the logit-layer backward runs inside
    the time loop and the embedding gradient is a dense [V+1, T*B] x [T*B, E] product."""
    g = torch.Generator().manual_seed(22)
    B, K, D, H, V, T = 4, 36, 128, 512, 9487, 16

    def lin(o, i, s=1.0):
        r = s / np.sqrt(i)
        return (torch.rand(o, i, generator=g) * 2 - 1) * r, (torch.rand(o, generator=g) * 2 - 1) * r
    W = {'embed.0.weight': torch.randn(V + 2, H, generator=g)}
    for nm, (o, i, s) in {'att_embed.0': (H, D, 1), 'logit': (V + 1, H, 6), 'ctx2att': (H, H, 1), 'core.a2c': (2 * H, H, 1),
                          'core.i2h': (5 * H, H, 1), 'core.h2h': (5 * H, H, 1), 'core.attention.h2att': (H, H, 1),
                          'core.attention.alpha_net': (1, H, 3)}.items():
        W[nm + '.weight'], W[nm + '.bias'] = lin(o, i, s)
    W['logit.bias'][0] = 3.0
    cfg = dict(vocab_size=V, seq_length=T, drop_prob_lm=0.5, gumbel_temp=0.7, multinomial_temp=1.0,
               prob_gumbel_softmax=0.5, prob_multinomial_soft=0.5, decoding_constraint=0)
    att_raw = torch.randn(B, K, D, generator=g).abs() * 0.5
    noise = dict(att_keep=(torch.rand(B, K, H, generator=g) >= 0.5).float(),
                 x_keep=(torch.rand(T + 1, B, H, generator=g) >= 0.5).float(),
                 out_keep=(torch.rand(T + 1, B, H, generator=g) >= 0.5).float(),
                 gumbel_u=torch.rand(T + 1, B, V + 1, generator=g), ps_u=torch.rand(T + 1, B, generator=g))
    _run(W, cfg, att_raw, noise, 'gumbel_ps', 5)


def test_decode_bwd_with_the_decoding_constraint():
    """a gradient decode with --decoding_constraint 1 (AttModel.py:438-442: the previously emitted word's
    column is -inf before the log-softmax).  The backward rebuilds log-probs from raw logits + lse, so it has to mask the
    same column: p = 0 there, no gradient.  Plain multinomial decode (the REINFORCE path: the only one where the reference's
    scatter_ of seq[-1] is defined), recorded weights and dropout masks, picks without immediate repeats."""
    z = GU.load_case('sample_multinomial_plain')
    cfg = dict(GU.cfg_dict(z), decoding_constraint=1)
    W = {k: T_(v) for k, v in z['weights'].items()}
    noise = {k: T_(v) for k, v in GU.noise_dict(z, 'noise').items()}
    pick = noise['pick'].clone()
    V = cfg['vocab_size']
    for t in range(2, pick.shape[0]):
        same = pick[t] == pick[t - 1]
        pick[t][same] = (pick[t][same] % V) + 1          # another word, never <eos>
    noise['pick'] = pick
    _run(W, cfg, T_(z['att_raw']), noise, 'multinomial', 6)


@pytest.mark.parametrize('temp', [0.5, 2.0])
def test_decode_bwd_multinomial_st_with_a_temperature(temp):
    """ST-multinomial at multinomial_temp != 1 (run_joint.sh passes --multinomial_temp) on a small
    vocabulary, where the row partials come from the fallback kernel with fewer than 64 parts per row: empty partials
    (-FLT_MAX) times 1/temp > 1 overflowed to -inf and (-inf) - (-inf) poisoned the straight-through value of every row."""
    z = GU.load_case('sample_multinomial_st')
    cfg = dict(GU.cfg_dict(z), multinomial_temp=temp)
    W = {k: T_(v) for k, v in z['weights'].items()}
    noise = {k: T_(v) for k, v in GU.noise_dict(z, 'noise').items()}
    _run(W, cfg, T_(z['att_raw']), noise, 'multinomial_st', 7)
