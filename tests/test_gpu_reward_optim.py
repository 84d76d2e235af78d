"""GPU: CIDEr-D reward kernels (exact integer n-gram / df tables, fp64 scores), sequence
losses and clamp+Adam against the golden fixtures from the reference and the oracle."""
import numpy as np
import pytest
import torch

import golden_util as GU

pytestmark = pytest.mark.gpu


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _pad(a, T=16):
    out = np.zeros((a.shape[0], T), np.int32)
    out[:, :a.shape[1]] = a
    return out


def _decode_key(k):
    n = (k >> 60) & 0x7
    return tuple(int((k >> (45 - 15 * j)) & 0x7fff) for j in range(n))


def _run_cider(gen, greedy, gts, debug=False):
    from cooperativeimagecaptioning_amd import engine
    refs, off = engine.pack_refs(gts, 'cuda')
    Lg = torch.tensor([gen.shape[1]], dtype=torch.int32).cuda()
    Lr = torch.tensor([greedy.shape[1]], dtype=torch.int32).cuda()
    out = engine.ciderd_reward(T_(_pad(gen)).cuda(), Lg, T_(_pad(greedy)).cuda(), Lr, refs, off, debug=debug)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize('name', ['ciderd', 'ciderd_spi2'])
def test_ciderd_golden(name):
    from oracle import ciderd
    z = GU.load_case(name)
    gts = GU.gts_list(z)
    out = _run_cider(z['gen'], z['greedy'], gts, debug=True)
    np.testing.assert_allclose(out['reward'].cpu().numpy(), z['reward'].astype(np.float32), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(out['stats'][1].item(), float(z['cider_greedy']), rtol=1e-12)
    if 'cider_gen' in z:
        np.testing.assert_allclose(out['scores'][:len(z['cider_gen'])].cpu().numpy(), z['cider_gen'], rtol=1e-11, atol=1e-13)
    # exact integer tables: every sentence's n-gram counts equal the oracle's dict, bit for bit
    B = z['gen'].shape[0]
    sents = [ciderd.row_to_tokens(r) for r in z['gen']] + [ciderd.row_to_tokens(r) for r in z['greedy']] + \
            [ciderd.row_to_tokens(r) for g in gts for r in g]
    keys, cnt, nu = out['dbg_keys'].cpu().numpy(), out['dbg_cnt'].cpu().numpy(), out['dbg_nuniq'].cpu().numpy()
    for s, toks in enumerate(sents):
        want = dict(ciderd.precook(toks))
        got = {_decode_key(int(keys[s, i]) & 0xFFFFFFFFFFFFFFFF): int(cnt[s, i]) for i in range(nu[s])}
        assert got == want, f'sentence {s}'
    # exact document frequencies
    spi = B // len(gts)
    crefs = [[ciderd.precook(ciderd.row_to_tokens(r)) for r in gts[(i % B) // spi]] for i in range(2 * B)]
    df = ciderd.compute_doc_freq(crefs)
    dfg = out['dbg_df'].cpu().numpy()
    for s in range(len(sents)):
        for i in range(nu[s]):
            ng = _decode_key(int(keys[s, i]) & 0xFFFFFFFFFFFFFFFF)
            assert int(dfg[s, i]) == int(df.get(ng, 0)), (s, ng)


def test_ciderd_out_of_vocabulary_token_poisons_the_scores():
    """A token id beyond the declared vocabulary would alias another word's 15-bit key field: the scores become NaN
    (and a vocabulary that cannot be represented at all is refused, tests/test_abi.py)."""
    from cooperativeimagecaptioning_amd import engine, _lib
    z = GU.load_case('ciderd')
    gts = GU.gts_list(z)
    refs, off = engine.pack_refs(gts, 'cuda')
    gen, greedy = _pad(z['gen']), _pad(z['greedy'])
    Lg = torch.tensor([z['gen'].shape[1]], dtype=torch.int32).cuda()
    Lr = torch.tensor([z['greedy'].shape[1]], dtype=torch.int32).cuda()
    ok = engine.ciderd_reward(T_(gen).cuda(), Lg, T_(greedy).cuda(), Lr, refs, off, vocab_size=23)
    assert torch.isfinite(ok['scores']).all()
    bad = gen.copy()
    bad[1, 0] = 40000                                     # row 1 has no <eos>: the token is inside the caption
    out = engine.ciderd_reward(T_(bad).cuda(), Lg, T_(greedy).cuda(), Lr, refs, off, vocab_size=23)
    assert torch.isnan(out['scores']).all()
    with pytest.raises(_lib.CicError):
        engine.ciderd_reward(T_(gen).cuda(), Lg, T_(greedy).cuda(), Lr, refs, off, vocab_size=40000)


def test_ciderd_random_full_size():
    """B=128, 5 refs each, vocabulary 9487, ragged lengths, L < 16 for the greedy half."""
    from oracle import ciderd
    rs = np.random.RandomState(3)
    B, V = 128, 9487

    def rows(n, L, p0):
        a = np.minimum(rs.zipf(1.2, size=(n, L)), V).astype(np.int64)
        for i in range(n):
            if rs.rand() < p0:
                a[i, rs.randint(0, L):] = 0
        return a
    gen, greedy = rows(B, 16, 0.8), rows(B, 13, 0.8)
    gts = [rows(rs.randint(3, 8), 16, 0.95) for _ in range(B)]
    gen[:5] = gts[0][0][None, :]                           # shared n-grams across images
    reward, cg = ciderd.get_self_critical_reward(gts, gen, greedy)
    out = _run_cider(gen, greedy, gts)
    np.testing.assert_allclose(out['reward'].cpu().numpy(), reward.astype(np.float32), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(out['stats'][1].item(), cg, rtol=1e-11)


def test_seq_loss_and_nll():
    from cooperativeimagecaptioning_amd import engine
    g = torch.Generator().manual_seed(1)
    B, T, L = 37, 16, 11
    slp = -torch.rand(B, T, generator=g) * 5
    seq = torch.randint(0, 4, (B, T), generator=g)
    seq[:, L:] = 3
    coef = torch.randn(B, generator=g)
    m = torch.cat([torch.ones(B, 1), (seq[:, :L - 1] > 0).float()], 1)
    want = (slp[:, :L] * (-coef).unsqueeze(1) * m).sum() / m.sum()
    dslp = torch.full((B, T), 7.0).cuda()
    loss = engine.seq_loss(slp.cuda(), seq.int().cuda(), torch.tensor([L], dtype=torch.int32).cuda(), coef.cuda(),
                           -1.0, 0.99, dslp=dslp, accumulate=True)
    np.testing.assert_allclose(float(loss), float(want), rtol=1e-5)
    wd = torch.zeros(B, T)
    wd[:, :L] = 0.99 * (-coef).unsqueeze(1) * m / m.sum()
    np.testing.assert_allclose(dslp.cpu().numpy(), (wd + 7.0).numpy(), rtol=1e-5, atol=1e-7)
    mask = (torch.rand(B, T + 2, generator=g) > 0.4).float()
    want = -(slp * mask[:, 1:T + 1]).sum() / mask[:, 1:T + 1].sum()
    d2 = torch.empty(B, T).cuda()
    loss = engine.masked_nll(slp.cuda(), mask.cuda()[:, 1:], 0.5, dslp=d2)
    np.testing.assert_allclose(float(loss), float(want), rtol=1e-5)
    np.testing.assert_allclose(d2.cpu().numpy(), (-0.5 * mask[:, 1:T + 1] / mask[:, 1:T + 1].sum()).numpy(), rtol=1e-5)


def test_clamp_adam_golden():
    from cooperativeimagecaptioning_amd import engine
    z = GU.load_case('clamp_adam')
    n = z['p0'].size
    pad = (-n) % 4
    p = torch.cat([T_(z['p0']).view(-1), torch.zeros(pad)]).cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i in range(z['grads'].shape[0]):
        g = torch.cat([T_(z['grads'][i]).view(-1), torch.zeros(pad)]).cuda()
        engine.clamp_adam(p, g, m, v, float(z['lr']), i + 1, float(z['grad_clip']))
        np.testing.assert_allclose(p[:n].cpu().numpy(), z['traj'][i].reshape(-1), rtol=2e-6, atol=2e-7)


def test_clamp_adam_clears_the_gradient_on_request():
    """cic_clamp_adam_zero: the update of cic_clamp_adam bit for bit, and the gradient buffer left at zero (the next
    step's zero_grad() folded into the kernel); odd length = the scalar tail path."""
    from cooperativeimagecaptioning_amd import engine
    g0 = torch.randn(4099, generator=torch.Generator().manual_seed(2)).cuda()
    outs = []
    for zero in (False, True):
        p = torch.linspace(-1, 1, 4099).cuda()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for step in (1, 2):
            g = (g0 * step).clone()
            engine.clamp_adam(p, g, m, v, 5e-4, step, 0.1, zero_grad=zero)
            assert bool((g == 0).all()) == zero
        outs.append((p.clone(), m.clone(), v.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
