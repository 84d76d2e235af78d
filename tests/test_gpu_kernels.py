"""GPU: each HIP kernel behind the C ABI against the CPU oracle (oracle/) on seeded inputs,
and against the golden fixtures made from the reference.  Tolerances are written per test;
token indices / integer outputs are compared exactly."""
import numpy as np
import pytest
import torch

import golden_util as GU

pytestmark = pytest.mark.gpu


def dev(t):
    return t.cuda().contiguous()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope='module')
def ops():
    from cooperativeimagecaptioning_amd import ops as o
    return o


# ------------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [
    # M, N, K, a_kc, b_kc
    (128, 512, 512, True, True),        # per-step h2att (skinny tile)
    (128, 9488, 512, True, True),       # per-step logit
    (4608, 512, 2048, True, True),      # att_embed (big tile)
    (2176, 512, 9488, True, False),     # d_out = dlogits @ W_logit
    (9488, 512, 2048, False, False),    # dW_logit = dlogits^T @ out
    (512, 2048, 4608, False, False),    # dW_att_embed
    (128, 128, 1024, True, True),       # contrastive scores
    (6, 98, 64, True, True),            # golden-size, unaligned N and M edge
    (102, 64, 98, True, False),         # K not a multiple of 4 -> scalar loads
    (70, 33, 45, False, True),
    (33, 70, 19, False, False),
]


@pytest.mark.parametrize('M,N,K,a_kc,b_kc', GEMM_SHAPES)
def test_gemm(ops, M, N, K, a_kc, b_kc):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((M, K) if a_kc else (K, M), generator=g)
    B = torch.randn((N, K) if b_kc else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    opA = A if a_kc else A.t()
    opB = B.t() if b_kc else B
    ref = opA.double() @ opB.double()
    Cd = dev(C0.clone())
    ops.gemm(dev(A), dev(B), Cd, a_kc, b_kc)
    torch.cuda.synchronize()
    # f32 fma chain vs f64: error ~ 1e-7 * sum|a*b| <= few e-6 * sqrt(K)
    tol = 3e-6 * np.sqrt(K) * 4
    np.testing.assert_allclose(Cd.cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=tol)
    # bias + accumulate + relu epilogue
    Cd = dev(C0.clone())
    ops.gemm(dev(A), dev(B), Cd, a_kc, b_kc, bias=dev(bias), accumulate=True, relu=True)
    ref2 = torch.relu(ref + bias.double() + C0.double())
    np.testing.assert_allclose(Cd.cpu().double().numpy(), ref2.numpy(), rtol=1e-5, atol=tol)


@pytest.mark.parametrize('M,N,K,a_kc,b_kc', [(4608, 512, 2048, True, True), (2048, 512, 9488, True, False),
                                              (9488, 512, 2048, False, False), (512, 2048, 4608, False, True),
                                              (300, 200, 136, True, True), (260, 132, 72, False, False)])
def test_gemm_precisions(ops, M, N, K, a_kc, b_kc):
    """The LDS-tiled products at their three arithmetic settings (cic.h, cic_gemm_args.precision) against an f64 product
    of the same f32 operands: the default - operands cut into three bf16 parts, six part products on the bf16 matrix
    cores - must be as accurate as the f32-input MFMA's own fma chain (both ~1e-7 of sum |a b|); plain bf16 operands
    are the reduced-precision setting (2^-8 per operand)."""
    g = torch.Generator().manual_seed(11 * M + N + K)
    A = torch.randn((M, K) if a_kc else (K, M), generator=g) * torch.logspace(-2, 2, K if a_kc else M, base=2.0)
    B = torch.randn((N, K) if b_kc else (K, N), generator=g)
    opA, opB = (A if a_kc else A.t()).double(), (B.t() if b_kc else B).double()
    ref = opA @ opB
    mag = (opA.abs() @ opB.abs()).numpy()                 # sum_k |a b|: the scale rounding errors ride on
    err = {}
    for name, prec in (('f32', 0), ('f32_mfma', 1), ('bf16', 2)):
        Cd = dev(torch.zeros(M, N))
        ops.gemm(dev(A), dev(B), Cd, a_kc, b_kc, precision=prec)
        torch.cuda.synchronize()
        err[name] = float(np.max(np.abs(Cd.cpu().double().numpy() - ref.numpy()) / mag))
    assert err['f32_mfma'] < 4e-7 and err['f32'] < 4e-7, err      # f32-level: a few 2^-24 of sum |a b|
    assert 1e-5 < err['bf16'] < 8e-3, err                          # bf16 operands: ~2^-9 .. 2^-8 per factor
    # every setting is deterministic run to run (forward products keep a fixed summation order)
    C1, C2 = dev(torch.zeros(M, N)), dev(torch.zeros(M, N))
    ops.gemm(dev(A), dev(B), C1, a_kc, b_kc)
    ops.gemm(dev(A), dev(B), C2, a_kc, b_kc)
    assert torch.equal(C1, C2)


@pytest.mark.parametrize('M,N,K,K2', [(256, 9488, 512, 0), (100, 9488, 512, 0), (256, 2560, 512, 512), (37, 3072, 512, 512),
                                      (200, 2048, 1024, 0)])
def test_walker_products_on_bf16_parts_are_f32_accurate(ops, M, N, K, K2):
    """The per-timestep walkers (logit product: K = 512, N >= 2048; gate product: K + K2 = 1024, N >= 2048) in their
    default form - operands cut into three bf16 parts - against an f64 product of the same f32 operands, next to their
    f32-input MFMA form: both within a few 2^-24 of sum |a b|; rows beyond M (ragged strips) are never written."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.logspace(-2, 2, K, base=2.0)
    W = torch.randn(N, K, generator=g) * 0.05
    A2 = torch.randn(M, K2, generator=g) if K2 else None
    W2 = torch.randn(N, K2, generator=g) * 0.05 if K2 else None
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    mag = A.double().abs() @ W.double().abs().t() + bias.double().abs()
    if K2:
        ref += A2.double() @ W2.double().t()
        mag += A2.double().abs() @ W2.double().abs().t()
    err = {}
    for name, prec in (('f32', 0), ('f32_mfma', 1)):
        Cd = dev(torch.full((M + 2, N), 7.0))
        ops.gemm(dev(A), dev(W), Cd[:M], True, True, bias=dev(bias), A2=dev(A2) if K2 else None,
                 B2=dev(W2) if K2 else None, precision=prec)
        torch.cuda.synchronize()
        err[name] = float(np.max(np.abs(Cd[:M].cpu().double().numpy() - ref.numpy()) / mag.numpy()))
        assert float((Cd[M:] - 7.0).abs().max()) == 0.0, name
    assert err['f32'] < 4e-7 and err['f32_mfma'] < 4e-7, err


def test_gemm_dual_and_strided(ops):
    # pre = x W_i2h^T + h W_h2h^T + b, written into a column window of a wider buffer
    g = torch.Generator().manual_seed(5)
    Bsz, E, H = 128, 512, 512
    x, h = torch.randn(Bsz, E, generator=g), torch.randn(Bsz, H, generator=g)
    W1, W2 = torch.randn(5 * H, E, generator=g) * .05, torch.randn(5 * H, H, generator=g) * .05
    b = torch.randn(5 * H, generator=g)
    out = dev(torch.zeros(Bsz, 5 * H + 64))
    ops.gemm(dev(x), dev(W1), out[:, 32:32 + 5 * H], True, True, bias=dev(b), A2=dev(h), B2=dev(W2))
    ref = x.double() @ W1.double().t() + h.double() @ W2.double().t() + b.double()
    np.testing.assert_allclose(out[:, 32:32 + 5 * H].cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    assert float(out[:, :32].abs().max()) == 0 and float(out[:, 32 + 5 * H:].abs().max()) == 0


def test_gemm_is_deterministic(ops):
    g = torch.Generator().manual_seed(1)
    A, B = dev(torch.randn(128, 512, generator=g)), dev(torch.randn(9488, 512, generator=g))
    C1, C2 = dev(torch.empty(128, 9488)), dev(torch.empty(128, 9488))
    ops.gemm(A, B, C1)
    ops.gemm(A, B, C2)
    assert torch.equal(C1, C2)


def test_colsum(ops):
    g = torch.Generator().manual_seed(2)
    for M, N in [(2176, 9488), (128, 512), (7, 33)]:
        X = torch.randn(M, N, generator=g)
        out = dev(torch.zeros(N))
        ops.colsum(dev(X), out)
        np.testing.assert_allclose(out.cpu().numpy(), X.double().sum(0).numpy(), rtol=1e-5, atol=1e-3)
        ops.colsum(dev(X), out, accumulate=True)
        np.testing.assert_allclose(out.cpu().numpy(), 2 * X.double().sum(0).numpy(), rtol=1e-5, atol=2e-3)


# ------------------------------------------------------------------------------- RNG
def test_rng(ops):
    n = 1 << 20
    u = dev(torch.empty(n))
    ops.uniform_(u, seed=7, offset=0)
    u2 = dev(torch.empty(n))
    ops.uniform_(u2, seed=7, offset=0)
    assert torch.equal(u, u2)
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 2e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
    ops.uniform_(u2, seed=8, offset=0)
    assert not torch.equal(u, u2)
    # offset continues the same stream: offset counts Philox calls (4 values each)
    u3 = dev(torch.empty(n - 4096))
    ops.uniform_(u3, seed=7, offset=1024)
    assert torch.equal(u3, u[4096:])
    # values are multiples of 2^-24 (24-bit mantissa draw like torch.rand)
    assert torch.equal((u * 16777216.0).round() / 16777216.0, u)
    keep = dev(torch.empty(n, dtype=torch.uint8))
    ops.dropout_keep_(keep, 0.5, seed=3)
    assert abs(float(keep.float().mean()) - 0.5) < 3e-3
    ops.dropout_keep_(keep, 0.2, seed=3)
    assert abs(float(keep.float().mean()) - 0.8) < 3e-3
    # several masks in one launch: each equals its own single launch (ragged sizes, own sub-streams)
    sizes, offs = [128 * 36 * 512, 17 * 128 * 512 + 3, 1001], [1 << 32, 2 << 32, 77]
    many = [dev(torch.zeros(k, dtype=torch.uint8)) for k in sizes]
    ops.dropout_keep_multi_(many, 0.5, 3, offs)
    for t, k, o in zip(many, sizes, offs):
        one = dev(torch.zeros(k, dtype=torch.uint8))
        ops.dropout_keep_(one, 0.5, seed=3, offset=o)
        assert torch.equal(t, one)
    # independent reference of Philox4x32-10 on the host for a few counters
    def philox(counter, seed):
        M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
        c = [counter & 0xffffffff, counter >> 32, 0, 0]
        k = [seed & 0xffffffff, seed >> 32]
        for _ in range(10):
            p0, p1 = M0 * c[0], M1 * c[2]
            c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xffffffff, p1 & 0xffffffff,
                 ((p0 >> 32) ^ c[3] ^ k[1]) & 0xffffffff, p0 & 0xffffffff]
            k = [(k[0] + W0) & 0xffffffff, (k[1] + W1) & 0xffffffff]
        return c
    uh = u.cpu().numpy()
    for q in (0, 1, 12345):
        want = [(r >> 8) / 16777216.0 for r in philox(q, 7)]
        np.testing.assert_array_equal(uh[4 * q:4 * q + 4], np.array(want, dtype=np.float32))


# ------------------------------------------------------------------------------- attention
def _rand_speaker_params(H, E, A, D, V, g):
    def lin(o, i):
        r = 1.0 / np.sqrt(i)
        return (torch.rand(o, i, generator=g) * 2 - 1) * r, (torch.rand(o, generator=g) * 2 - 1) * r
    P = {}
    P['embed.0.weight'] = torch.randn(V + 2, E, generator=g)
    for name, (o, i) in {'att_embed.0': (H, D), 'logit': (V + 1, H), 'ctx2att': (A, H),
                         'core.a2c': (2 * H, H), 'core.i2h': (5 * H, E), 'core.h2h': (5 * H, H),
                         'core.attention.h2att': (A, H), 'core.attention.alpha_net': (1, A)}.items():
        w, b = lin(o, i)
        P[name + '.weight'], P[name + '.bias'] = w, b
    return P


@pytest.mark.parametrize('B,K,H,masked', [(128, 36, 512, False), (16, 36, 512, True), (6, 7, 64, False),
                                          (5, 7, 64, True), (4, 50, 512, True), (3, 20, 1024, False),
                                          (3, 36, 256, False)])
def test_attn_fwd(ops, B, K, H, masked):
    from oracle import speaker as S
    g = torch.Generator().manual_seed(B + K + H)
    P = _rand_speaker_params(H, H, H, 32, 20, g)
    P['core.attention.alpha_net.weight'] *= 4.0       # peaky softmax
    h = torch.randn(B, H, generator=g)
    att = torch.randn(B, K, H, generator=g).abs()
    p_att = torch.randn(B, K, H, generator=g)
    masks = None
    if masked:
        masks = (torch.rand(B, K, generator=g) > 0.3).float()
        masks[:, 0] = 1
    att_res, alpha = S.attention_step(P, h, att, p_att, masks)
    att_h = torch.nn.functional.linear(h, P['core.attention.h2att.weight'], P['core.attention.h2att.bias'])
    o_res, o_alpha, o_dot = dev(torch.empty(B, H)), dev(torch.empty(B, K)), dev(torch.empty(B, K))
    ops.attn_fwd(dev(att_h), dev(p_att), dev(att), dev(P['core.attention.alpha_net.weight'].view(-1)),
                 dev(P['core.attention.alpha_net.bias']), dev(masks) if masked else None, o_res, o_alpha, o_dot)
    # f32 streaming reduction over H<=1024 terms of O(1): 1e-5 relative / 2e-6 absolute
    np.testing.assert_allclose(o_alpha.cpu().numpy(), alpha.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(o_res.cpu().numpy(), att_res.numpy(), rtol=2e-5, atol=5e-6)
    assert abs(float(o_alpha.sum(1).mean()) - 1.0) < 1e-5


def test_attn_fwd_golden(ops):
    z = GU.load_case('kernels_speaker')
    W = {k: T(v) for k, v in z['weights'].items()}
    h, att, p_att = T(z['h']), T(z['att']), T(z['p_att'])
    att_h = torch.nn.functional.linear(h, W['core.attention.h2att.weight'], W['core.attention.h2att.bias'])
    B, K, H = att.shape
    for masks, key in ((None, 'att_res'), (T(z['att_masks']), 'att_res_masked')):
        o_res, o_alpha = dev(torch.empty(B, H)), dev(torch.empty(B, K))
        ops.attn_fwd(dev(att_h), dev(p_att), dev(att), dev(W['core.attention.alpha_net.weight'].view(-1)),
                     dev(W['core.attention.alpha_net.bias']), dev(masks) if masks is not None else None,
                     o_res, o_alpha, None)
        np.testing.assert_allclose(o_res.cpu().numpy(), z[key], rtol=2e-5, atol=2e-6)


# ------------------------------------------------------------------------------- cell / embed
@pytest.mark.parametrize('B,H,p', [(128, 512, 0.5), (6, 64, 0.0), (7, 128, 0.3)])
def test_cell_fwd(ops, B, H, p):
    from oracle import speaker as S
    g = torch.Generator().manual_seed(B * H)
    P = _rand_speaker_params(H, H, H, 32, 20, g)
    xt, att_res = torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)
    h, c = torch.randn(B, H, generator=g) * .5, torch.randn(B, H, generator=g)
    keep = (torch.rand(B, H, generator=g) >= p).float() if p > 0 else None
    h2, c2 = S.att2in2_cell(P, xt, att_res, h, c)
    out = S.dropout(h2, keep, p)
    F = torch.nn.functional
    pre = F.linear(xt, P['core.i2h.weight'], P['core.i2h.bias']) + F.linear(h, P['core.h2h.weight'], P['core.h2h.bias'])
    pre[:, 3 * H:] += F.linear(att_res, P['core.a2c.weight'], P['core.a2c.bias'])
    oh, oc, oo = dev(torch.empty(B, H)), dev(torch.empty(B, H)), dev(torch.empty(B, H))
    ops.cell_fwd(dev(pre), dev(c), dev(keep.to(torch.uint8)) if keep is not None else None, p, oh, oc, oo)
    np.testing.assert_allclose(oc.cpu().numpy(), c2.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(oh.cpu().numpy(), h2.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(oo.cpu().numpy(), out.numpy(), rtol=1e-5, atol=4e-6)


def test_embed_and_keep(ops):
    from oracle import speaker as S
    g = torch.Generator().manual_seed(3)
    V, E, B, p = 97, 64, 33, 0.5
    P = {'embed.0.weight': torch.randn(V + 2, E, generator=g)}
    it = torch.randint(0, V + 2, (B,), generator=g)
    keep = (torch.rand(B, E, generator=g) >= p).float()
    x = dev(torch.empty(B, E))
    ops.embed_fwd(dev(P['embed.0.weight']), dev(it.int()), dev(keep.to(torch.uint8)), p, x)
    assert torch.equal(x.cpu(), S.embed_token(P, it, keep, p))
    ops.embed_fwd(dev(P['embed.0.weight']), dev(it.int()), None, 0.0, x)
    assert torch.equal(x.cpu(), S.embed_token(P, it, None, 0.0))
    a = torch.randn(B, 7, E, generator=g)
    k3 = (torch.rand(B, 7, E, generator=g) >= p).float()
    y = dev(torch.empty(B, 7, E))
    ops.apply_keep(dev(a), dev(k3.to(torch.uint8)), p, y)
    assert torch.equal(y.cpu(), S.dropout(a, k3, p))


# ------------------------------------------------------------------------------- log-softmax + sampler
def _sampler_state(B, T_):
    return dict(unfinished=dev(torch.ones(B, dtype=torch.int32)), it_next=dev(torch.zeros(B, dtype=torch.int32)),
                seq=dev(torch.zeros(B, T_, dtype=torch.int32)), slp=dev(torch.zeros(B, T_)),
                stv=dev(torch.zeros(B, T_)), any_unfinished=dev(torch.zeros(T_ + 1, dtype=torch.int32)))


@pytest.mark.parametrize('B,V1', [(128, 9488), (6, 98), (5, 1500)])
def test_logsoftmax_greedy_and_gumbel(ops, B, V1):
    from oracle import speaker as S
    from cooperativeimagecaptioning_amd import _lib
    g = torch.Generator().manual_seed(V1)
    logits = torch.randn(B, V1, generator=g) * 3
    logits[0, 5] = logits[0, 17] = logits[0].max() + 1.0          # exact tie -> lowest index wins
    logits[1, 0] = logits[1].max() + 5.0                           # EOS
    logp = torch.log_softmax(logits, 1)
    st = _sampler_state(B, 16)
    lg = dev(logits.clone())
    ops.logsoftmax_sample(lg, _lib.SAMPLE_GREEDY, step=1, **st)
    np.testing.assert_allclose(lg.cpu().numpy(), logp.numpy(), rtol=1e-5, atol=2e-5)
    slp, it = torch.max(logp, 1)
    assert torch.equal(st['it_next'].cpu().long(), it) and int(it[0]) == 5
    np.testing.assert_allclose(st['slp'][:, 0].cpu().numpy(), slp.numpy(), rtol=1e-5, atol=2e-5)
    assert torch.equal(st['unfinished'].cpu(), (it > 0).int())
    assert torch.equal(st['seq'][:, 0].cpu().long(), it * (it > 0))
    assert int(st['any_unfinished'][1]) == 1
    # gumbel straight-through, tau = 0.5, step 2 keeps row 1 finished
    U = torch.rand(B, V1, generator=g)
    tau = 0.5
    one_hot, ind, y = S.gumbel_st(logp, tau, U)
    lg = dev(logits.clone())
    ops.logsoftmax_sample(lg, _lib.SAMPLE_GUMBEL_ST, temp=tau, U=dev(U), step=2, **st)
    zz = (logp + S.sample_gumbel_from_u(U)) / tau
    top2 = zz.topk(2, 1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4                       # rows whose arg-max cannot flip by rounding
    assert safe.float().mean() > 0.9
    got = st['it_next'].cpu().long()
    assert torch.equal(got[safe], ind[safe])
    v_ref = one_hot.gather(1, ind.view(-1, 1)).view(-1)
    unf = ((it > 0) & (ind > 0))
    np.testing.assert_allclose(st['stv'][:, 1].cpu()[safe & unf].numpy(), v_ref[safe & unf].numpy(), atol=1.2e-7)
    assert float(st['stv'][1, 1]) == 1.0 and int(st['seq'][1, 1]) == 0      # finished row -> exact EOS one-hot
    np.testing.assert_allclose(st['slp'][:, 1].cpu()[safe].numpy(),
                               logp.gather(1, ind.view(-1, 1)).view(-1)[safe].numpy(), rtol=1e-5, atol=2e-5)


def test_sampler_multinomial_pick_and_distribution(ops):
    from cooperativeimagecaptioning_amd import _lib
    g = torch.Generator().manual_seed(9)
    B, V1 = 4096, 12
    base = torch.randn(1, V1, generator=g) * 1.5
    logits = base.expand(B, V1).contiguous()
    p = torch.softmax(base[0] / 0.7, 0)
    st = _sampler_state(B, 16)
    U = dev(torch.empty(B, V1))
    ops.uniform_(U, seed=11)
    ops.logsoftmax_sample(dev(logits.clone()), _lib.SAMPLE_MULTINOMIAL, temp=0.7, U=U, step=1, **st)
    hist = torch.bincount(st['it_next'].cpu().long(), minlength=V1).float() / B
    assert float((hist - p).abs().max()) < 0.03                   # Gumbel-max draw ~ softmax(logp / temp)
    pick = torch.randint(0, V1, (B,), generator=g)
    ops.logsoftmax_sample(dev(logits.clone()), _lib.SAMPLE_MULTINOMIAL_ST, temp=1.0, pick=dev(pick), step=1, **st)
    assert torch.equal(st['it_next'].cpu().long(), pick)
    y = torch.softmax(torch.log_softmax(logits, 1), 1).gather(1, pick.view(-1, 1)).view(-1)
    v = (1 - y) + y
    unf = pick > 0
    np.testing.assert_allclose(st['stv'][:, 0].cpu()[unf].numpy(), v[unf].numpy(), atol=1.2e-7)


def test_finalize_len(ops):
    for flags, want in (([0] + [1] * 16, 16), ([0, 1, 1, 1, 0] + [0] * 12, 3), ([0, 0] + [0] * 15, 0)):
        L = dev(torch.zeros(1, dtype=torch.int32))
        ops.finalize_len(dev(torch.tensor(flags, dtype=torch.int32)), 16, L)
        assert int(L) == want


# ---- the gradient-product options of cic_gemm_f32: unordered sums (K-sliced tail tiles, K split over workgroups),
# ---- the bias-gradient by-product, pre-zeroed targets, row blocks of a decode pair
ORDER_FREE_SHAPES = [
    # M, N, K, a_kc, b_kc          (what the shape exercises)
    (2048, 512, 9488, True, False),   # 64 big tiles: every tile K-sliced, C zeroed by the launcher
    (9488, 512, 2048, False, False),  # 1192 small tiles: no tail
    (2560, 512, 2048, False, False),  # 80 big tiles, K-sliced + column sums
    (512, 512, 4608, False, False),   # 64 small tiles < 256: small tiles K-sliced
    (1000, 300, 1111, False, False),  # ragged everything (scalar loads), tail of partial tiles
    (300, 1000, 777, True, False),
    (128, 512, 3072, True, False),    # BPTT dX: K split over workgroups (register-streaming kernel)
    (96, 1024, 2048, True, False),
    (4700, 520, 600, False, True),    # 37 x 5 big tiles -> tail 185
]


@pytest.mark.parametrize('M,N,K,a_kc,b_kc', ORDER_FREE_SHAPES)
@pytest.mark.parametrize('accumulate', [False, True])
def test_gemm_order_free_and_colsum(ops, M, N, K, a_kc, b_kc, accumulate):
    import ctypes as C
    from cooperativeimagecaptioning_amd import _lib
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((M, K) if a_kc else (K, M), generator=g)
    B = torch.randn((N, K) if b_kc else (K, N), generator=g)
    C0 = torch.randn(M, N, generator=g)
    ref = (A if a_kc else A.t()).double() @ (B.t() if b_kc else B).double()
    if accumulate:
        ref = ref + C0.double()
    Ad, Bd, Cd = dev(A), dev(B), dev(C0.clone())
    ga = _lib.GemmArgs()
    ga.M, ga.N, ga.K = M, N, K
    ga.A, ga.lda, ga.a_kc = Ad.data_ptr(), A.shape[1], int(a_kc)
    ga.B, ga.ldb, ga.b_kc = Bd.data_ptr(), B.shape[1], int(b_kc)
    ga.C, ga.ldc, ga.accumulate, ga.sum_order_free = Cd.data_ptr(), N, int(accumulate), 1
    cs0 = torch.randn(M, generator=g)
    cs, cs2 = dev(cs0.clone()), dev(cs0.clone() * 2)
    if not a_kc:                              # dW = dY^T X: db += colsum(dY) rides along
        ga.colsum_A, ga.colsum_A2 = cs.data_ptr(), cs2.data_ptr()
    _lib.check(_lib.lib.cic_gemm_f32(C.byref(ga), None), 'cic_gemm_f32')
    torch.cuda.synchronize()
    tol = 3e-6 * np.sqrt(K) * 4
    np.testing.assert_allclose(Cd.cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=tol)
    if not a_kc:
        want = cs0.double() + A.double().sum(0)
        np.testing.assert_allclose(cs.cpu().double().numpy(), want.numpy(), rtol=1e-5, atol=tol)
        np.testing.assert_allclose(cs2.cpu().double().numpy(), (want + cs0.double()).numpy(), rtol=1e-5, atol=tol)


@pytest.mark.parametrize('N,K,K2,b_kc', [(512, 512, 0, True), (2560, 512, 512, True), (9488, 512, 0, True), (1024, 512, 0, True)])
def test_gemm_row_blocks_equal_two_products(ops, N, K, K2, b_kc):
    """rows_blk (a decode pair in one launch): rows [B, 2B) through the second pointer set, bit-identical to the two
    separate M = B products (the kernel choice depends on N and K only)."""
    import ctypes as C
    from cooperativeimagecaptioning_amd import _lib
    g = torch.Generator().manual_seed(N + K)
    Bh = 128
    A = [dev(torch.randn(Bh, K, generator=g)) for _ in range(2)]
    A2 = [dev(torch.randn(Bh, K2, generator=g)) for _ in range(2)] if K2 else [None, None]
    W = dev(torch.randn(N, K, generator=g) * 0.05)
    W2 = dev(torch.randn(N, K2, generator=g) * 0.05) if K2 else None
    bias = dev(torch.randn(N, generator=g))
    sep = [dev(torch.zeros(Bh, N)) for _ in range(2)]
    for i in range(2):
        ops.gemm(A[i], W, sep[i], True, True, bias=bias, A2=A2[i], B2=W2)
    both = [dev(torch.zeros(Bh, N)) for _ in range(2)]
    ga = _lib.GemmArgs()
    ga.M, ga.N, ga.K, ga.K2 = 2 * Bh, N, K, K2
    ga.A, ga.lda, ga.a_kc = A[0].data_ptr(), K, 1
    ga.B, ga.ldb, ga.b_kc = W.data_ptr(), K, 1
    if K2:
        ga.A2, ga.lda2, ga.B2, ga.ldb2 = A2[0].data_ptr(), K2, W2.data_ptr(), K2
        ga.A2_b = A2[1].data_ptr()
    ga.C, ga.ldc, ga.bias = both[0].data_ptr(), N, bias.data_ptr()
    ga.rows_blk, ga.A_b, ga.C_b = Bh, A[1].data_ptr(), both[1].data_ptr()
    _lib.check(_lib.lib.cic_gemm_f32(C.byref(ga), None), 'cic_gemm_f32')
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(sep[i], both[i])


@pytest.mark.parametrize('M', [37, 100, 200, 256])
def test_gemm_logit_walker_ragged_rows(ops, M):
    """The LDS-staged logit walker (K = 512, N = 9488) with row counts that are no multiple of its 16-row tiles /
    64-row groups: rows beyond M are clamped loads and never stored."""
    g = torch.Generator().manual_seed(M)
    A, W, bias = torch.randn(M, 512, generator=g), torch.randn(9488, 512, generator=g) * 0.05, torch.randn(9488, generator=g)
    guard = 7.0
    Cd = dev(torch.full((M + 3, 9488), guard))
    ops.gemm(dev(A), dev(W), Cd[:M], True, True, bias=dev(bias))
    torch.cuda.synchronize()
    ref = A.double() @ W.double().t() + bias.double()
    np.testing.assert_allclose(Cd[:M].cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=2e-4)
    assert float((Cd[M:] - guard).abs().max()) == 0.0


@pytest.mark.parametrize('pair', [False, True])
def test_gemm_column_split(ops, pair):
    """cic_gemm_args.n_split: [i2h(x) + h2h(h) | h2att(h)] in one launch.  The first 2560 columns equal the unsplit
    product bit for bit (same kernel, same per-row arithmetic); the tail columns are h W_h2att^T + b."""
    import ctypes as C
    from cooperativeimagecaptioning_amd import _lib
    g = torch.Generator().manual_seed(11)
    Bh, E, H, A_ = 128, 512, 512, 512
    nb = 2 if pair else 1
    x = [dev(torch.randn(Bh, E, generator=g)) for _ in range(nb)]
    h = [dev(torch.randn(Bh, H, generator=g)) for _ in range(nb)]
    W1, W2 = dev(torch.randn(5 * H, E, generator=g) * .05), dev(torch.randn(5 * H, H, generator=g) * .05)
    Wt, b1, bt = dev(torch.randn(A_, H, generator=g) * .05), dev(torch.randn(5 * H, generator=g)), dev(torch.randn(A_, generator=g))
    pre = [dev(torch.zeros(Bh, 5 * H)) for _ in range(nb)]
    att_h = [dev(torch.zeros(Bh, A_)) for _ in range(nb)]
    ga = _lib.GemmArgs()
    ga.M, ga.N, ga.K, ga.K2 = nb * Bh, 5 * H + A_, E, H
    ga.A, ga.lda, ga.a_kc = x[0].data_ptr(), E, 1
    ga.B, ga.ldb, ga.b_kc = W1.data_ptr(), E, 1
    ga.A2, ga.lda2, ga.B2, ga.ldb2 = h[0].data_ptr(), H, W2.data_ptr(), H
    ga.C, ga.ldc, ga.bias = pre[0].data_ptr(), 5 * H, b1.data_ptr()
    ga.n_split, ga.B2_tail, ga.ldb2_tail, ga.bias_tail = 5 * H, Wt.data_ptr(), H, bt.data_ptr()
    ga.C_tail, ga.ldc_tail = att_h[0].data_ptr(), A_
    if pair:
        ga.rows_blk, ga.A_b, ga.A2_b, ga.C_b = Bh, x[1].data_ptr(), h[1].data_ptr(), pre[1].data_ptr()
        ga.C_tail_b = att_h[1].data_ptr()
    assert _lib.lib.cic_gemm_split_ok(C.byref(ga)) == 1
    _lib.check(_lib.lib.cic_gemm_f32(C.byref(ga), None), 'cic_gemm_f32')
    torch.cuda.synchronize()
    for i in range(nb):
        plain = dev(torch.zeros(Bh, 5 * H))
        ops.gemm(x[i], W1, plain, True, True, bias=b1, A2=h[i], B2=W2)
        assert torch.equal(plain, pre[i])
        ref = h[i].double().cpu() @ Wt.double().cpu().t() + bt.double().cpu()
        np.testing.assert_allclose(att_h[i].cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    # unsupported forms are refused, not mis-computed
    ga.K2 = 0
    assert _lib.lib.cic_gemm_split_ok(C.byref(ga)) == 0
