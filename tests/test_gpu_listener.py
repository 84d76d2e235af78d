"""GPU: listener engine (image/text encoders, contrastive loss, backward) against the golden
listener fixture from the reference and against the oracle's autograd."""
import numpy as np
import pytest
import torch

import golden_util as GU

pytestmark = pytest.mark.gpu


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def setup(z, Lp, T=16, pool='last'):
    from cooperativeimagecaptioning_amd import engine
    cfg = GU.cfg_dict(z)
    W = {k: T_(v).cuda().contiguous() for k, v in z['weights'].items()}
    J, F = W['img_enc.fc.weight'].shape
    E = W['txt_enc.embed.weight'].shape[1]
    B = z['fc'].shape[0]
    dims = engine.listener_dims(B, F, E, J, cfg['vocab_size'], T, Lp, cfg['vse_margin'], cfg['vse_max_violation'],
                                cfg['vse_no_imgnorm'], cfg['vse_use_abs'], pool=pool)
    return engine, cfg, W, dims, engine.listener_params(W)


def test_listener_labels_golden_and_grads():
    from oracle import listener as Lst
    z = GU.load_case('listener')
    labels, masks, fc = T_(z['labels']), T_(z['masks']), T_(z['fc'])
    engine, cfg, W, dims, params = setup(z, labels.shape[1])
    for oor in ('off', 'image', 'caption'):
        f = engine.listener_fwd(dims, params, fc.cuda(), labels=labels.cuda(), masks=masks.cuda(),
                                only_one_retrieval=oor, want_emb=True)
        np.testing.assert_allclose(f['loss_sum'].cpu().numpy()[0], z[f'loss_wb0_{oor}'], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(f['loss_rows'].cpu().numpy(), z[f'loss_wb1_{oor}'], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(f['img_emb'].cpu().numpy(), z['img_emb'], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(f['cap_emb'].cpu().numpy(), z['cap_emb'], rtol=2e-5, atol=2e-6)
    # gradients of a weighted per-row loss vs oracle autograd
    P = {k: T_(v).clone().requires_grad_(True) for k, v in z['weights'].items()}
    g = torch.Generator().manual_seed(0)
    gr = torch.rand(labels.shape[0], generator=g) + 0.5
    (Lst.vse_forward(P, cfg, fc, labels, masks, True, 'off') * gr).sum().backward()
    f = engine.listener_fwd(dims, params, fc.cuda(), labels=labels.cuda(), masks=masks.cuda())
    grads = {k: torch.zeros_like(v) for k, v in W.items()}
    engine.listener_bwd(dims, params, f, g_rows=gr.cuda(), grads=grads)
    for k in grads:
        ref = P[k].grad.numpy()
        np.testing.assert_allclose(grads[k].cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * np.abs(ref).mean() + 1e-8, err_msg=k)


@pytest.mark.parametrize('B,J,E,F,V,maxv', [(6, 128, 64, 96, 97, 1), (128, 1024, 512, 2048, 9487, 1), (9, 64, 32, 40, 50, 0)])
def test_listener_generated_vs_oracle(B, J, E, F, V, maxv):
    """ST one-hot captions: forward = scaled gather, backward = dense d one_hot (VSEFCModel.py:102-104)."""
    from cooperativeimagecaptioning_amd import engine
    from oracle import listener as Lst
    g = torch.Generator().manual_seed(B + J)
    T = 16
    W = {'img_enc.fc.weight': torch.randn(J, F, generator=g) * 0.05, 'img_enc.fc.bias': torch.randn(J, generator=g) * .1,
         'txt_enc.embed.weight': (torch.rand(V + 2, E, generator=g) - .5) * .2,
         'txt_enc.rnn.weight_ih_l0': (torch.rand(3 * J, E, generator=g) - .5) * .2,
         'txt_enc.rnn.weight_hh_l0': (torch.rand(3 * J, J, generator=g) - .5) * .1,
         'txt_enc.rnn.bias_ih_l0': (torch.rand(3 * J, generator=g) - .5) * .1,
         'txt_enc.rnn.bias_hh_l0': (torch.rand(3 * J, generator=g) - .5) * .1}
    fc = torch.randn(B, F, generator=g).abs()
    L = 11
    seq = torch.randint(1, V + 1, (B, T), generator=g)
    for b in range(B):                       # ragged: EOS at a random place, zeros after
        e = int(torch.randint(1, L + 3, (1,), generator=g))
        seq[b, e:] = 0
    seq[0, :] = 0                            # EOS first
    seq[1, :L] = torch.randint(1, V + 1, (L,), generator=g)   # never finishes inside L
    seq[:, L:] = torch.randint(0, V + 1, (B, T - L), generator=g)   # garbage past L must be ignored
    stv = 1.0 + (torch.rand(B, T, generator=g) - 0.5) * 2e-7
    stv = torch.where(seq > 0, stv, torch.ones_like(stv))
    # oracle: dense one-hot rows [B, L+1, V+2], masks [1,1,(seq>0)[:, :L-1]]
    P = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    sq = seq[:, :L]
    oh = torch.zeros(B, L + 1, V + 2)
    oh[:, 0, V + 1] = 1
    oh[:, 1:, :].scatter_(2, sq.unsqueeze(2), stv[:, :L].unsqueeze(2))
    oh.requires_grad_(True)
    masks = torch.cat([torch.ones(B, 2), (sq > 0).float()[:, :-1]], 1)
    cfg = dict(vse_margin=0.2, vse_max_violation=maxv)
    loss = Lst.vse_forward(P, cfg, fc, oh, masks, False, 'off')
    loss.backward()
    dims = engine.listener_dims(B, F, E, J, V, T, T + 1, 0.2, maxv)
    Wd = {k: v.cuda().contiguous() for k, v in W.items()}
    params = engine.listener_params(Wd)
    f = engine.listener_fwd(dims, params, fc.cuda(), seq=seq.int().cuda(), stv=stv.cuda(),
                            L=torch.tensor([L], dtype=torch.int32).cuda())
    np.testing.assert_allclose(float(f['loss_sum']), float(loss), rtol=5e-5)
    grads = {k: torch.zeros_like(v) for k, v in Wd.items()}
    d_onehot = torch.zeros(T, B, V + 1).cuda()
    engine.listener_bwd(dims, params, f, g_scalar=torch.ones(1).cuda(), grads=grads, d_onehot=d_onehot)
    for k in grads:
        ref = P[k].grad.numpy()
        np.testing.assert_allclose(grads[k].cpu().numpy(), ref, rtol=3e-4, atol=3e-4 * np.abs(ref).mean() + 1e-8, err_msg=k)
    ref = oh.grad[:, 1:, :V + 1].permute(1, 0, 2).numpy()          # [L, B, V+1]
    got = d_onehot[:L].cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=3e-4, atol=3e-4 * np.abs(ref).mean() + 1e-9)


@pytest.mark.parametrize('name,pool', [('listener_mean', 'mean'), ('listener_max', 'max')])
def test_listener_pooling_modes(name, pool):
    """vse_pool_type 'mean' / 'max' (VSEFCModel.py:118-127): loss vs the value recorded from the reference, every
    parameter gradient vs the oracle's autograd."""
    from oracle import listener as Lst
    z = GU.load_case(name)
    labels, masks, fc = T_(z['labels']), T_(z['masks']), T_(z['fc'])
    engine, cfg, W, dims, params = setup(z, labels.shape[1], pool=pool)
    assert cfg['vse_pool_type'] == pool
    f = engine.listener_fwd(dims, params, fc.cuda(), labels=labels.cuda(), masks=masks.cuda())
    np.testing.assert_allclose(f['loss_sum'].cpu().numpy()[0], float(z['loss']), rtol=2e-5, atol=1e-6)
    P = {k: T_(v).clone().requires_grad_(True) for k, v in z['weights'].items()}
    g = torch.Generator().manual_seed(1)
    gr = torch.rand(labels.shape[0], generator=g) + 0.5
    (Lst.vse_forward(P, cfg, fc, labels, masks, True, 'off') * gr).sum().backward()
    grads = {k: torch.zeros_like(v) for k, v in W.items()}
    engine.listener_bwd(dims, params, f, g_rows=gr.cuda(), grads=grads)
    for k in grads:
        ref = P[k].grad.numpy()
        np.testing.assert_allclose(grads[k].cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * np.abs(ref).mean() + 1e-8, err_msg=k)


def test_listener_bf16_variant_full_size_vs_f32():
    """(r4) cic_listener_dims.compute_dtype = bf16 at the flagship widths: the GRU pass and its BPTT loop on bf16 MFMA fragments
    (gru_seq_kernel<8, 8, true>, gru_seq_bwd_kernel<8, true>), the text encoder's batched products on one bf16 part; image
    encoder, similarities and loss f32.  Against the f32 engine on the same inputs (ragged generated captions): loss 2e-3,
    every parameter gradient and the straight-through gradient within 3e-2 of its norm - the variant's tolerance."""
    from cooperativeimagecaptioning_amd import engine, status
    B, J, E, F, V, T = 128, 1024, 512, 2048, 9487, 16
    g = torch.Generator().manual_seed(77)
    r = 2.0 / np.sqrt(J)                                # nn.GRU's own initialisation: U(-1/sqrt(J), 1/sqrt(J))
    W = {'img_enc.fc.weight': torch.randn(J, F, generator=g) * 0.05, 'img_enc.fc.bias': torch.randn(J, generator=g) * .1,
         'txt_enc.embed.weight': (torch.rand(V + 2, E, generator=g) - .5) * .2,
         'txt_enc.rnn.weight_ih_l0': (torch.rand(3 * J, E, generator=g) - .5) * r,
         'txt_enc.rnn.weight_hh_l0': (torch.rand(3 * J, J, generator=g) - .5) * r,
         'txt_enc.rnn.bias_ih_l0': (torch.rand(3 * J, generator=g) - .5) * r,
         'txt_enc.rnn.bias_hh_l0': (torch.rand(3 * J, generator=g) - .5) * r}
    Wd = {k: v.cuda().contiguous() for k, v in W.items()}
    params = engine.listener_params(Wd)
    fc = torch.randn(B, F, generator=g).abs().cuda()
    seq = torch.randint(1, V + 1, (B, T), generator=g)
    for b in range(B):
        seq[b, int(torch.randint(2, T + 1, (1,), generator=g)):] = 0
    seq = seq.int().cuda()
    stv = torch.ones(B, T).cuda()
    L = torch.tensor([T], dtype=torch.int32).cuda()
    out = {}
    for dt in ('f32', 'bf16'):
        dims = engine.listener_dims(B, F, E, J, V, T, T + 1, 0.2, 1, compute_dtype=dt)
        f = engine.listener_fwd(dims, params, fc, seq=seq, stv=stv, L=L)
        grads = {k: torch.zeros_like(v) for k, v in Wd.items()}
        d_onehot = torch.zeros(T, B, V + 1).cuda()
        engine.listener_bwd(dims, params, f, g_scalar=torch.ones(1).cuda(), grads=grads, d_onehot=d_onehot)
        torch.cuda.synchronize()
        out[dt] = (float(f['loss_sum']), {k: v.double().cpu() for k, v in grads.items()}, d_onehot.double().cpu())
    status.check(None, 'bf16 listener')
    l32, g32, d32 = out['f32']
    l16, g16, d16 = out['bf16']
    assert l16 != l32                                   # another arithmetic did run
    np.testing.assert_allclose(l16, l32, rtol=2e-3)
    errs = {k: float((g16[k] - g32[k]).norm() / g32[k].norm()) for k in g32}
    errs['d_onehot'] = float((d16 - d32).norm() / d32.norm())
    print('bf16 listener: loss', l16, l32, 'gradient errors', {k: round(v, 4) for k, v in errs.items()})
    assert max(errs.values()) < 3e-2, errs
