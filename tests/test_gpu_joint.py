"""GPU: the whole joint step through the mirrored module API (models.AlternatingJointModel of
cooperativeimagecaptioning_amd) replaying the golden steps recorded from the reference: same
weights, same batch, same noise -> loss within 5e-5 relative, every parameter gradient within
5e-4 of its scale (fp32, different summation order)."""
import numpy as np
import pytest
import torch

import golden_util as GU

pytestmark = pytest.mark.gpu


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def decode_tags(cfg, turn, n):
    rr = cfg['retrieval_reward']
    names = []
    if turn == 'listener':
        names = ['sample']
    else:
        if cfg['caption_loss_weight'] > 0:
            names.append('mle')
        if cfg['retrieval_reward_weight'] > 0:
            names.append('sample')
            if rr == 'reinforce' and cfg['reinforce_baseline_type'] == 'greedy':
                names.append('greedy')
        if cfg['cider_optimization']:
            if rr in ('gumbel_softmax', 'multinomial_soft') or 'sample' not in names:
                names.append('cider_gen')        # gen_result_for_cider, AlternatingJointModel.py:378-389
            if 'greedy' not in names:
                names.append('greedy')
    assert len(names) == n, (names, n)
    return names


CASES = ['joint_gumbel', 'joint_gumbel_dropout', 'joint_gumbel_tau', 'joint_multinomial', 'joint_reinforce_gt',
         'joint_reinforce_greedy', 'joint_reinforce_no', 'joint_reinforce_listener', 'joint_gumbel_mle',
         'joint_plain_all', 'joint_gumbel_ps', 'joint_multinomial_ps', 'masked_joint_gumbel', 'bn_masked_joint_gumbel', 'fullwidth_joint_gumbel', 'fullwidth_plain_all', 'fullwidth_reinforce_listener', 'fullwidth_reinforce_speaker', 'fullsize_joint_gumbel', 'fullsize_mle', 'fullsize_reinforce_speaker',
         'fc_joint_reinforce_gt', 'fc_joint_reinforce_greedy']      # fc_*: the fc-feature speaker under REINFORCE / CIDEr


@pytest.mark.parametrize('name', CASES)
def test_joint_step_matches_reference(name):
    from cooperativeimagecaptioning_amd import models
    from cooperativeimagecaptioning_amd.misc import rewards
    z = GU.load_case(name)
    cfg = GU.cfg_dict(z)
    turn = str(z['turn'])
    B = z['fc'].shape[0]
    opt = GU.make_opt(cfg, B)
    rewards.init_scorer('corpus')
    model = models.AlternatingJointModel(opt)
    sd = {k: T_(v) for k, v in z['weights'].items()}
    model.load_state_dict(sd)
    bns = [m_ for m_ in model.modules() if isinstance(m_, torch.nn.BatchNorm1d)]
    for bn in bns:      # use_bn: the fixture's w.* buffers were read off AFTER the reference's step, which started from a fresh BatchNorm1d
        bn.reset_running_stats()
    model.cuda().train()
    tags = decode_tags(cfg, turn, int(z['n_decodes']))
    model.caption_generator.noise.override = {t: GU.noise_dict(z, f'noise{i}') for i, t in enumerate(tags)}
    fc, att = T_(z['fc']).cuda(), T_(z['att_raw']).cuda()
    labels, masks = T_(z['labels']).cuda(), T_(z['masks']).cuda()
    data = {'gts': GU.gts_list(z)}
    att_masks = T_(z['att_masks']).cuda() if 'att_masks' in z else None      # masked_*: ragged region counts
    model.zero_grad()
    if turn == 'None':
        loss = model(fc, labels, masks, data, att, att_masks)
    else:
        loss = model(fc, labels, masks, data, att, att_masks, is_alternating=True, alternating_turn=turn)
    loss.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.detach()), float(np.asarray(z['loss']).reshape(-1)[0]), rtol=5e-5, atol=1e-6)
    n = 0
    grads = {k: p.grad for k, p in model.named_parameters()}
    glob = max(float(np.abs(z[k][1]) / max(grads[k[5:]].numel(), 1)) for k in z if k.startswith('gdig.'))
    for k in z:
        if not k.startswith('gdig.'):
            continue
        g = grads[k[5:]]
        assert g is not None, k
        d = GU.digest(g.detach().cpu().numpy())
        ref = z[k]
        scale = abs(ref[1]) / max(g.numel(), 1)
        np.testing.assert_allclose(d[2:], ref[2:], rtol=5e-4, atol=5e-4 * scale + 1e-5 * glob, err_msg=k)
        np.testing.assert_allclose(d[1], ref[1], rtol=5e-4, atol=1e-5 * glob * g.numel(), err_msg=k + ' abs-sum')
        n += 1
    assert n > 0
    # parameters the reference left without gradient (frozen agent) must have none / zero here too
    for k, g in grads.items():
        if 'gdig.' + k not in z and g is not None:
            assert float(g.abs().max()) == 0.0, k
    if bns:                 # use_bn: the running statistics after the step - one update per decode, as the reference's att_embed calls
        sd_after = model.state_dict()
        for k in z:
            if k.startswith('after.'):
                np.testing.assert_allclose(sd_after[k[6:]].cpu().numpy(), z[k], rtol=1e-5, atol=1e-6, err_msg=k)
    if 'tokens0' in z:      # full-width case (BASELINE widths, recorded from the reference): token ids bit for bit
        for tag, key in (('sample', 'tokens0'), ('greedy', 'tokens1')):
            if key not in z:
                continue
            ref = z[key]
            got = model.last_decodes[tag].seq[:, :ref.shape[1]].cpu().numpy()
            np.testing.assert_array_equal(got, ref, err_msg=tag + ' tokens')
            assert int(model.last_decodes[tag].seq[:, ref.shape[1]:].abs().max().item() if
                       model.last_decodes[tag].seq.shape[1] > ref.shape[1] else 0) == 0
    # logged side-channel values (model.loss()) that the reference logged too
    logged = model.loss()
    for k in ('loss_cider', 'cider_greedy', 'avg_reward', 'retrieval_sc_loss', 'loss_cap'):
        if 'aux.' + k in z and k in logged:
            np.testing.assert_allclose(float(logged[k]), float(z['aux.' + k]), rtol=1e-4, atol=1e-6, err_msg=k)


def test_cpu_input_fails_loudly():
    from cooperativeimagecaptioning_amd import models, _lib
    z = GU.load_case('joint_gumbel')
    cfg = GU.cfg_dict(z)
    opt = GU.make_opt(cfg, z['fc'].shape[0])
    model = models.AlternatingJointModel(opt)
    with pytest.raises(_lib.CicError):
        model(T_(z['fc']), T_(z['labels']), T_(z['masks']), {'gts': GU.gts_list(z)}, T_(z['att_raw']), None,
              is_alternating=True, alternating_turn='speaker')


@pytest.mark.parametrize('name', ['mle_plain', 'mle_dropout', 'mle_ss', 'masked_mle'])
def test_mle_forward_backward_matches_reference(name):
    """Att2in2Model.forward (teacher-forced MLE, models/AttModel.py:103-148) vs the reference."""
    from cooperativeimagecaptioning_amd import models
    z = GU.load_case(name)
    cfg = GU.cfg_dict(z)
    B = z['fc'].shape[0]
    opt = GU.make_opt(cfg, B)
    cg = models.setup(opt, 'att2in2', 'caption_model')
    cg.load_state_dict({k: T_(v) for k, v in z['weights'].items()})
    cg.cuda().train()
    cg.noise.override = {'mle': GU.noise_dict(z, 'noise')}
    cg.ss_prob = float(z['ss_prob'])       # scheduled sampling (AttModel.py:118-129): recorded uniforms and draws
    cg.zero_grad()
    att_masks = T_(z['att_masks']).cuda() if 'att_masks' in z else None
    loss = cg(T_(z['fc']).cuda(), T_(z['att_raw']).cuda(), att_masks, T_(z['labels']).cuda(), T_(z['masks']).cuda())
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z['loss']), rtol=5e-5)
    grads = {k: p.grad for k, p in cg.named_parameters()}
    glob = max(float(np.abs(z[k][1]) / max(grads[k[5:]].numel(), 1)) for k in z if k.startswith('gdig.'))
    for k in z:
        if k.startswith('gdig.'):
            g = grads[k[5:]]
            d = GU.digest(g.detach().cpu().numpy())
            scale = abs(z[k][1]) / max(g.numel(), 1)
            np.testing.assert_allclose(d[2:], z[k][2:], rtol=5e-4, atol=5e-4 * scale + 1e-5 * glob, err_msg=k)


def test_use_bn_mle_forward_backward_and_running_stats_match_reference():
    """use_bn = 1: BatchNorm1d(att_feat_size) over the PACKED valid region rows in front of att_embed's Linear
    (models/AttModel.py:82-85, pack_wrapper :44-51), ragged region counts.  The product folds the normalisation into the
    Linear (csrc/batchnorm.hip); loss, every gradient (the norm's weight and bias included) and the running statistics
    after the forward pass are the reference's (fixture bn_masked_mle, recorded from the reference, tools/gen_golden.py)."""
    from cooperativeimagecaptioning_amd import models
    z = GU.load_case('bn_masked_mle')
    cfg = GU.cfg_dict(z)
    assert cfg['use_bn'] == 1
    B = z['fc'].shape[0]
    opt = GU.make_opt(cfg, B, use_bn=1)
    cg = models.setup(opt, 'att2in2', 'caption_model')
    sd = cg.state_dict()
    assert {'att_embed.0.running_mean', 'att_embed.0.running_var', 'att_embed.0.num_batches_tracked', 'att_embed.1.weight'} <= set(sd)
    cg.load_state_dict({k: T_(v) for k, v in z['weights'].items()})
    # the fixture's w.* buffers were read off the reference model AFTER its forward pass (tools/gen_golden.py: mle_case); the
    # reference model started from a fresh BatchNorm1d: running_mean 0, running_var 1, no batch tracked
    bn = cg.att_embed[0]
    bn.running_mean.zero_(), bn.running_var.fill_(1.0), bn.num_batches_tracked.zero_()
    cg.cuda().train()
    cg.noise.override = {'mle': GU.noise_dict(z, 'noise')}
    cg.ss_prob = float(z['ss_prob'])
    cg.zero_grad()
    args = (T_(z['fc']).cuda(), T_(z['att_raw']).cuda(), T_(z['att_masks']).cuda(), T_(z['labels']).cuda(), T_(z['masks']).cuda())
    loss = cg(*args)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z['loss']), rtol=5e-5)
    grads = {k: p.grad for k, p in cg.named_parameters()}
    glob = max(float(np.abs(z[k][1]) / max(grads[k[5:]].numel(), 1)) for k in z if k.startswith('gdig.'))
    for k in z:
        if k.startswith('gdig.'):
            g = grads[k[5:]]
            d = GU.digest(g.detach().cpu().numpy())
            scale = abs(z[k][1]) / max(g.numel(), 1)
            np.testing.assert_allclose(d[2:], z[k][2:], rtol=5e-4, atol=5e-4 * scale + 1e-5 * glob, err_msg=k)
    bn = cg.att_embed[0]
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), z['after.att_embed.0.running_mean'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), z['after.att_embed.0.running_var'], rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(z['after.att_embed.0.num_batches_tracked'])
    # evaluation mode: the running statistics normalise; identical to torch's own modules on the same rows
    cg.eval()
    with torch.no_grad():
        ev = cg(*args).item()
        att = args[1][args[2] > 0]
        y = torch.relu(cg.att_embed[1](cg.att_embed[0](att)))
    assert np.isfinite(ev) and abs(ev - loss.item()) > 0            # other statistics, no dropout: another value
    pre = cg.att_embed_pre(args[1], args[2])
    np.testing.assert_allclose(pre[args[2] > 0].cpu().numpy(), y.cpu().numpy(), rtol=2e-5, atol=2e-5)
    with pytest.raises(ValueError):
        cg.att_embed_pre(args[1], None)                               # the reference: a BatchNorm1d shape error


def _full_size_step(opt, turn, decodes, logged_exact=(), logged_close=(), loss_rtol=1e-4, grad_tol=1e-3, ragged=False,
                    eos_prob=0.07, want_L_below=None):
    """One step of the mirrored AlternatingJointModel on the GPU against the CPU oracle: same weights, batch, dropout
    masks and sampler noise.  decodes: {tag: 'u' (Gumbel uniforms) | 'pick' (injected multinomial draws) | None}.
    ragged: images have 20..36 valid regions (att_masks on both sides); eos_prob: how often an injected draw is <eos> (a high
    value makes every caption end early: L < seq_length); want_L_below: the sampled decode's length must be below it."""
    from cooperativeimagecaptioning_amd import models, synthetic
    from cooperativeimagecaptioning_amd.misc import rewards
    from oracle import joint as J
    rewards.init_scorer('corpus')
    B, T, V, H, E = opt.batch_size, opt.seq_length, opt.vocab_size, opt.rnn_size, opt.input_encoding_size
    torch.manual_seed(0)
    model = models.AlternatingJointModel(opt)
    model.caption_generator.logit.weight.data.mul_(3.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = synthetic.make_batch(opt, seed=77)
    g = torch.Generator().manual_seed(5)
    p = opt.drop_prob_lm

    def decode_noise(kind):
        d = dict(att_keep=(torch.rand(B, 36, H, generator=g) >= p).float().numpy(),
                 x_keep=(torch.rand(T + 1, B, E, generator=g) >= p).float().numpy(),
                 out_keep=(torch.rand(T + 1, B, H, generator=g) >= p).float().numpy())
        if kind == 'u':
            d['gumbel_u'] = torch.rand(T + 1, B, V + 1, generator=g).numpy()
        elif kind == 'pick':
            # injected draws: Zipf-like tokens, an EOS now and then so that the captions end at different lengths
            tok = (torch.rand(T + 1, B, generator=g) ** 4 * V).long() + 1
            eos = torch.rand(T + 1, B, generator=g) < eos_prob
            eos[:3] = False
            d['pick'] = torch.where(eos, torch.zeros_like(tok), tok).numpy()
        return d
    noise = {tag: decode_noise(kind) for tag, kind in decodes.items()}

    Ps = {k[len('caption_generator.'):]: v.clone().requires_grad_(True) for k, v in sd.items()
          if k.startswith('caption_generator.')}
    Pl = {k[len('vse.'):]: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith('vse.')}
    cfg = dict(vars(opt))
    tn = {t: {k: torch.from_numpy(v) for k, v in d.items()} for t, d in noise.items()}
    att_masks = None
    if ragged:
        nreg = torch.randint(20, 37, (B,), generator=g)
        att_masks = (torch.arange(36).unsqueeze(0) < nreg.unsqueeze(1)).float()
    ob = dict(fc_feats=batch['fc_feats'], att_feats=batch['att_feats'], att_masks=att_masks, labels=batch['labels'],
              masks=batch['masks'], gts=batch['gts'])
    ref_loss, aux = J.joint_forward(Ps, Pl, cfg, ob, tn, turn or 'speaker', turn is not None)
    ref_loss.backward()

    model.cuda().train()
    model.caption_generator.noise.override = noise
    model.zero_grad()
    args = (batch['fc_feats'].cuda(), batch['labels'].cuda(), batch['masks'].cuda(), batch, batch['att_feats'].cuda(),
            att_masks.cuda() if att_masks is not None else None)
    loss = model(*args) if turn is None else model(*args, is_alternating=True, alternating_turn=turn)
    loss.backward()
    torch.cuda.synchronize()
    # every decoded token id of the step equals the oracle's, directly (north_star: bit-exact greedy arg-max indices;
    # at V = 9487 near-ties between logits are likeliest)
    for tag, key in (('sample', 'gen_result'), ('greedy', 'greedy_res')):
        ref_tok = aux.get(key)
        if ref_tok is None or tag not in decodes:
            continue
        got = model.last_decodes[tag]
        L = int(got.L)
        assert L == ref_tok.shape[1], (tag, L, ref_tok.shape)
        if want_L_below is not None and tag == 'sample':
            assert L < want_L_below, (tag, L)
        np.testing.assert_array_equal(got.seq[:, :L].cpu().numpy(), ref_tok.numpy(), err_msg=tag + ' tokens')
        assert int((ref_tok > 0).sum()) > B, tag                  # real captions, not all-EOS
    np.testing.assert_allclose(float(loss.detach()), float(ref_loss.detach()), rtol=loss_rtol, atol=1e-6)
    logged = model.loss()
    for k in logged_exact:
        assert float(logged[k]) == pytest.approx(float(aux[k]), rel=1e-6, abs=1e-9), k     # f64 on both sides
    for k in logged_close:
        np.testing.assert_allclose(float(logged[k]), float(aux[k]), rtol=loss_rtol, atol=1e-6, err_msg=k)
    grads = {k: q.grad for k, q in model.named_parameters()}
    checked = 0
    for prefix, P in (('caption_generator.', Ps), ('vse.', Pl)):
        for k, v in P.items():
            got = grads[prefix + k]
            if v.grad is None or float(v.grad.abs().max()) == 0.0:
                assert got is None or float(got.abs().max()) == 0.0, k      # frozen agent / unused parameter
                continue
            if k.endswith('alpha_net.bias'):     # a softmax shift: its gradient is mathematically 0, rounding noise
                continue
            want = v.grad.double()
            err = float((got.detach().cpu().double() - want).norm() / want.norm())
            assert err < grad_tol, (k, err)
            checked += 1
    return checked


@pytest.mark.timeout(600)
def test_joint_step_full_size_matches_oracle():
    """BASELINE configs[2] at its full size (B = 128, 36 x 2048 regions, vocabulary 9487, 16 steps, dropout 0.5,
    ST-Gumbel + self-critical CIDEr-D): loss and logged terms within 1e-4 relative, every parameter gradient within
    1e-3 of its norm (fp32, different summation order; the CIDEr reward is integer n-gram work and agrees exactly, which
    it only does when every one of the 2 x 128 x 16 decoded tokens agrees)."""
    from cooperativeimagecaptioning_amd import synthetic
    n = _full_size_step(synthetic.default_opt(), 'speaker', {'sample': 'u', 'greedy': None},
                        logged_exact=('avg_reward', 'cider_greedy'), logged_close=('loss_cider',))
    assert n >= 20


@pytest.mark.timeout(900)
@pytest.mark.parametrize('turn', ['speaker', 'listener'])
def test_reinforce_step_full_size_matches_oracle(turn):
    """BASELINE configs[3] at its full size: REINFORCE with the ground-truth baseline + CIDEr-D, B = 256, both turns
    (multinomial draws injected as `pick` rows on both sides)."""
    from cooperativeimagecaptioning_amd import synthetic
    # run_joint.sh -o reinforce: the listener turn trains on the contrastive loss of the generated captions
    opt = synthetic.default_opt(batch_size=256, retrieval_reward='reinforce', reinforce_baseline_type='gt',
                                vse_loss_weight=1.0)
    decodes = {'sample': 'pick', 'greedy': None} if turn == 'speaker' else {'sample': 'pick'}
    n = _full_size_step(opt, turn, decodes,
                        logged_exact=('avg_reward', 'cider_greedy') if turn == 'speaker' else ())
    assert n >= (16 if turn == 'speaker' else 6)


@pytest.mark.timeout(600)
def test_mle_step_full_size_matches_oracle():
    """BASELINE configs[1] at its full size: att2in2 teacher-forced MLE, B = 64, dropout 0.5."""
    from cooperativeimagecaptioning_amd import synthetic
    opt = synthetic.default_opt(batch_size=64, caption_loss_weight=1.0, retrieval_reward_weight=0.0, cider_optimization=0,
                                is_alternating=0)
    n = _full_size_step(opt, None, {'mle': None}, logged_close=('loss_cap',))
    assert n >= 16


@pytest.mark.timeout(600)
def test_mle_step_bf16_variant_full_size_vs_f32_oracle():
    """BASELINE configs[1] in its reduced-precision variant (--compute_dtype bf16; the reference itself is f32 only, so
    the yardstick is the f32 oracle): bf16 operands in the batched products (f32 accumulation), the embedded regions and
    their projection stored in bf16, everything else f32.  Stated tolerance: loss within 2e-3 relative, every parameter
    gradient within 3e-2 of its norm (bf16 keeps 8 mantissa bits: 2^-9 = 2e-3 per rounded operand)."""
    from cooperativeimagecaptioning_amd import synthetic
    opt = synthetic.default_opt(batch_size=64, caption_loss_weight=1.0, retrieval_reward_weight=0.0, cider_optimization=0,
                                is_alternating=0, compute_dtype='bf16')
    n = _full_size_step(opt, None, {'mle': None}, logged_close=('loss_cap',), loss_rtol=2e-3, grad_tol=3e-2)
    assert n >= 16


# ---- the one-launch recurrences at their EDGE shapes, product build, against the ORACLE ------------
# tests/test_gpu_persistent.py compares the loops with the per-step launches of the development build; here the product
# library meets the CPU oracle at batch sizes that end in a partial 16-row strip (B = 100 = 6 strips + 4 rows) and that span
# two row blocks (B = 144 = a block of 8 strips + a block of one), with ragged region masks and with captions that all end
# early (the steps at or beyond L: BPTT, the GRU's longest caption per strip).  Tokens exact, loss 1e-4, every parameter
# gradient within 1e-3 of its norm.

@pytest.mark.timeout(900)
@pytest.mark.parametrize('B,ragged', [(100, True), (144, False)])
def test_joint_gumbel_step_at_edge_batch_sizes_matches_oracle(B, ragged):
    """listener GRU pass + its BPTT loop + speaker BPTT loop: partial last strip / two row blocks (B is not a multiple of 32
    or exceeds 128, so the sampled and the greedy decode run as two launch chains)."""
    from cooperativeimagecaptioning_amd import synthetic
    n = _full_size_step(synthetic.default_opt(batch_size=B), 'speaker', {'sample': 'u', 'greedy': None},
                        logged_exact=('avg_reward', 'cider_greedy'), logged_close=('loss_cider',), ragged=ragged)
    assert n >= 20


@pytest.mark.timeout(900)
@pytest.mark.parametrize('B,turn', [(100, 'speaker'), (144, 'listener')])
def test_reinforce_step_with_early_ending_captions_at_edge_batch_sizes_matches_oracle(B, turn):
    """REINFORCE (gt baseline) + CIDEr-D with injected draws that are <eos> 45 % of the time: every caption ends early
    (L < 16), so the BPTT loop skips steps at or beyond L and the GRU strips see different longest captions; ragged masks."""
    from cooperativeimagecaptioning_amd import synthetic
    opt = synthetic.default_opt(batch_size=B, retrieval_reward='reinforce', reinforce_baseline_type='gt', vse_loss_weight=1.0)
    decodes = {'sample': 'pick', 'greedy': None} if turn == 'speaker' else {'sample': 'pick'}
    n = _full_size_step(opt, turn, decodes, logged_exact=('avg_reward', 'cider_greedy') if turn == 'speaker' else (),
                        ragged=True, eos_prob=0.45, want_L_below=16)
    assert n >= (16 if turn == 'speaker' else 6)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('B,ragged', [(100, True), (144, False)])
def test_mle_step_at_edge_batch_sizes_matches_oracle(B, ragged):
    """the teacher-forced recurrence (spk_teacher_seq_kernel) and the BPTT loop behind it."""
    from cooperativeimagecaptioning_amd import synthetic
    opt = synthetic.default_opt(batch_size=B, caption_loss_weight=1.0, retrieval_reward_weight=0.0, cider_optimization=0,
                                is_alternating=0)
    n = _full_size_step(opt, None, {'mle': None}, logged_close=('loss_cap',), ragged=ragged)
    assert n >= 16


@pytest.mark.timeout(900)
@pytest.mark.parametrize('turn', ['speaker', 'listener'])
def test_joint_step_bf16_variant_full_size_vs_f32_oracle(turn):
    """The joint step in the reduced-precision variant (--compute_dtype bf16: bf16 operands in the batched products, bf16
    region features, and - r4 - the speaker's one-launch recurrences on bf16 MFMA fragments), against the f32 oracle at the
    tolerance of the MLE variant test for the loss (2e-3 relative) and every parameter gradient within 6e-2 of its norm
    (the table gradient, measured 3.8e-2 since the per-step logit / gate products read ONE bf16 image of their weights, sums
    every rounding of a 16-step recurrence; 3e-2 held while those products were still f32).  The
    configuration is the one whose token ids do not depend on the arithmetic: REINFORCE with the ground-truth baseline and
    INJECTED multinomial draws, CIDEr term off (a greedy or Gumbel arg-max over 9488 bf16-perturbed logits may legitimately pick
    another token than the f32 oracle, after which the two losses are losses of different captions)."""
    from cooperativeimagecaptioning_amd import synthetic
    opt = synthetic.default_opt(batch_size=128, retrieval_reward='reinforce', reinforce_baseline_type='gt', vse_loss_weight=1.0,
                                cider_optimization=0, retrieval_reward_weight=1.0, compute_dtype='bf16')
    n = _full_size_step(opt, turn, {'sample': 'pick'}, loss_rtol=2e-3, grad_tol=6e-2)
    assert n >= (14 if turn == 'speaker' else 6)
