"""CPU: the oracle (oracle/) replayed against the golden fixtures made from the
reference itself by tools/gen_golden.py.  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

import golden_util as GU
from oracle import speaker as S, listener as Lst, ciderd, joint as J

torch.set_num_threads(4)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def params(z, prefix='', grad=False):
    out = {}
    for k, v in z['weights'].items():
        if k.startswith(prefix):
            t = T(v).clone()
            if grad and t.is_floating_point() and 'running_' not in k:      # (buffers of a BatchNorm layer stay plain tensors)
                t.requires_grad_(True)
            out[k[len(prefix):]] = t
    return out


def masks_t(z):
    """att_masks of a ragged-region fixture (masked_*), else None."""
    return T(z['att_masks']) if 'att_masks' in z else None


def noise_t(z, prefix):
    nd = GU.noise_dict(z, prefix)
    return {k: T(v) for k, v in nd.items()} or None


def close(a, b, rtol=2e-5, atol=2e-6):
    if torch.is_tensor(a):
        a = a.detach().numpy()
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64),
                               rtol=rtol, atol=atol)


def check_digest(grad, dig, name, rtol=2e-4):
    d = GU.digest(grad.detach().numpy())
    scale = max(abs(dig[1]) / max(grad.numel(), 1), 1e-8)       # mean |g|
    # 1e-8 absolute slack: e.g. d(loss)/d(alpha_net.bias) is exactly 0 in exact arithmetic
    # (softmax shift invariance) and pure rounding noise (~1e-10) in fp32
    np.testing.assert_allclose(d[2:], dig[2:], rtol=rtol, atol=rtol * scale + 1e-8, err_msg=name)
    np.testing.assert_allclose(d[1], dig[1], rtol=rtol, atol=1e-8 * grad.numel(), err_msg=name + ' abs-sum')


def test_kernels_speaker():
    z = GU.load_case('kernels_speaker')
    P = params(z)
    cfg = GU.cfg_dict(z)
    att = S.att_embed(P, T(z['att_raw']), None, 0.0)
    p_att = S.ctx2att(P, att)
    close(att, z['att'])
    close(p_att, z['p_att'])
    h, c, xt = T(z['h']), T(z['c']), T(z['xt'])
    att_res, alpha = S.attention_step(P, h, att, p_att, None)
    close(att_res, z['att_res'])
    att_res_m, _ = S.attention_step(P, h, att, p_att, T(z['att_masks']))
    close(att_res_m, z['att_res_masked'])
    out, h2, c2, _ = S.core_step(P, xt, att, p_att, None, h, c, None, 0.0)
    close(h2, z['h2'])
    close(c2, z['c2'])
    close(out, z['out'])
    logp, _ = S.logprobs_from_output(P, out)
    close(logp, z['logp'])
    assert cfg['vocab_size'] == 97


SAMPLE_CASES = ['sample_greedy_full', 'sample_greedy_early', 'sample_greedy_dropout',
                'sample_multinomial_plain', 'sample_multinomial_temp', 'sample_gumbel_st',
                'sample_gumbel_st_tau', 'sample_multinomial_st', 'sample_gumbel_ps',
                'sample_multinomial_ps', 'sample_multinomial_ps_tau', 'sample_constraint',
                'masked_sample_greedy', 'masked_sample_gumbel_st']


@pytest.mark.parametrize('name', SAMPLE_CASES)
def test_sample(name):
    z = GU.load_case(name)
    P = params(z)
    cfg = GU.cfg_dict(z)
    opt = {k[4:]: (int(v) if k[4:] != 'temperature' else float(v)) for k, v in z.items()
           if k.startswith('opt.')}
    res = S.sample(P, cfg, T(z['fc']), T(z['att_raw']), masks_t(z), opt, noise_t(z, 'noise'),
                   cfg['retrieval_reward'])
    assert res[0].shape == z['res0'].shape, 'L differs'
    np.testing.assert_array_equal(res[0].numpy(), z['res0'])          # token ids: exact
    for i in range(1, len(res)):
        close(res[i], z[f'res{i}'], rtol=5e-5, atol=5e-6)


@pytest.mark.parametrize('name', ['mle_plain', 'mle_dropout', 'mle_ss', 'masked_mle', 'bn_masked_mle'])
def test_mle(name):
    z = GU.load_case(name)
    P = params(z, grad=True)
    cfg = GU.cfg_dict(z)
    loss = S.mle_forward(P, cfg, T(z['fc']), T(z['att_raw']), masks_t(z), T(z['labels']), T(z['masks']),
                         noise_t(z, 'noise'), float(z['ss_prob']))
    close(loss, z['loss'])
    loss.backward()
    for k, p in P.items():
        if 'gdig.' + k in z:
            check_digest(p.grad, z['gdig.' + k], k)


def test_listener():
    z = GU.load_case('listener')
    P = params(z, grad=True)
    cfg = GU.cfg_dict(z)
    fc, labels, masks = T(z['fc']), T(z['labels']), T(z['masks'])
    close(Lst.encode_image(P, fc), z['img_emb'])
    close(Lst.encode_text(P, labels, masks), z['cap_emb'])
    V = cfg['vocab_size']
    onehot = torch.zeros(labels.shape[0], labels.shape[1], V + 2).scatter_(2, labels.unsqueeze(2), 1.0)
    close(Lst.encode_text(P, onehot, masks), z['cap_emb_onehot'])
    for wb in (False, True):
        for oor in ('off', 'image', 'caption'):
            close(Lst.vse_forward(P, cfg, fc, labels, masks, wb, oor), z[f'loss_wb{int(wb)}_{oor}'])
    soft = T(z['soft']).clone().requires_grad_(True)
    loss = Lst.vse_forward(P, cfg, fc, soft, masks)
    close(loss, z['loss_soft'])
    loss.backward()
    close(soft.grad, z['grad_soft'], rtol=2e-4, atol=1e-8)
    for k, p in P.items():
        check_digest(p.grad, z['gdig.' + k], k)


@pytest.mark.parametrize('name', ['listener_mean', 'listener_max'])
def test_listener_pool(name):
    z = GU.load_case(name)
    P = params(z)
    cfg = GU.cfg_dict(z)
    close(Lst.vse_forward(P, cfg, T(z['fc']), T(z['labels']), T(z['masks'])), z['loss'])


@pytest.mark.parametrize('name', ['ciderd', 'ciderd_spi2'])
def test_ciderd(name):
    z = GU.load_case(name)
    gts = GU.gts_list(z)
    reward, cg = ciderd.get_self_critical_reward(gts, z['gen'], z['greedy'])
    np.testing.assert_allclose(reward, z['reward'], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(cg, z['cider_greedy'], rtol=1e-12)
    if 'cider_gen' in z:
        gen, _, _ = ciderd.get_self_critical_reward(gts, z['gen'], z['greedy'], True)
        np.testing.assert_allclose(gen, z['cider_gen'], rtol=1e-12, atol=1e-14)


def test_ciderd_counts_quirks():
    # R1: the first 0 is kept as a token; tokens after it are dropped
    assert ciderd.row_to_tokens([5, 0, 7]) == [5, 0]
    assert ciderd.row_to_tokens([5, 6, 7]) == [5, 6, 7]
    c = ciderd.precook([4, 5, 4, 5, 0])
    assert c[(4,)] == 2 and c[(4, 5)] == 2 and c[(5, 0)] == 1 and c[(4, 5, 4, 5)] == 1
    assert sum(v for k, v in c.items() if len(k) == 2) == 4       # "length" counts bigrams


JOINT_CASES = ['joint_gumbel', 'joint_gumbel_dropout', 'joint_gumbel_tau', 'joint_multinomial',
               'joint_gumbel_ps', 'joint_multinomial_ps', 'joint_reinforce_gt',
               'joint_reinforce_greedy', 'joint_reinforce_no', 'joint_reinforce_listener',
               'joint_gumbel_mle', 'joint_plain_all', 'masked_joint_gumbel', 'bn_masked_joint_gumbel', 'fullwidth_joint_gumbel', 'fullwidth_plain_all', 'fullwidth_reinforce_listener', 'fullwidth_reinforce_speaker', 'fullsize_joint_gumbel', 'fullsize_mle', 'fullsize_reinforce_speaker',
               'fc_joint_reinforce_gt', 'fc_joint_reinforce_greedy']   # fc_*: the fc-feature speaker under REINFORCE / CIDEr


def joint_noise(z, cfg, turn):
    """Map the recorded decode order onto the oracle's named decodes."""
    n = int(z['n_decodes'])
    decs = [noise_t(z, f'noise{i}') for i in range(n)]
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
                names.append('greedy_baseline')
        if cfg['cider_optimization']:
            if rr in ('gumbel_softmax', 'multinomial_soft'):
                names.append('cider_gen')
            if 'greedy_baseline' not in names:
                names.append('greedy')
    assert len(names) == n, (names, n)
    out = dict(zip(names, decs))
    if 'greedy_baseline' in out:
        out['greedy'] = out['greedy_baseline']
    return out


@pytest.mark.parametrize('name', JOINT_CASES)
def test_joint(name):
    z = GU.load_case(name)
    cfg = GU.cfg_dict(z)
    Ps = params(z, 'caption_generator.', grad=True)
    Pl = params(z, 'vse.', grad=True)
    turn = str(z['turn'])
    batch = dict(fc_feats=T(z['fc']), att_feats=T(z['att_raw']), att_masks=masks_t(z), labels=T(z['labels']),
                 masks=T(z['masks']), gts=GU.gts_list(z))
    noise = joint_noise(z, cfg, turn)
    if turn == 'None':
        loss, aux = J.joint_forward(Ps, Pl, cfg, batch, noise, None, is_alternating=False)
    else:
        loss, aux = J.joint_forward(Ps, Pl, cfg, batch, noise, turn, is_alternating=True)
    close(loss, z['loss'], rtol=5e-5)
    if 'tokens0' in z:              # full-width case: the reference's decoded token ids (sampled, greedy), bit for bit
        np.testing.assert_array_equal(aux['gen_result'].numpy(), z['tokens0'])
        if 'tokens1' in z:
            np.testing.assert_array_equal(aux['greedy_res'].numpy(), z['tokens1'])
    loss.backward()
    freeze_l = cfg['retrieval_reward'] == 'reinforce' and turn == 'speaker'
    freeze_s = turn == 'listener'
    n = 0
    for pre, P, frozen in (('caption_generator.', Ps, freeze_s), ('vse.', Pl, freeze_l)):
        for k, p in P.items():
            key = 'gdig.' + pre + k
            if frozen or key not in z:
                continue
            assert p.grad is not None, key
            check_digest(p.grad, z[key], key, rtol=5e-4)
            n += 1
    assert n > 0


def test_clamp_adam():
    z = GU.load_case('clamp_adam')
    p = {'p': T(z['p0']).clone()}
    st = {}
    for i in range(z['grads'].shape[0]):
        J.clamp_adam_step(p, {'p': T(z['grads'][i])}, st, float(z['lr']), float(z['grad_clip']))
        close(p['p'], z['traj'][i], rtol=1e-6, atol=1e-7)


# ---- FCModel (BASELINE configs[0], the reference's CPU plumbing case) ------------------------------------
@pytest.mark.parametrize('name', ['fc_mle', 'fc_mle_dropout'])
def test_fc_mle_matches_reference(name):
    from oracle import fc as FC
    z = GU.load_case(name)
    cfg = GU.cfg_dict(z)
    P = params(z, grad=True)
    loss = FC.fc_forward(P, cfg, T(z['fc']), T(z['labels']), T(z['masks']), noise_t(z, 'noise'))
    close(loss, z['loss'][0])
    loss.backward()
    for k, p in P.items():
        check_digest(p.grad, z['gdig.' + k], k)


@pytest.mark.parametrize('name', ['fc_sample_greedy', 'fc_sample_greedy_dropout', 'fc_sample_multinomial',
                                  'fc_sample_multinomial_temp'])
def test_fc_sample_matches_reference(name):
    from oracle import fc as FC
    z = GU.load_case(name)
    cfg = GU.cfg_dict(z)
    P = params(z)
    opt = {'sample_max': int(z['opt.sample_max']), 'temperature': float(z['opt.temperature'])}
    with torch.no_grad():
        seq, slp = FC.fc_sample(P, cfg, T(z['fc']), opt, noise_t(z, 'noise'))
    np.testing.assert_array_equal(seq.numpy(), z['res0'])
    close(slp, z['res1'])


# ---- AttModel.sample_beam (the evaluation decode) -------------------------------------------
@pytest.mark.parametrize('name', ['beam2', 'beam3_early', 'beam5_constraint', 'masked_beam3'])
def test_sample_beam_matches_reference(name):
    z = GU.load_case(name)
    cfg = GU.cfg_dict(z)
    P = params(z)
    with torch.no_grad():
        seq, lps, score = S.sample_beam(P, cfg, T(z['fc']), T(z['att_raw']), masks_t(z),
                                        {'beam_size': int(z['beam']), 'decoding_constraint': cfg['decoding_constraint']})
    np.testing.assert_array_equal(seq.numpy(), z['res0'])
    close(lps, z['res1'])
    close(score, z['score'], rtol=1e-5)


# ---- retrieval-rank evaluation (eval_utils.i2t / t2i) ---------------------------------------
@pytest.mark.parametrize('name', ['retrieval_5cap', 'retrieval_gen_1cap'])
def test_retrieval_ranks_match_reference(name):
    from oracle import retrieval as R
    z = dict(np.load(os.path.join(GU.GOLDEN, name + '.npz')))
    cpi = int(z['cpi'])
    if cpi == 5:
        r, (ranks, top1) = R.i2t(z['images'], z['captions'])
        np.testing.assert_array_equal(ranks, z['i2t_ranks'])
        np.testing.assert_array_equal(top1, z['i2t_top1'])
        np.testing.assert_allclose(np.array(r), z['i2t_r'])
    ri, (ranks_i, top1_i) = R.t2i(z['images'], z['captions'], cpi)
    np.testing.assert_array_equal(ranks_i, z['t2i_ranks'])
    np.testing.assert_array_equal(top1_i, z['t2i_top1'])
    np.testing.assert_allclose(np.array(ri), z['t2i_r'])


# ---- share_embed = 1: ONE embedding table owned by both agents and both Adam instances (AlternatingJointModel.py:83-88) --------
import share_util as SU  # noqa: E402


@pytest.mark.parametrize('name', SU.SHARE_CASES)
def test_share_embed_trajectory_matches_reference(name):
    """Several iterations of the reference's trainer (zeroing, forward, backward, clamp + Adam for the agents of the turn)
    with the shared table: per iteration the loss, the decoded tokens, every gradient digest and the digest of EVERY weight
    after the update - the table's included, which pins who moves it, with which moments, from which gradient."""
    z, cfg, w0 = SU.load(name)
    Ps = {k[len('caption_generator.'):]: T(v).clone().requires_grad_(True) for k, v in w0.items() if k.startswith('caption_generator.')}
    Pl = {k[len('vse.'):]: T(v).clone().requires_grad_(True) for k, v in w0.items() if k.startswith('vse.')}
    Pl['txt_enc.embed.weight'] = Ps['embed.0.weight']                 # the same tensor object: the shared table
    st_s, st_l = {}, {}
    for s in range(int(z['n_steps'])):
        d = SU.step_view(z, s)
        turn = d['turn']
        noise = {k: {kk: T(vv) for kk, vv in v.items()} for k, v in SU.step_noise(d, cfg, turn).items()}
        b = SU.step_batch(d)
        batch = dict(fc_feats=T(b['fc_feats']), att_feats=T(b['att_feats']), att_masks=None, labels=T(b['labels']),
                     masks=T(b['masks']), gts=b['gts'])
        loss, aux = J.train_step(Ps, Pl, cfg, batch, noise, turn, st_s, st_l, cfg['learning_rate'], cfg['grad_clip'])
        close(loss, d['loss'], rtol=1e-4)
        np.testing.assert_array_equal(aux['gen_result'].numpy(), d['tokens0'])
        assert bool(Ps['embed.0.weight'].requires_grad) == bool(int(d['embed_requires_grad']))
        n = 0
        for pre, P in (('caption_generator.', Ps), ('vse.', Pl)):
            for k, p in P.items():
                key = 'gdig.' + pre + k
                # (the agent that does not step in a reinforce turn keeps the previous turn's gradients, clamped in place by
                # clip_gradient: stale values nobody reads)
                frozen = cfg['retrieval_reward'] == 'reinforce' and (pre == 'vse.') == (turn == 'speaker')
                if key in d and float(np.abs(d[key][1])) > 0 and not frozen:
                    check_digest(p.grad, d[key], f'step {s} {key}', rtol=5e-4)
                    n += 1
                if k.endswith('alpha_net.bias'):     # a softmax shift: its gradient is rounding noise, and Adam turns noise into +-lr
                    continue
                wk = 'wdig.' + pre + k
                got = GU.digest(p.detach().numpy())
                np.testing.assert_allclose(got, d[wk], rtol=2e-5, atol=2e-6, err_msg=f'step {s} {wk}')
        assert n > 0
