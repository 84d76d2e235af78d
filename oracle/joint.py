"""Oracle (test infrastructure): loss composition of the joint step + clamp/Adam.

Follows /root/reference/models/AlternatingJointModel.py:196-555 (forward and its
helpers), misc/utils.py:65-69 (clip_gradient = elementwise clamp) and
optimizer.py:25-27 (Adam with torch defaults).  Functional style: parameters are
dicts ``Ps`` (speaker, keys as in oracle/speaker.py) and ``Pl`` (listener, keys as
in oracle/listener.py); gradients come from torch autograd on the CPU.

``noise`` is a dict of per-decode noise dicts (see oracle/speaker.py):
  'sample'  the sampled decode of the DISC term        (:228, :346, :539)
  'greedy'  the greedy decode of the CIDEr baseline    (:391-403)
  'greedy_baseline'  the greedy decode of the REINFORCE baseline (:250-266)
  'cider_gen'        the extra sampled decode of PS modes (:378-389)
  'mle'     the teacher-forced pass                    (:196-207)
"""
import numpy as np
import torch

from . import speaker as _att_speaker
from . import fc as FC
from . import listener as Lst
from . import ciderd


class _Speaker:
    """The caption generator of the step: att2in2 (oracle/speaker.py) or, with cfg['caption_model'] == 'fc', the
    fc-feature FCModel (oracle/fc.py), whose sample() returns (seq, logprobs) only - the reference can drive it with the
    MLE, REINFORCE and CIDEr terms (reinforce_disc :226-247 unpacks two values), not with the straight-through ones."""

    def __init__(self, cfg):
        self.fc = cfg.get('caption_model', 'att2in2') == 'fc'

    def sample(self, P, cfg, fc, att, att_masks, opt, noise, rr):
        if not self.fc:
            return _att_speaker.sample(P, cfg, fc, att, att_masks, opt, noise, rr)
        if opt.get('use_one_hot'):
            raise ValueError('FCModel.sample returns two values: the straight-through modes cannot run on it')
        return FC.fc_sample(P, cfg, fc, opt, noise)

    def mle_forward(self, P, cfg, fc, att, att_masks, seq, masks, noise, ss_prob):
        if not self.fc:
            return _att_speaker.mle_forward(P, cfg, fc, att, att_masks, seq, masks, noise, ss_prob)
        return FC.fc_forward(P, cfg, fc, seq, masks, noise)


def gen_masks_from(word_index):
    """[1, 1, (w>0)[:, :-1]]  — AlternatingJointModel.py:232-234,353-355,385-387,542-544."""
    B = word_index.shape[0]
    return torch.cat([torch.ones(B, 2), (word_index > 0).float()[:, :-1]], 1)


def joint_forward(Ps, Pl, cfg, batch, noise=None, turn='speaker', is_alternating=True):
    """AlternatingJointModel.forward (:433-555).  batch: dict(fc_feats, att_feats,
    att_masks, labels, masks, gts).  Returns (loss, aux)."""
    noise = noise or {}
    flags = dict(vse=cfg['vse_loss_weight'], mle=cfg['caption_loss_weight'],
                 cider=cfg['cider_optimization'], disc=cfg['retrieval_reward_weight'])
    seq, masks = batch['labels'], batch['masks']
    fc, att, att_masks = batch['fc_feats'], batch['att_feats'], batch['att_masks']
    V = cfg['vocab_size']
    S = _Speaker(cfg)
    if is_alternating:
        if turn == 'speaker':                                          # :508-526
            flags['vse'] = 0
        elif turn == 'listener':                                       # :528-555
            flags.update(mle=0, cider=0, disc=0)
            _seqs, _ = S.sample(Ps, cfg, fc, att, att_masks, {'sample_max': 0, 'temperature': 1},
                                noise.get('sample'), cfg['retrieval_reward'])
            _seqs = _seqs.detach()
            masks = gen_masks_from(_seqs)
            seq = torch.cat([torch.full((_seqs.shape[0], 1), V + 1, dtype=torch.long), _seqs], 1)
    loss, aux = _forward_plain(Ps, Pl, cfg, flags, fc, att, att_masks, seq, masks, batch.get('gts'), noise)
    if is_alternating and turn == 'listener':
        aux['gen_result'] = _seqs                                       # the captions the listener was trained on
    return loss, aux


def _forward_plain(Ps, Pl, cfg, flags, fc, att, att_masks, seq, masks, gts, noise):
    """The non-alternating branch, AlternatingJointModel.py:443-504."""
    aux = {}
    S = _Speaker(cfg)
    rr = cfg['retrieval_reward']
    oor = cfg.get('only_one_retrieval', 'off')
    V = cfg['vocab_size']
    B = fc.shape[0]
    loss_cap = torch.zeros(1)
    if flags['mle'] > 0:                                               # ce_loss :196-207
        loss_cap = S.mle_forward(Ps, cfg, fc, att, att_masks, seq, masks, noise.get('mle'),
                                 cfg.get('ss_prob', 0.0))
        aux['loss_cap'] = loss_cap.detach()
    loss_vse = torch.zeros(1)
    if flags['vse'] > 0:                                               # vse_loss :209-224
        loss_vse = Lst.vse_forward(Pl, cfg, fc, seq, masks, False, oor)
        aux['loss_vse'] = loss_vse.detach()
    loss = flags['mle'] * loss_cap + flags['vse'] * loss_vse           # :451-452
    gen_result = greedy_res = None
    if flags['disc'] > 0:                                              # :455-488
        if rr == 'reinforce':
            _seqs, slp = S.sample(Ps, cfg, fc, att, att_masks, {'sample_max': 0, 'temperature': 1},
                                  noise.get('sample'), rr)             # reinforce_disc :226-247
            gen_result, sample_logprobs = _seqs, slp
            _masks = gen_masks_from(_seqs)
            gen_masks = _masks
            _seqs_b = torch.cat([torch.full((B, 1), V + 1, dtype=torch.long), _seqs], 1)
            retrieval_loss = Lst.vse_forward(Pl, cfg, fc, _seqs_b, _masks, True, oor)
            btype = cfg.get('reinforce_baseline_type', 'greedy')
            if btype == 'greedy':                                      # greedy_baseline :250-298
                with torch.no_grad():
                    g, _ = S.sample(Ps, cfg, fc, att, att_masks, {'sample_max': 1, 'temperature': 1},
                                    noise.get('greedy_baseline'), rr)
                greedy_res = g
                gm = gen_masks_from(g)
                gb = torch.cat([torch.full((B, 1), V + 1, dtype=torch.long), g], 1)
                baseline = Lst.vse_forward(Pl, cfg, fc, gb, gm, True, oor)
            elif btype == 'gt':                                        # gt_baseline :300-310
                baseline = Lst.vse_forward(Pl, cfg, fc, seq, masks, True, oor)
            else:                                                      # no_baseline :312-319
                baseline = torch.zeros(B)
            sc_loss = slp * (retrieval_loss - baseline).detach().unsqueeze(1) * _masks[:, 1:].detach()
            sc_loss = sc_loss.sum() / _masks[:, 1:].sum()              # loss_configuration :321-332
            loss = loss + flags['disc'] * sc_loss
            aux.update(retrieval_sc_loss=sc_loss.detach(), retrieval_loss=retrieval_loss.sum().detach(),
                       retrieval_loss_greedy=baseline.sum().detach())
        else:                                                          # st_and_ps_methods :343-376
            word_index, _seqs, slp = S.sample(
                Ps, cfg, fc, att, att_masks, {'sample_max': 0, 'temperature': 1, 'use_one_hot': 1},
                noise.get('sample'), rr)
            gen_result, sample_logprobs = word_index, slp
            _masks = gen_masks_from(word_index)
            gen_masks = _masks
            bos = torch.zeros(B, 1, V + 2)
            bos[:, 0, V + 1] = 1
            _seqs = torch.cat([bos, _seqs], 1)
            loss_vse_gen = Lst.vse_forward(Pl, cfg, fc, _seqs, _masks, False, oor)
            loss = loss + loss_vse_gen * flags['disc']
            aux['loss_vse_gen'] = loss_vse_gen.detach()
    if flags['cider']:                                                 # :490-503
        if gen_result is None or rr in ('multinomial_soft', 'gumbel_softmax'):
            gen_result, sample_logprobs = S.sample(Ps, cfg, fc, att, att_masks, {'sample_max': 0},
                                                   noise.get('cider_gen'), rr)   # :378-389
            gen_masks = gen_masks_from(gen_result)
        if greedy_res is None:
            with torch.no_grad():                                      # :391-403 (dropout stays on)
                greedy_res, _ = S.sample(Ps, cfg, fc, att, att_masks, {'sample_max': 1},
                                         noise.get('greedy'), rr)
        out = ciderd.get_self_critical_reward(gts, gen_result.numpy(), greedy_res.numpy(),
                                              bool(cfg.get('use_gen_cider_scores', 0)))
        if cfg.get('use_gen_cider_scores', 0):                         # traditional_cider :405-431
            reward, _, cider_greedy = out
        else:
            reward, cider_greedy = out
        r = torch.from_numpy(-reward.astype('float32'))
        loss_cider = sample_logprobs * r.unsqueeze(1) * gen_masks[:, 1:].detach()
        loss_cider = loss_cider.sum() / gen_masks[:, 1:].sum()
        loss = loss + flags['cider'] * loss_cider
        aux.update(avg_reward=float(reward.mean()), cider_greedy=float(cider_greedy),
                   loss_cider=loss_cider.detach(), reward=reward, greedy_res=greedy_res)
    aux['gen_result'] = gen_result
    aux.setdefault('greedy_res', greedy_res)
    return loss, aux


def clamp_adam_step(params, grads, state, lr, grad_clip=0.1, betas=(0.9, 0.999), eps=1e-8,
                    weight_decay=0.0):
    """clip_gradient (misc/utils.py:65-69) then torch.optim.Adam.step with default
    betas/eps (optimizer.py:25-27, 233-242).  In-place on the dicts of fp32 tensors.
    state: dict name -> dict(step, exp_avg, exp_avg_sq)."""
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        g = g.clamp(-grad_clip, grad_clip)
        if weight_decay != 0:
            g = g + weight_decay * p
        st = state.setdefault(k, dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p)))
        st['step'] += 1
        b1, b2 = betas
        st['exp_avg'].mul_(b1).add_(g, alpha=1 - b1)
        st['exp_avg_sq'].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** st['step']
        bc2 = 1 - b2 ** st['step']
        step_size = lr / bc1
        denom = (st['exp_avg_sq'].sqrt() / np.sqrt(bc2)).add_(eps)
        p.addcdiv_(st['exp_avg'], denom, value=-step_size)


def train_step(Ps, Pl, cfg, batch, noise, turn, state_s, state_l, lr, grad_clip=0.1):
    """One iteration of the reference's trainer around joint_forward (train.py:487-531): zeroing_optimizer
    (optimizer.py:224-230), the requires_grad effect of changeModelUpdateStatus in the reinforce turns
    (AlternatingJointModel.py:508-555,571-685: the vse loop runs first, the caption model's last), forward, backward,
    update_optimizer (optimizer.py:233-242: clamp, then Adam, for one or both agents).  With share_embed = 1
    (AlternatingJointModel.py:83-88) the SAME tensor object sits in Ps['embed.0.weight'] and Pl['txt_enc.embed.weight']:
    both paths accumulate into its one .grad, each optimizer that steps moves it with its own moments, and in a listener turn
    it is frozen because the caption model's loop sets its flag last.  torch 0.4.1's zero_grad() zero-fills (a frozen
    parameter then has a ZERO gradient, not None, and Adam still visits it).  Returns (loss, aux)."""
    rr = cfg['retrieval_reward']
    both = rr != 'reinforce'                                  # gumbel / multinomial*: the listener turn is removed, both step
    agents = (('s', Ps, state_s), ('l', Pl, state_l))
    stepping = [a for a in agents if both or (a[0] == 's') == (turn == 'speaker')]
    for _, P, _ in stepping:                                  # zeroing_optimizer
        for v in P.values():
            v.grad = torch.zeros_like(v)
    if rr == 'reinforce':                                     # forward :508-555
        flags = {'speaker': (False, True), 'listener': (True, False)}[turn]
        for v in Pl.values():
            v.requires_grad_(flags[0])
        for v in Ps.values():                                 # (a shared table takes the caption model's flag)
            v.requires_grad_(flags[1])
    loss, aux = joint_forward(Ps, Pl, cfg, batch, noise, turn, True)
    if loss.requires_grad:
        loss.backward()
    with torch.no_grad():
        for _, P, st in stepping:                             # update_optimizer: speaker first, then listener
            clamp_adam_step(P, {k: v.grad for k, v in P.items()}, st, lr, grad_clip)
    return loss, aux
