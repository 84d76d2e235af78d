"""Oracle (test infrastructure): FCModel, the fc-feature speaker of the reference's CPU plumbing
configuration (BASELINE configs[0]), fp32 PyTorch-CPU restatement.

Follows /root/reference/models/FCModel.py: LSTMCore (:12-43), FCModel.forward (:91-131) and
FCModel.sample (:260-327, beam_size == 1, sample_max 1 or 0).  Parameters are a dict ``P`` keyed by the
reference's state-dict names (img_embed.weight/bias, core.i2h.*, core.h2h.*, embed.weight, logit.*).

Noise (see oracle/__init__.py): ``out_keep`` f32[T+2,B,H] — keep masks of LSTMCore's dropout, one row per
core call: row 0 the image step, row 1 the <bos>/first-token step, ...; ``pick`` i64[T+2,B] multinomial picks
(row t = the draw made at loop iteration t of FCModel.sample).
"""
import torch
import torch.nn.functional as F

from .speaker import dropout, language_model_criterion, _n


def lstm_core(P, xt, h, c, keep, p):
    """LSTMCore.forward, FCModel.py:24-43.  NOTE the recurrent state is the DROPPED-OUT next_h (:38-42),
    unlike Att2in2Core, which feeds the un-dropped state back."""
    H = h.shape[1]
    s = F.linear(xt, P['core.i2h.weight'], P['core.i2h.bias']) + \
        F.linear(h, P['core.h2h.weight'], P['core.h2h.bias'])                 # :26
    sig = torch.sigmoid(s[:, :3 * H])                                         # :27-28
    i, f, o = sig[:, :H], sig[:, H:2 * H], sig[:, 2 * H:3 * H]                # :29-31
    g = torch.max(s[:, 3 * H:4 * H], s[:, 4 * H:])                            # :33-35
    c2 = f * c + i * g                                                        # :36
    h2 = dropout(o * torch.tanh(c2), keep, p)                                 # :37-39
    return h2, c2


def fc_forward(P, cfg, fc_feats, seq, masks, noise=None):
    """FCModel.forward, FCModel.py:91-131 (ss_prob == 0)."""
    p = cfg['drop_prob_lm']
    B = fc_feats.shape[0]
    H = P['core.h2h.weight'].shape[1]
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    outputs = []
    for i in range(seq.shape[1]):                                             # :97
        if i == 0:
            xt = F.linear(fc_feats, P['img_embed.weight'], P['img_embed.bias'])   # :99
        else:
            it = seq[:, i - 1].clone()                                        # :115
            if i >= 2 and seq[:, i - 1].sum() == 0:                           # :117-118
                break
            xt = P['embed.weight'][it]                                        # :119
        h, c = lstm_core(P, xt, h, c, _n(noise, 'out_keep', i), p)            # :121
        outputs.append(F.log_softmax(F.linear(h, P['logit.weight'], P['logit.bias']), dim=1))   # :122
    output = torch.stack(outputs[1:], 1)                                      # :125-126
    return language_model_criterion(output, seq[:, 1:], masks[:, 1:])          # :127


def fc_sample(P, cfg, fc_feats, opt=None, noise=None):
    """FCModel.sample, FCModel.py:260-327 (beam_size 1; sample_max 1 = greedy, 0 = multinomial)."""
    opt = opt or {}
    sample_max = opt.get('sample_max', 1)
    temperature = opt.get('temperature', 1.0)
    decoding_constraint = opt.get('decoding_constraint', cfg.get('decoding_constraint', 0))
    p = cfg['drop_prob_lm']
    V = cfg['vocab_size']
    B = fc_feats.shape[0]
    H = P['core.h2h.weight'].shape[1]
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    seq, seq_logp = [], []
    logprobs = None
    unfinished = None
    for t in range(cfg['seq_length'] + 2):                                    # :274
        if t == 0:
            xt = F.linear(fc_feats, P['img_embed.weight'], P['img_embed.bias'])   # :276
        else:
            if t == 1:
                it = torch.full((B,), V + 1, dtype=torch.long)                # :278-280
            elif sample_max == 1:
                slp, it = torch.max(logprobs, 1)                              # :281-283
            elif sample_max == 0:
                prob_prev = torch.exp(logprobs) if temperature == 1.0 else torch.exp(logprobs / temperature)
                pk = _n(noise, 'pick', t)
                it = torch.multinomial(prob_prev, 1).view(-1) if pk is None else pk   # :290-300
                slp = logprobs.gather(1, it.unsqueeze(1)).view(-1)
            else:
                raise NotImplementedError('sample_max == 2 (in-place Gumbel argmax, FCModel.py:284-289)')
            xt = P['embed.weight'][it]                                        # :302
        if t >= 2:                                                            # :304-313
            unfinished = (it > 0) if t == 2 else unfinished * (it > 0)
            if unfinished.sum() == 0:
                break
            it = it * unfinished.type_as(it)
            seq.append(it)
            seq_logp.append(slp.view(-1))
        h, c = lstm_core(P, xt, h, c, _n(noise, 'out_keep', t), p)            # :315
        logits = F.linear(h, P['logit.weight'], P['logit.bias'])
        if decoding_constraint and len(seq) > 0:                              # :317-320
            tmp = torch.zeros_like(logits)
            tmp.scatter_(1, seq[-1].unsqueeze(1), float('-inf'))
            logits = logits + tmp
        logprobs = F.log_softmax(logits, dim=1)                               # :322
    return torch.stack(seq, 1), torch.stack(seq_logp, 1)                      # :324-325
