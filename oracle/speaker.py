"""Oracle (test infrastructure): att2in2 speaker, fp32 PyTorch-CPU restatement.

Follows /root/reference/models/AttModel.py (Att2in2Model = AttModel + Att2in2Core
+ Attention) and the four samplers models/{gumbel,multinomial,gumbel_softmax,
multinomial_soft}.py.  Every function cites the lines it restates.  Parameters
are passed as a dict ``P`` keyed by the reference's state-dict names
(``embed.0.weight``, ``att_embed.0.weight``, ``logit.weight``, ``ctx2att.weight``,
``core.a2c.weight``, ``core.i2h.weight``, ``core.h2h.weight``,
``core.attention.h2att.weight``, ``core.attention.alpha_net.weight`` + biases).

Noise is explicit (see oracle/__init__.py).  A ``noise`` dict may hold:
  att_keep  f32[B,K,H]    keep mask of the att_embed dropout      (AttModel.py:82-85)
  x_keep    f32[T,B,E]    keep mask of the token-embed dropout    (AttModel.py:74-78)
  out_keep  f32[T,B,H]    keep mask of the core output dropout    (AttModel.py:506,529)
  gumbel_u  f32[T,B,V+1]  uniforms of sample_gumbel, row t is used when choosing
                          the input of step t (t>=1)              (gumbel.py:6-11)
  pick      i64[T,B]      multinomial picks (row t as above)      (multinomial.py:15)
  ps_u      f32[T,B]      partial-sampling row uniforms           (gumbel_softmax.py:31)
  ss_u      f32[T,B]      scheduled-sampling uniforms (MLE)       (AttModel.py:119)
Missing entries mean "no dropout" (keep masks) or "draw with torch" (the rest).
"""
import torch
import torch.nn.functional as F


def dropout(x, keep, p):
    """nn.Dropout in training mode with an explicit keep mask: x * keep / (1-p)."""
    if keep is None or p == 0.0:
        return x
    return x * (keep / (1.0 - p))


def att_embed(P, att_raw, keep, p, att_masks=None):
    """AttModel.py:82-85,110,315 (use_bn=0) through pack_wrapper :44-51.  With att_masks the reference packs the first
    len_b = sum_k att_masks[b, k] region rows of every image (sort_pack_padded_sequence :30-36), applies the
    module to the packed rows only and pads back with zeros (pad_unsort_packed_sequence :38-41): rows k >= len_b
    of the embedded features are exactly 0 (not relu(bias)), and the keep mask covers the packed rows only."""
    lin = 'att_embed.0'
    if 'att_embed.1.weight' in P:
        # use_bn = 1 (:82-85): BatchNorm1d(att_feat_size) in front of the Linear.  pack_wrapper hands the module the PACKED valid
        # region rows [N, D], so the batch statistics run over an image's own regions only; training mode: biased variance
        # in the normalisation (eps 1e-5), as torch.nn.functional.batch_norm.  Without att_masks the reference feeds the 3-D
        # [B, K, D] tensor, which BatchNorm1d(D) rejects (K channels): the option cannot run there, and does not here.
        if att_masks is None:
            raise ValueError('use_bn = 1 needs att_masks: BatchNorm1d(att_feat_size) rejects the unpacked [B, K, D] features')
        valid_rows = att_masks > 0
        x = att_raw[valid_rows]                                      # [N, D] (the statistics do not depend on the row order)
        if P.get('_bn_training', True):
            mean, var = x.mean(0), x.var(0, unbiased=False)
        else:
            mean, var = P['att_embed.0.running_mean'], P['att_embed.0.running_var']
        att_raw = (att_raw - mean) / torch.sqrt(var + 1e-5) * P['att_embed.0.weight'] + P['att_embed.0.bias']
        lin = 'att_embed.1'
    y = F.linear(att_raw, P[lin + '.weight'], P[lin + '.bias'])
    y = dropout(torch.relu(y), keep, p)
    if att_masks is not None:
        lens = att_masks.long().sum(1)
        K = att_raw.shape[1]
        assert int(lens.max()) == K, 'pad_packed_sequence pads to the longest image: one image must use all K rows'
        valid = (torch.arange(K).unsqueeze(0) < lens.unsqueeze(1)).to(y.dtype)
        y = y * valid.unsqueeze(2)
    return y


def ctx2att(P, att):
    """AttModel.py:88,114,319."""
    return F.linear(att, P['ctx2att.weight'], P['ctx2att.bias'])


def attention_step(P, h, att, p_att, att_masks=None):
    """Attention.forward, AttModel.py:465-489.  Returns (att_res[B,H], alpha[B,K])."""
    att_h = F.linear(h, P['core.attention.h2att.weight'], P['core.attention.h2att.bias'])
    dot = torch.tanh(p_att + att_h.unsqueeze(1))                      # :472-474
    dot = F.linear(dot, P['core.attention.alpha_net.weight'],
                   P['core.attention.alpha_net.bias']).squeeze(2)     # :476-478
    weight = torch.softmax(dot, dim=1)                                # :480
    if att_masks is not None:                                         # :481-483
        weight = weight * att_masks.float()
        weight = weight / weight.sum(1, keepdim=True)
    att_res = torch.bmm(weight.unsqueeze(1), att).squeeze(1)          # :487
    return att_res, weight


def att2in2_cell(P, xt, att_res, h, c):
    """Att2in2Core.forward, AttModel.py:514-527 (without the output dropout)."""
    H = h.shape[1]
    s = F.linear(xt, P['core.i2h.weight'], P['core.i2h.bias']) + \
        F.linear(h, P['core.h2h.weight'], P['core.h2h.bias'])         # :514
    sig = torch.sigmoid(s[:, :3 * H])                                 # :515-516
    i, f, o = sig[:, :H], sig[:, H:2 * H], sig[:, 2 * H:3 * H]        # :517-519
    g = s[:, 3 * H:] + F.linear(att_res, P['core.a2c.weight'], P['core.a2c.bias'])  # :521-522
    g = torch.max(g[:, :H], g[:, H:])                                 # :523-525
    c2 = f * c + i * g                                                # :526
    h2 = o * torch.tanh(c2)                                           # :527
    return h2, c2


def core_step(P, xt, att, p_att, att_masks, h, c, out_keep, p):
    """Att2in2Core.forward AttModel.py:510-531: attention + cell + output dropout."""
    att_res, alpha = attention_step(P, h, att, p_att, att_masks)
    h2, c2 = att2in2_cell(P, xt, att_res, h, c)
    return dropout(h2, out_keep, p), h2, c2, alpha


def logprobs_from_output(P, out, neg_inf_cols=None):
    """logit + log_softmax, AttModel.py:140,438-444.  neg_inf_cols: i64[B] columns
    to suppress (decoding_constraint, :438-442) or None."""
    logits = F.linear(out, P['logit.weight'], P['logit.bias'])
    if neg_inf_cols is not None:
        tmp = torch.zeros_like(logits)
        tmp.scatter_(1, neg_inf_cols.unsqueeze(1), float('-inf'))
        logits = logits + tmp
    return F.log_softmax(logits, dim=1), logits


def embed_token(P, it, keep, p):
    """self.embed = Embedding -> ReLU -> Dropout, AttModel.py:74-76,399."""
    return dropout(torch.relu(P['embed.0.weight'][it]), keep, p)


# ----------------------------------------------------------------------------
# samplers
# ----------------------------------------------------------------------------
def sample_gumbel_from_u(U, eps=1e-20):
    """gumbel.py:6-11 with U supplied: -log(-log(U+eps)+eps) in fp32."""
    return -torch.log(-torch.log(U + eps) + eps)


def gumbel_st(logp, tau, U):
    """gumbel_softmax, gumbel.py:13-30.  Returns (one_hot, ind, y)."""
    y = torch.softmax((logp + sample_gumbel_from_u(U)) / tau, dim=-1)
    ind = y.max(dim=-1)[1]
    y_hard = torch.zeros_like(y).scatter_(1, ind.view(-1, 1), 1.0)
    one_hot = (y_hard - y).detach() + y
    return one_hot, ind, y


def multinomial_st(logp, tau, pick=None):
    """multinomial, multinomial.py:4-27."""
    y = torch.softmax(logp, 1) if tau == 1 else torch.softmax(logp / tau, 1)
    ind = torch.multinomial(y, 1).squeeze(1) if pick is None else pick
    y_hard = torch.zeros_like(y).scatter_(1, ind.view(-1, 1), 1.0)
    one_hot = (y_hard - y).detach() + y
    return one_hot, ind, y


def gumbel_ps(logp, tau, ss_prob, U, row_u=None):
    """gumbel_soft, gumbel_softmax.py:17-42 (partial sampling)."""
    y = torch.softmax((logp + sample_gumbel_from_u(U)) / tau, dim=-1)
    ind = y.max(dim=-1)[1]
    y_hard = torch.zeros_like(y).scatter_(1, ind.view(-1, 1), 1.0)
    if ss_prob > 0.0:
        if row_u is None:
            row_u = torch.zeros(logp.shape[0]).uniform_(0, 1)
        m = (row_u < ss_prob).view(-1, 1).to(y.dtype)
        out = (y_hard * m - y * m).detach() + y
    else:
        out = y
    return out, ind, y


def multinomial_ps(logp, tau, ss_prob, pick=None, row_u=None):
    """multinomial_soft, multinomial_soft.py:5-35 (exp(logp/tau), unnormalised for tau!=1)."""
    y = torch.exp(logp) if tau == 1 else torch.exp(logp / tau)
    ind = torch.multinomial(y, 1).squeeze(1) if pick is None else pick
    y_hard = torch.zeros_like(y).scatter_(1, ind.view(-1, 1), 1.0)
    if ss_prob > 0.0:
        if row_u is None:
            row_u = torch.zeros(logp.shape[0]).uniform_(0, 1)
        m = (row_u < ss_prob).view(-1, 1).to(y.dtype)
        out = (y_hard * m - y * m).detach() + y
    else:
        out = y
    return out, ind, y


# ----------------------------------------------------------------------------
# decode loops
# ----------------------------------------------------------------------------
def _n(noise, key, t=None):
    if noise is None or key not in noise or noise[key] is None:
        return None
    return noise[key] if t is None else noise[key][t]


def sample(P, cfg, fc_feats, att_raw, att_masks, opt=None, noise=None, retrieval_reward='gumbel',
           return_trace=False):
    """AttModel.sample, AttModel.py:291-452 (beam_size==1).

    cfg: dict(vocab_size, seq_length, drop_prob_lm, gumbel_temp, multinomial_temp,
              prob_gumbel_softmax, prob_multinomial_soft, decoding_constraint)
    Returns (seq, logprobs) or (word_index, one_hot/soft, logprobs) exactly as the
    reference does; with return_trace also a dict of per-step internals.
    """
    opt = opt or {}
    use_one_hot = opt.get('use_one_hot', 0)
    sample_max = opt.get('sample_max', 1)
    temperature = opt.get('temperature', 1.0)
    decoding_constraint = opt.get('decoding_constraint', cfg.get('decoding_constraint', 0))
    V = cfg['vocab_size']
    p = cfg['drop_prob_lm']
    B = fc_feats.shape[0]
    H = P['core.h2h.weight'].shape[1]
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    att = att_embed(P, att_raw, _n(noise, 'att_keep'), p, att_masks)  # :315
    p_att = ctx2att(P, att)                                           # :319
    eos_one_hot = torch.zeros(1, V + 2)
    eos_one_hot[0, 0] = 1.0                                           # :297-304
    plain = (retrieval_reward == 'reinforce') or (not use_one_hot)
    word_index, seq, seq_logp = [], [], []
    trace = {'logprobs': [], 'alpha': [], 'h': [], 'c': [], 'it': [], 'y': []}
    logprobs = None
    unfinished = None
    for t in range(cfg['seq_length'] + 1):                            # :323
        soft_vec = None
        one_hot = None
        if t == 0:
            it = torch.full((B,), V + 1, dtype=torch.long)            # :324-326
        elif sample_max:
            slp, it = torch.max(logprobs, 1)                          # :328-329
        elif plain:                                                   # :332-343
            prob_prev = torch.exp(logprobs) if temperature == 1.0 else torch.exp(logprobs / temperature)
            pk = _n(noise, 'pick', t)
            it = torch.multinomial(prob_prev, 1).view(-1) if pk is None else pk
            slp = logprobs.gather(1, it.unsqueeze(1)).view(-1)
        elif retrieval_reward == 'gumbel':                            # :345-354
            U = _n(noise, 'gumbel_u', t)
            if U is None:
                U = torch.rand(logprobs.shape)
            one_hot, it, y = gumbel_st(logprobs, cfg['gumbel_temp'], U)
            slp = logprobs.gather(1, it.unsqueeze(1)).view(-1)
            one_hot = torch.cat([one_hot, torch.zeros(B, 1)], 1)
            trace['y'].append(y)
        elif retrieval_reward == 'multinomial':                       # :356-365
            one_hot, it, y = multinomial_st(logprobs, cfg['multinomial_temp'], _n(noise, 'pick', t))
            slp = logprobs.gather(1, it.unsqueeze(1)).view(-1)
            one_hot = torch.cat([one_hot, torch.zeros(B, 1)], 1)
            trace['y'].append(y)
        elif retrieval_reward == 'gumbel_softmax':                    # :367-378
            U = _n(noise, 'gumbel_u', t)
            if U is None:
                U = torch.rand(logprobs.shape)
            soft_vec, it, y = gumbel_ps(logprobs, cfg['gumbel_temp'], cfg['prob_gumbel_softmax'],
                                        U, _n(noise, 'ps_u', t))
            slp = logprobs.gather(1, it.unsqueeze(1)).view(-1)
            soft_vec = torch.cat([soft_vec, torch.zeros(B, 1)], 1)
        elif retrieval_reward == 'multinomial_soft':                  # :381-392
            soft_vec, it, y = multinomial_ps(logprobs, cfg['multinomial_temp'],
                                             cfg['prob_multinomial_soft'],
                                             _n(noise, 'pick', t), _n(noise, 'ps_u', t))
            slp = logprobs.gather(1, it.unsqueeze(1)).view(-1)
            soft_vec = torch.cat([soft_vec, torch.zeros(B, 1)], 1)
        else:
            raise ValueError(retrieval_reward)

        if soft_vec is not None and not plain and t >= 1:             # :395-397
            xt = dropout(torch.relu(soft_vec @ P['embed.0.weight']), _n(noise, 'x_keep', t), p)
        else:
            xt = embed_token(P, it, _n(noise, 'x_keep', t), p)        # :399
        trace['it'].append(it)

        if t >= 1:                                                    # :401-434
            unfinished = (it > 0) if t == 1 else unfinished * (it > 0)
            if unfinished.sum() == 0:
                break
            it = it * unfinished.type_as(it)
            if plain:
                seq.append(it)
                seq_logp.append(slp.view(-1))
            elif retrieval_reward in ('gumbel', 'multinomial'):
                word_index.append(it)
                one_hot = one_hot * unfinished.unsqueeze(1).float()
                if bool((unfinished == 0).any()):
                    one_hot = torch.where((unfinished == 0).unsqueeze(1), eos_one_hot.expand(B, -1), one_hot)
                seq.append(one_hot)
                seq_logp.append(slp.view(-1))
            else:
                word_index.append(it)
                soft_vec = soft_vec * unfinished.unsqueeze(1).float()
                if bool((unfinished == 0).any()):
                    soft_vec = torch.where((unfinished == 0).unsqueeze(1), eos_one_hot.expand(B, -1), soft_vec)
                seq.append(soft_vec)
                seq_logp.append(slp.view(-1))

        out, h, c, alpha = core_step(P, xt, att, p_att, att_masks, h, c, _n(noise, 'out_keep', t), p)  # :436
        neg = None
        if decoding_constraint and len(seq) > 0 and plain:            # :438-442
            neg = seq[-1]
        logprobs, _ = logprobs_from_output(P, out, neg)               # :444
        trace['logprobs'].append(logprobs)
        trace['alpha'].append(alpha)
        trace['h'].append(h)
        trace['c'].append(c)

    cat = lambda xs: torch.cat([x.unsqueeze(1) for x in xs], 1)       # noqa: E731
    if plain:
        res = (cat(seq), cat(seq_logp))                               # :445-447
    else:
        res = (cat(word_index), cat(seq), cat(seq_logp))              # :448-452
    if return_trace:
        trace['att'] = att
        trace['p_att'] = p_att
        return res, trace
    return res


def language_model_criterion(logp, target, mask):
    """LanguageModelCriterion.forward, misc/utils.py:49-58."""
    T = logp.shape[1]
    target = target[:, :T]
    mask = mask[:, :T]
    out = -logp.gather(2, target.unsqueeze(2)).squeeze(2) * mask
    return out.sum() / mask.sum()


def mle_forward(P, cfg, fc_feats, att_raw, att_masks, seq, masks, noise=None, ss_prob=0.0):
    """AttModel.forward, AttModel.py:103-148 (training mode)."""
    p = cfg['drop_prob_lm']
    B = fc_feats.shape[0]
    H = P['core.h2h.weight'].shape[1]
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    att = att_embed(P, att_raw, _n(noise, 'att_keep'), p, att_masks)
    p_att = ctx2att(P, att)
    outputs = []
    for i in range(seq.shape[1] - 1):                                 # :116
        if i >= 1 and ss_prob > 0.0:                                  # :118-129
            u = _n(noise, 'ss_u', i)
            if u is None:
                u = torch.zeros(B).uniform_(0, 1)
            sample_mask = u < ss_prob
            it = seq[:, i].clone()
            if sample_mask.sum() != 0:
                pk = _n(noise, 'pick', i)
                if pk is None:
                    pk = torch.multinomial(torch.exp(outputs[-1].detach()), 1).view(-1)
                it = torch.where(sample_mask, pk, it)
        else:
            it = seq[:, i].clone()                                    # :131
        if i >= 1 and seq[:, i].sum() == 0:                           # :133-134
            break
        xt = embed_token(P, it, _n(noise, 'x_keep', i), p)            # :136
        out, h, c, _ = core_step(P, xt, att, p_att, att_masks, h, c, _n(noise, 'out_keep', i), p)
        logp, _ = logprobs_from_output(P, out)                        # :140
        outputs.append(logp)
    output = torch.stack(outputs, 1)                                  # :143
    return language_model_criterion(output, seq[:, 1:], masks[:, 1:])  # :144


def sample_beam(P, cfg, fc_feats, att_raw, att_masks, opt=None):
    """AttModel.sample_beam, AttModel.py:150-289 (evaluation mode: no dropout).  One image at a time, `beam_size`
    beams; per step every beam's log-probs are sorted, the top beam_size words of each beam form the candidates
    (word-rank major, beam minor), the candidates are stably sorted by cumulative log-prob and the first beam_size
    survive.  A beam that emits <eos> = 0 is recorded as done BUT KEEPS DECODING (:256-263 record, nothing removes
    it); at t == seq_length every beam is recorded.  The recorded score 'p' is `beam_logprobs_sum[vix]`, a 0-dim
    VIEW of the running sums (:262), so what the final sort (:281-282) compares is the FINAL running sum of the
    beam slot an entry was recorded from, not the sum at recording time; seq / logps are cloned snapshots.  The
    first recorded entry among those with the largest such score is returned.
    Returns (seq i64[B,T], logps f32[B,T], score f32[B])."""
    opt = opt or {}
    beam = opt.get('beam_size', 10)
    dc = opt.get('decoding_constraint', cfg.get('decoding_constraint', 0))
    V, T = cfg['vocab_size'], cfg['seq_length']
    B = att_raw.shape[0]
    H = P['core.h2h.weight'].shape[1]
    att_all = att_embed(P, att_raw, None, 0.0, att_masks)
    p_att_all = ctx2att(P, att_all)
    seq = torch.zeros(B, T, dtype=torch.long)
    logps = torch.zeros(B, T)
    score = torch.zeros(B)
    for k in range(B):
        att = att_all[k:k + 1].expand(beam, -1, -1)
        p_att = p_att_all[k:k + 1].expand(beam, -1, -1)
        am = att_masks[k:k + 1].expand(beam, -1) if att_masks is not None else None
        h = torch.zeros(beam, H)
        c = torch.zeros(beam, H)
        beam_seq = torch.zeros(T, beam, dtype=torch.long)
        beam_lp = torch.zeros(T, beam)
        beam_sum = torch.zeros(beam)
        done = []
        logprobs = None
        for t in range(T + 1):
            if t == 0:
                it = torch.full((beam,), V + 1, dtype=torch.long)                        # :190-193
            else:
                lp = logprobs.clone()
                if dc and t > 1:                                                          # :201-204
                    lp.scatter_(1, beam_seq[t - 2:t - 1].t(), float('-inf'))
                ys, ix = torch.sort(lp, 1, True)                                          # :207
                rows = 1 if t == 1 else beam                                              # :211-213
                cands = []
                for cc in range(min(beam, ys.shape[1])):
                    for q in range(rows):
                        r = ys[q, cc]
                        cands.append((int(ix[q, cc]), q, (beam_sum[q] + r).item(), r.item()))   # fp32 add (:218-219)
                cands = sorted(cands, key=lambda x: -x[2])                                # stable (:224)
                h_prev, c_prev = h.clone(), c.clone()
                seq_prev, lp_prev = beam_seq[:t - 1].clone(), beam_lp[:t - 1].clone()
                for vix in range(beam):
                    cw, q, p, r = cands[vix]
                    if t > 1:
                        beam_seq[:t - 1, vix] = seq_prev[:, q]
                        beam_lp[:t - 1, vix] = lp_prev[:, q]
                    h[vix], c[vix] = h_prev[q], c_prev[q]                                 # :241-246
                    beam_seq[t - 1, vix] = cw
                    beam_lp[t - 1, vix] = r
                    beam_sum[vix] = p
                    if cw == 0 or t == T:                                                 # :256-263
                        done.append((beam_seq[:, vix].clone(), beam_lp[:, vix].clone(), vix))   # 'p' aliases slot vix
                it = beam_seq[t - 1]
            xt = embed_token(P, it, None, 0.0)
            _, h, c, _ = core_step(P, xt, att, p_att, am, h, c, None, 0.0)
            logprobs, _ = logprobs_from_output(P, h)
        best = sorted(done, key=lambda x: -float(beam_sum[x[2]]))[0]                      # :281-285, final slot sums
        seq[k], logps[k], score[k] = best[0], best[1], beam_sum[best[2]]
    return seq, logps, score
