"""Oracle (test infrastructure): VSE-fc listener, fp32 PyTorch-CPU restatement.

Follows /root/reference/models/VSEFCModel.py.  Parameters are a dict ``P`` keyed by
the reference's state-dict names: ``img_enc.fc.weight/bias``,
``txt_enc.embed.weight``, ``txt_enc.rnn.{weight_ih_l0,weight_hh_l0,bias_ih_l0,bias_hh_l0}``.
The GRU is written out (gate order r,z,n as in torch.nn.GRU) instead of calling
nn.GRU on a packed sequence; picking h at step len-1 is what
pack_padded_sequence + gather(len-1) does (VSEFCModel.py:108-129).
"""
import torch
import torch.nn.functional as F


def l2norm(X):
    """VSEFCModel.py:12-17 — the 1e-7 is added to the norm, not under the sqrt."""
    norm = torch.norm(X, dim=1, keepdim=True) + 1e-7
    return X / norm


def encode_image(P, fc_feats, no_imgnorm=0, use_abs=0):
    """EncoderImage.forward, VSEFCModel.py:40-54."""
    f = F.linear(fc_feats, P['img_enc.fc.weight'], P['img_enc.fc.bias'])
    if not no_imgnorm:
        f = l2norm(f)
    if use_abs:
        f = torch.abs(f)
    return f


def gru_cell(P, x, h):
    """One torch.nn.GRU step (cuDNN/ATen gate order r, z, n)."""
    J = h.shape[1]
    gi = F.linear(x, P['txt_enc.rnn.weight_ih_l0'], P['txt_enc.rnn.bias_ih_l0'])
    gh = F.linear(h, P['txt_enc.rnn.weight_hh_l0'], P['txt_enc.rnn.bias_hh_l0'])
    r = torch.sigmoid(gi[:, :J] + gh[:, :J])
    z = torch.sigmoid(gi[:, J:2 * J] + gh[:, J:2 * J])
    n = torch.tanh(gi[:, 2 * J:] + r * gh[:, 2 * J:])
    return (1 - z) * n + z * h


def encode_text(P, seqs, masks, pool_type='last', use_abs=0):
    """EncoderText.forward, VSEFCModel.py:95-140.  seqs: i64[B,L] indices or
    f32[B,L,V+2] one-hot/soft rows (dense matmul path :102-104)."""
    lens = (masks > 0).long().sum(1)                                   # :84
    if seqs.dim() > 2:
        emb = torch.matmul(seqs, P['txt_enc.embed.weight'])           # :104
    else:
        emb = P['txt_enc.embed.weight'][seqs]                         # :106
    B, L = emb.shape[0], emb.shape[1]
    J = P['txt_enc.rnn.weight_hh_l0'].shape[1]
    h = torch.zeros(B, J)
    hs = []
    for t in range(L):
        h = gru_cell(P, emb[:, t], h)
        hs.append(h)
    hs = torch.stack(hs, 1)                                            # [B,L,J]
    if pool_type == 'mean':                                            # :118-122
        m = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float()
        out = (hs * m.unsqueeze(-1) * masks[:, :L].float().unsqueeze(-1)).sum(1) / \
            masks.float().sum(1, keepdim=True)
    elif pool_type == 'max':                                           # :123-127
        m = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float()
        mm = masks[:, :L].float()
        out = ((hs * m.unsqueeze(-1)) * mm.unsqueeze(-1) + (mm == 0).unsqueeze(-1).float() * -1e10).max(1)[0]
    else:                                                              # :128-129
        idx = (lens - 1).view(-1, 1, 1).expand(B, 1, J)
        out = hs.gather(1, idx).squeeze(1)
    out = l2norm(out)                                                  # :132
    if use_abs:
        out = torch.abs(out)
    return out


def contrastive_loss(im, s, margin=0.2, max_violation=1, whole_batch=False, only_one_retrieval='off'):
    """ContrastiveLoss.forward, VSEFCModel.py:167-207."""
    scores = im.mm(s.t())                                              # :169 (cosine_sim :143-146)
    diagonal = scores.diag().view(im.size(0), 1)
    d1 = diagonal.expand_as(scores)
    d2 = diagonal.t().expand_as(scores)
    cost_s = (margin + scores - d1).clamp(min=0)                       # :176
    cost_im = (margin + scores - d2).clamp(min=0)                      # :179
    I = torch.eye(scores.size(0)) > .5
    cost_s = cost_s.masked_fill(I, 0)                                  # :186-187
    cost_im = cost_im.masked_fill(I, 0)
    if max_violation:                                                  # :190-195
        cost_s = cost_s.max(1)[0]
        cost_im = cost_im.max(0)[0]
    else:
        cost_s = cost_s.mean(1)
        cost_im = cost_im.mean(0)
    fn = (lambda x: x) if whole_batch else (lambda x: x.sum())         # :197-200
    if only_one_retrieval == 'image':
        return fn(cost_im)
    elif only_one_retrieval == 'caption':
        return fn(cost_s)
    return fn(cost_s) + fn(cost_im)


def vse_forward(P, cfg, fc_feats, seq, masks, whole_batch=False, only_one_retrieval='off'):
    """VSEFCModel.forward, VSEFCModel.py:230-241.  cfg: vse_margin, vse_max_violation,
    vse_no_imgnorm, vse_use_abs, vse_pool_type."""
    img = encode_image(P, fc_feats, cfg.get('vse_no_imgnorm', 0), cfg.get('vse_use_abs', 0))
    cap = encode_text(P, seq, masks, cfg.get('vse_pool_type', 'last'), cfg.get('vse_use_abs', 0))
    return contrastive_loss(img, cap, cfg.get('vse_margin', 0.2), cfg.get('vse_max_violation', 1),
                            whole_batch, only_one_retrieval)
