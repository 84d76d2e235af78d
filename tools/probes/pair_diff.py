#!/usr/bin/env python3
"""Debug probe: where do the workspaces of a paired decode and of two sequential decodes differ?"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cooperativeimagecaptioning_amd import engine, _lib
B = 32; K, D, H, V, T = 36, 64, 512, 9487, 16
g = torch.Generator().manual_seed(100 + B)
def lin(o, i, s=1.0):
    r = s / np.sqrt(i)
    return ((torch.rand(o, i, generator=g) * 2 - 1) * r).cuda(), ((torch.rand(o, generator=g) * 2 - 1) * r).cuda()
W = {'embed.0.weight': torch.randn(V + 2, H, generator=g).cuda()}
for nm, (o, i, s) in {'att_embed.0': (H, D, 1), 'logit': (V + 1, H, 6), 'ctx2att': (H, H, 1), 'core.a2c': (2 * H, H, 1),
                      'core.i2h': (5 * H, H, 1), 'core.h2h': (5 * H, H, 1), 'core.attention.h2att': (H, H, 1),
                      'core.attention.alpha_net': (1, H, 3)}.items():
    W[nm + '.weight'], W[nm + '.bias'] = lin(o, i, s)
W['logit.bias'][0] = 2.5
p = 0.5
d = engine.speaker_dims(B, K, D, H, H, H, V, T, p)
params = engine.speaker_params(W)
att_pre = engine.speaker_att_embed_fwd(d, params, (torch.randn(B, K, D, generator=g).abs() * 0.5).cuda())
def noise():
    return dict(att_keep=(torch.rand(B, K, H, generator=g) >= p).to(torch.uint8).cuda(),
                x_keep=(torch.rand(T + 1, B, H, generator=g) >= p).to(torch.uint8).cuda(),
                out_keep=(torch.rand(T + 1, B, H, generator=g) >= p).to(torch.uint8).cuda())
na, nb_ = noise(), noise()
U = torch.rand(T + 1, B, V + 1, generator=g).cuda()
def specs():
    a = engine.speaker_decode_io(d, params, att_pre, _lib.SAMPLE_GUMBEL_ST, 1.0, U=U, want_stv=True, **na)
    b = engine.speaker_decode_io(d, params, att_pre, _lib.SAMPLE_GREEDY, 1.0, **nb_)
    a['ws'].zero_(), b['ws'].zero_()
    return a, b
a0, b0 = specs()
engine.speaker_decode_launch(d, params, a0)
engine.speaker_decode_launch(d, params, b0)
a1, b1 = specs()
engine.speaker_decode_fwd_pair(d, params, a1, b1)
torch.cuda.synchronize()
V1 = V + 1
regions = [('att', B*K*H*4), ('p_att', B*K*H*4), ('x_all', T*B*H*4), ('h_all', (T+1)*B*H*4), ('c_all', (T+1)*B*H*4), ('att_h_all', T*B*H*4),
           ('att_res_all', T*B*H*4), ('alpha_all', T*B*K*4), ('dot_all', T*B*K*4), ('pre_all', T*B*5*H*4), ('out_all', T*B*H*4),
           ('logp_all', T*B*V1*4), ('bias_ih', 5*H*4), ('pre_img', B*5*H*4), ('zeros', B*H*4), ('it_all', (T+1)*B*4), ('unfinished', B*4),
           ('any_unf', (T+1)*4), ('att_bf', B*K*H*2), ('p_att_bf', B*K*H*2), ('part', 6*16384*4), ('lse_all', T*B*4),
           ('logit_parts', 3*V1*H*2), ('gate_parts', (5*H*H+5*H*H+H*H)*2), ('tsync', ((B+15)//16*T*3+1+3)//4*4*4)]
off = 0
print('L', int(a0['L']), int(a1['L']), int(b0['L']), int(b1['L']), 'ws bytes', a0['ws'].numel())
for name, nbytes in regions:
    off = (off + 255) // 256 * 256
    for tag, x, y in (('a', a0, a1), ('b', b0, b1)):
        xs, ys = x['ws'][off:off + nbytes], y['ws'][off:off + nbytes]
        nd = int((xs != ys).sum())
        if nd:
            idx = int((xs != ys).nonzero()[0])
            print(f'{tag} {name}: {nd} of {nbytes} bytes differ, first at +{idx} (element {idx // 4})')
    off += nbytes
print('end offset', (off + 255) // 256 * 256)
