import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cooperativeimagecaptioning_amd import engine, _lib
B, K, D, H, V, T = 32, 36, 64, 512, 9487, 16
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
engine.speaker_decode_launch(d, params, a0); engine.speaker_decode_launch(d, params, b0)
a2, b2 = specs()
engine.speaker_decode_launch(d, params, a2); engine.speaker_decode_launch(d, params, b2)
a1, b1 = specs()
engine.speaker_decode_fwd_pair(d, params, a1, b1)
torch.cuda.synchronize()
V1 = V + 1
secs = [('att', B*K*H), ('p_att', B*K*H), ('x', T*B*H), ('h', (T+1)*B*H), ('c', (T+1)*B*H), ('att_h', T*B*H), ('att_res', T*B*H),
        ('alpha', T*B*K), ('dot', T*B*K), ('pre', T*B*5*H), ('out', T*B*H), ('logp', T*B*V1), ('bias_ih', 5*H)]
def walk(x, y, tag):
    off = 0
    for name, n in secs:
        off = (off + 255) & ~255
        u = x['ws'][off:off + 4*n].view(torch.float32); v = y['ws'][off:off + 4*n].view(torch.float32)
        off += 4*n
        df = (u - v).abs()
        if float(df.max()) > 0:
            idx = int(df.argmax()); per_t = n // (T if name not in ('att','p_att','bias_ih','h','c') else 1)
            print(tag, name, 'maxdiff', float(df.max()), 'first diff elem', int((df > 0).nonzero()[0]), 'n', n, 'ndiff', int((df>0).sum()))
print('seq vs seq (determinism):'); walk(a0, a2, 'a'); walk(b0, b2, 'b')
print('seq vs pair:'); walk(a0, a1, 'a'); walk(b0, b1, 'b')
