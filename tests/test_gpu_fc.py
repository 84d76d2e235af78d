"""GPU: FCModel (models/FCModel.py, the fc-feature speaker of BASELINE configs[0]) through the mirrored module
API on the HIP decode engine (fc_mode), replaying the fixtures recorded from the reference: MLE loss 5e-5
relative and every parameter gradient 5e-4 of its scale; greedy / multinomial decodes token-exact."""
import numpy as np
import pytest
import torch

import golden_util as GU

pytestmark = pytest.mark.gpu


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build(z, B):
    from cooperativeimagecaptioning_amd import models
    cfg = GU.cfg_dict(z)
    opt = GU.make_opt(cfg, B, caption_model='fc')
    m = models.setup(opt, 'fc', 'caption_model')
    m.load_state_dict({k: T_(v) for k, v in z['weights'].items()})
    return m.cuda().train(), cfg


@pytest.mark.parametrize('name', ['fc_mle', 'fc_mle_dropout'])
def test_fc_mle_forward_backward_matches_reference(name):
    z = GU.load_case(name)
    m, cfg = build(z, z['fc'].shape[0])
    m.noise.override = {'mle': GU.noise_dict(z, 'noise')}
    m.zero_grad()
    loss = m(T_(z['fc']).cuda(), None, None, T_(z['labels']).cuda(), T_(z['masks']).cuda())
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z['loss'][0]), rtol=5e-5)
    grads = {k: p.grad for k, p in m.named_parameters()}
    glob = max(float(np.abs(z[k][1]) / max(grads[k[5:]].numel(), 1)) for k in z if k.startswith('gdig.'))
    n = 0
    for k in z:
        if k.startswith('gdig.'):
            g = grads[k[5:]]
            d = GU.digest(g.detach().cpu().numpy())
            scale = abs(z[k][1]) / max(g.numel(), 1)
            np.testing.assert_allclose(d[2:], z[k][2:], rtol=5e-4, atol=5e-4 * scale + 1e-5 * glob, err_msg=k)
            np.testing.assert_allclose(d[1], z[k][1], rtol=5e-4, atol=1e-5 * glob * g.numel(), err_msg=k + ' abs-sum')
            n += 1
    assert n == 9


@pytest.mark.parametrize('name', ['fc_sample_greedy', 'fc_sample_greedy_dropout', 'fc_sample_multinomial',
                                  'fc_sample_multinomial_temp'])
def test_fc_sample_matches_reference(name):
    z = GU.load_case(name)
    m, cfg = build(z, z['fc'].shape[0])
    smax = int(z['opt.sample_max'])
    m.noise.override = {('greedy' if smax == 1 else 'sample'): GU.noise_dict(z, 'noise')}
    with torch.no_grad():
        seq, slp = m.sample(T_(z['fc']).cuda(), None, None, {'sample_max': smax, 'temperature': float(z['opt.temperature'])})
    np.testing.assert_array_equal(seq.cpu().numpy(), z['res0'])                   # token ids bit-exact
    np.testing.assert_allclose(slp.cpu().numpy(), z['res1'], rtol=5e-5, atol=5e-5)


def test_fc_cpu_input_fails_loudly():
    from cooperativeimagecaptioning_amd import _lib
    z = GU.load_case('fc_mle')
    m, _ = build(z, z['fc'].shape[0])
    with pytest.raises(_lib.CicError):
        m(T_(z['fc']), None, None, T_(z['labels']).cuda(), T_(z['masks']).cuda())
