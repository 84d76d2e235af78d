"""GPU: share_embed = 1 (AlternatingJointModel.py:83-88, train.py:390-391, optimizer.py:224-242) - ONE embedding table owned by
both agents and both Adam instances - against iterations recorded from the reference's own trainer functions
(tests/golden/share_*.npz, tools/gen_golden.py --only-share-embed): per iteration the loss, the decoded tokens, the gradient
digests and the digest of every weight after the update.  The weight trajectory is what pins the semantics: in the gumbel
configuration the table is stepped by BOTH optimizers (each with moments of its own) from the one gradient both agents
accumulated into; in the reinforce configuration it is stepped by the speaker's optimizer in speaker turns and is frozen in
listener turns (the caption model's requires_grad loop runs last)."""
import numpy as np
import pytest
import torch

import golden_util as GU
import share_util as SU

pytestmark = pytest.mark.gpu


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize('name', SU.SHARE_CASES)
def test_share_embed_trajectory_matches_reference(name):
    from cooperativeimagecaptioning_amd import models, optimizer as optim
    from cooperativeimagecaptioning_amd.misc import rewards
    z, cfg, w0 = SU.load(name)
    B = z['s0.fc_feats'].shape[0]
    opt = GU.make_opt(cfg, B, share_embed=1, learning_rate=cfg['learning_rate'], grad_clip=cfg['grad_clip'], weight_decay=0.0)
    rewards.init_scorer('corpus')
    model = models.AlternatingJointModel(opt)
    cg, vse = model.caption_generator, model.vse
    assert cg.embed[0].weight is vse.txt_enc.embed.weight
    sd = model.state_dict()
    assert SU.SPK_EMBED in sd and SU.LST_EMBED in sd                  # both names, as in the reference's checkpoints
    model.load_state_dict({k: T_(v) for k, v in w0.items()})
    model.cuda().train()
    model.tie_embeddings()                                            # train.py:390-391
    assert cg.embed[0].weight is vse.txt_enc.embed.weight
    opt.is_alternating = 1
    opt.alternating_turn = ['speaker', 'listener']
    od = optim.load_optimizer(model, opt)
    optim.fuse_zero_grad(od)                                          # as train.py: gradients cleared inside the Adam kernels
    for s in range(int(z['n_steps'])):
        d = SU.step_view(z, s)
        turn = d['turn']
        optimizer = od[turn] if turn in od else od['speaker']
        model.caption_generator.noise.override = SU.step_noise(d, cfg, turn)
        b = SU.step_batch(d)
        optim.zeroing_optimizer(opt, od, optimizer)
        loss = model(T_(b['fc_feats']).cuda(), T_(b['labels']).cuda(), T_(b['masks']).cuda(), {'gts': b['gts']},
                     T_(b['att_feats']).cuda(), None, is_alternating=True, alternating_turn=turn)
        loss.backward()
        torch.cuda.synchronize()
        np.testing.assert_allclose(float(loss.detach()), float(d['loss']), rtol=1e-4, atol=1e-6, err_msg=f'step {s} loss')
        ref_tok = d['tokens0']
        got = model.last_decodes['sample'].seq[:, :ref_tok.shape[1]].cpu().numpy()
        np.testing.assert_array_equal(got, ref_tok, err_msg=f'step {s} tokens')
        assert bool(vse.txt_enc.embed.weight.requires_grad) == bool(int(d['embed_requires_grad']))
        grads = {k: p.grad for k, p in model.named_parameters()}
        glob = max(float(np.abs(d[k][1]) / max(grads[k[5:]].numel(), 1)) for k in d if k.startswith('gdig.'))
        n = 0
        for k in d:
            if not k.startswith('gdig.') or float(np.abs(d[k][1])) == 0.0:
                continue
            frozen = cfg['retrieval_reward'] == 'reinforce' and k[5:].startswith('vse.') == (turn == 'speaker')
            if frozen:
                continue                       # the agent that does not step keeps stale, clamped-in-place gradients in the reference
            g = grads[k[5:]]
            dg = GU.digest(g.detach().cpu().numpy())
            scale = abs(d[k][1]) / max(g.numel(), 1)
            np.testing.assert_allclose(dg[2:], d[k][2:], rtol=5e-4, atol=5e-4 * scale + 1e-5 * glob, err_msg=f'step {s} {k}')
            n += 1
        assert n > 0
        optim.update_optimizer(od, optimizer, opt)
        torch.cuda.synchronize()
        assert cg.embed[0].weight is vse.txt_enc.embed.weight          # train.py:530-531 check_equal_embed_weights
        for k, v in model.state_dict().items():
            if k.endswith('alpha_net.bias'):    # gradient = rounding noise of a softmax shift; Adam turns noise into +-lr
                continue
            np.testing.assert_allclose(GU.digest(v.detach().cpu().numpy()), d['wdig.' + k], rtol=2e-5, atol=2e-6,
                                       err_msg=f'step {s} weights {k}')


def test_share_embed_optimizer_state_round_trip():
    """The speaker's optimizer state dict carries its OWN moments of the shared table (parameter order of
    caption_generator.parameters(), as torch.optim.Adam's), and loading it restores them."""
    from cooperativeimagecaptioning_amd import models, optimizer as optim
    from cooperativeimagecaptioning_amd.misc import rewards
    z, cfg, w0 = SU.load('share_joint_gumbel')
    B = z['s0.fc_feats'].shape[0]
    opt = GU.make_opt(cfg, B, share_embed=1, learning_rate=cfg['learning_rate'], grad_clip=cfg['grad_clip'], weight_decay=0.0)
    rewards.init_scorer('corpus')
    model = models.AlternatingJointModel(opt)
    model.load_state_dict({k: T_(v) for k, v in w0.items()})
    model.cuda().train()
    opt.is_alternating, opt.alternating_turn = 1, ['speaker', 'listener']
    od = optim.load_optimizer(model, opt)
    d = SU.step_view(z, 0)
    b = SU.step_batch(d)
    model.caption_generator.noise.override = SU.step_noise(d, cfg, 'speaker')
    optim.zeroing_optimizer(opt, od, od['speaker'])
    loss = model(T_(b['fc_feats']).cuda(), T_(b['labels']).cuda(), T_(b['masks']).cuda(), {'gts': b['gts']},
                 T_(b['att_feats']).cuda(), None, is_alternating=True, alternating_turn='speaker')
    loss.backward()
    optim.update_optimizer(od, od['speaker'], opt)
    spk = od['speaker']['speaker']
    sd = spk.state_dict()
    names = [n for n, _ in model.caption_generator.named_parameters()]
    i = names.index('embed.0.weight')
    assert i in sd['state'] and float(sd['state'][i]['exp_avg'].abs().max()) > 0
    fresh = optim.define_optimizer(model.caption_generator, opt)
    fresh.load_state_dict(sd)
    st = fresh._ext_state['embed.0.weight']
    assert torch.equal(st[0].view(-1), spk._ext_state['embed.0.weight'][0].view(-1))
    assert fresh.flat.step == spk.flat.step
