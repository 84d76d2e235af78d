"""AlternatingJointModel with the reference's constructor signature, call signature, loss-flag
logic and state-dict layout (models/AlternatingJointModel.py:71-686), with every tensor
computation delegated to the HIP engines.

One call = one joint training step's forward: sampled decode -> listener on the generated
(straight-through) captions -> greedy decode -> CIDEr-D reward -> loss; ``loss.backward()``
launches the backward engines.  Nothing synchronises with the host inside the step (the
reference syncs once per decode step, AttModel.py:407).
"""
import os

import torch
import torch.nn as nn

from .. import engine
from ..autograd_glue import EngineLoss, engine_loss
from ..misc import utils
from ..misc import rewards


class AlternatingJointModel(nn.Module):
    def __init__(self, opt, iteration=None):
        super().__init__()
        from . import setup, load
        self.opt = opt
        self.use_word_weights = getattr(opt, 'use_word_weights', 0)
        self.caption_generator = setup(opt, opt.caption_model, 'caption_model')
        if opt.vse_model != 'None':
            self.vse = setup(opt, opt.vse_model, 'vse_model')
            self.share_embed = opt.share_embed
            if self.share_embed:
                self.tie_embeddings()                                   # :83-88
        else:
            self.vse = None
            self.share_embed = 0
        if opt.retrieval_reward == 'reinforce':                       # :95-98
            if opt.vse_loss_weight == 0 and isinstance(self.vse, nn.Module):
                for p in self.vse.parameters():
                    p.requires_grad = False
        self.batch_size = opt.batch_size
        self.vse_loss_weight = opt.vse_loss_weight
        self.caption_loss_weight = opt.caption_loss_weight
        self.df = getattr(opt, 'df', 'coco-val')
        self.retrieval_reward = opt.retrieval_reward
        if opt.alternating_turn is not None:                          # :110-114
            if len(opt.alternating_turn) == 1 and opt.retrieval_reward == 'reinforce':
                if opt.alternating_turn[0] == 'listener':
                    opt.retrieval_reward_weight = 0
        self.retrieval_reward_weight = opt.retrieval_reward_weight
        self.reinforce_baseline_type = getattr(opt, 'reinforce_baseline_type', 'greedy')
        self.sheriff_baseline_type = getattr(opt, 'sheriff_baseline_type', 'greedy')
        self.only_one_retrieval = getattr(opt, 'only_one_retrieval', 'off')
        self.cider_optimization = getattr(opt, 'cider_optimization', 0)
        self.use_gen_cider_scores = getattr(opt, 'use_gen_cider_scores', 0)
        self._loss = {}
        self.last_decodes = {}
        # MI355X: the greedy (baseline) decode and the sampled decode of a step read the same images and weights
        # and are independent until the reward, so they advance in lock step through shared launches
        # (cic_speaker_decode_fwd_pair: every per-timestep kernel once over 2B rows; bit-identical results).
        self.pair_decodes = True
        # data-parallel runs: optimizer.overlap_gradient_exchange() sets this to the listener optimizer's
        # begin_all_reduce, called from backward() as soon as the listener's gradient is final
        self.listener_grads_ready = None
        # ... and this to the speaker optimizer's begin_all_reduce('logit'): called inside the last speaker backward
        # engine of a step, right after the logit layer's gradient is final (before the BPTT loop)
        self.speaker_logit_grads_ready = None
        # Load model (:131-177)
        if opt.is_alternating:
            if getattr(opt, 'continue_from_existing_models', False):
                path = None
                if opt.start_from and os.path.isfile(os.path.join(opt.start_from, 'alternatingModel.pth')):
                    name = 'alternatingModel-' + iteration + '.pth' if iteration else 'alternatingModel.pth'
                    path = os.path.join(opt.start_from, name)
                    msg = f'Loaded alternating model from {path}'
                else:
                    path = opt.speaker_stage_2_model_path
                    msg = f'Loaded pre-trained "speaker" model, after stage 2 from {path}'
                if path and os.path.isfile(str(path)):
                    utils.load_state_dict(self, torch.load(path, map_location='cpu', weights_only=True))
                    print(msg)
                else:
                    print(f'no pre-trained model at {path!r}: starting the joint model from its random initialisation')
        else:
            load(self, opt, iteration)
            if getattr(opt, 'initialize_retrieval', None) is not None:
                print("Make sure the vse opt are the same !!!!!")
                sd = torch.load(opt.initialize_retrieval, map_location='cpu', weights_only=True)
                utils.load_state_dict(self, {k: v for k, v in sd.items() if 'vse.' in k})

    def tie_embeddings(self):
        """share_embed = 1 (:83-88; train.py:390-391 repeats it after the move to the GPU): the listener's embedding module
        becomes the speaker's token embedding - ONE Parameter, reachable under both state-dict names, owned by both Adam
        instances.  MI355X layout: the table lives in the LISTENER's flat buffers; the speaker's FlatAgent lists it as external
        (flat.FlatAgent), its engines read it and add their gradient into the listener's gradient segment, and the speaker's
        FlatAdam steps it with moments of its own (optimizer.FlatAdam)."""
        cg = self.caption_generator
        if not isinstance(getattr(cg, 'embed', None), nn.Sequential):
            raise NotImplementedError("share_embed = 1 needs the att2in2 speaker: the reference assigns caption_generator.embed[0] "
                                      "(AlternatingJointModel.py:85), which FCModel's bare nn.Embedding does not support")
        if cg.embed[0] is not self.vse.txt_enc.embed:
            if tuple(cg.embed[0].weight.shape) != tuple(self.vse.txt_enc.embed.weight.shape):
                raise ValueError('share_embed = 1 needs input_encoding_size to be the same for both agents')
            cg.embed[0] = self.vse.txt_enc.embed
            cg._flat = None                                             # (a layout made before the tie is stale)
        cg._external = ('embed.0.weight',)
        object.__setattr__(cg, '_external_owner', self.vse)         # (a plain attribute: not a registered sub-module of the speaker)
        if getattr(self.opt, 'phase', None) == 2:                        # second phase (MLE) only: the table is frozen
            for p in cg.embed.parameters():
                p.requires_grad = False

    # ---- flags (:180-194) ---------------------------------------------------------------------
    def getLossFlags(self):
        return [self.vse_loss_weight, self.caption_loss_weight, self.cider_optimization, self.retrieval_reward_weight]

    def setLossFlages(self, VSEWeight, MLEWeight, ciderFlag, DISCWeight):
        self.vse_loss_weight = VSEWeight
        self.caption_loss_weight = MLEWeight
        self.cider_optimization = ciderFlag
        self.retrieval_reward_weight = DISCWeight

    def changeModelUpdateStatus(self, gradDic, printWeights=False):
        """:571-685 — only the requires_grad effect; the deep-copy/compare diagnostics of the reference
        (26 M parameters copied per turn) are not reproduced."""
        if 'vseModel' in gradDic and self.vse is not None:
            for p in self.vse.parameters():
                p.requires_grad = gradDic['vseModel']
        if 'captionModel' in gradDic:
            for p in self.caption_generator.parameters():
                p.requires_grad = gradDic['captionModel']

    # ---- the step -------------------------------------------------------------------------------
    def _refs(self, data, device):
        """Reference captions of the batch packed for the CIDEr-D kernels.  Packing + upload happen once per batch
        object (a loader may also attach data['_cic_refs'] itself, from pinned memory): a pageable H2D copy in the
        middle of the step would stall the host until the stream drains."""
        hit = data.get('_cic_refs') if isinstance(data, dict) else None
        if hit is not None and hit[0] is data['gts'] and hit[1].device == torch.device(device):
            return hit[1], hit[2]
        refs, ref_off = engine.pack_refs(data['gts'], device)
        if isinstance(data, dict):
            data['_cic_refs'] = (data['gts'], refs, ref_off)
        return refs, ref_off

    def _plain_forward(self, fc_feats, seq, masks, data, att_feats, att_masks, gen_override=None):
        """The non-alternating branch, :443-504.  Returns a 0-dim loss whose backward() runs the
        backward engines."""
        cg, vse = self.caption_generator, self.vse
        dev = fc_feats.device
        rr = self.retrieval_reward
        oor = self.only_one_retrieval
        spk_grad = any(p.requires_grad for p in cg.parameters()) and torch.is_grad_enabled()
        lst_grad = vse is not None and any(p.requires_grad for p in vse.parameters()) and torch.is_grad_enabled()
        from .FCModel import FCModel
        fc_speaker = isinstance(cg, FCModel)
        if fc_speaker:
            # BASELINE configs[0]: the fc-feature speaker.  Its sample() returns (seq, logprobs) only, so the reference can
            # drive it with the MLE, VSE, REINFORCE (reinforce_disc :226-247) and CIDEr terms, not the straight-through ones
            if self.retrieval_reward_weight > 0 and rr != 'reinforce':
                raise NotImplementedError(f"caption_model 'fc' under retrieval_reward {rr!r}: FCModel.sample returns two "
                                          "values, the reference's straight-through path unpacks three (:346)")
            cg._step_fc = fc_feats
        B = fc_feats.shape[0]
        T = cg.seq_length
        terms = []          # (weight, device scalar)
        bwd_steps = []      # closures run in order by backward(go)
        att_pre = cg.att_embed_pre(att_feats, att_masks)
        cw, vw, dw, ciw = self.caption_loss_weight, self.vse_loss_weight, self.retrieval_reward_weight, self.cider_optimization

        # MLE (ce_loss :196-207)
        if cw > 0:
            Tm = seq.shape[1] - 1
            mle = cg.decode(att_feats, att_masks, 'teacher', 1.0, att_pre=att_pre, grad=True, T=Tm,
                            pick=seq.t().contiguous().long(), first_token=seq[:, 0].contiguous().long(), tag='mle',
                            decoding_constraint=0, want_stv=False, ss_prob=cg.ss_prob if cg.training else 0.0)
            d_mle = cg._buf.get('d_mle', (B, Tm), torch.float32, dev)
            l_mle = engine.masked_nll(mle.slp, masks.float()[:, 1:], cw, dslp=d_mle)
            cg._loss['xe'] = l_mle.detach()[0]
            self._loss['loss_cap'] = l_mle.detach()[0]
            terms.append((cw, l_mle))
            if spk_grad:
                def bwd_mle(go, logit_ready=None):
                    cg.decode_backward(mle, dslp=d_mle, dslp_scale=go, logit_grads_ready=logit_ready)
                bwd_mle.is_speaker = True
                bwd_steps.append(bwd_mle)
        # VSE on ground-truth captions (vse_loss :209-224)
        if vw > 0:
            gt = vse.run(fc_feats, labels=seq, masks=masks, only_one_retrieval=oor, slot=1)
            self._loss['loss_vse'] = gt.loss_sum.detach()[0]
            vse._loss['contrastive'] = gt.loss_sum.detach()[0]
            terms.append((vw, gt.loss_sum))
            if lst_grad:
                def bwd_gt(go, gt=gt):
                    vse.run_backward(gt, g_scalar=(go * vw).reshape(1).contiguous())
                bwd_gt.is_listener = True
                bwd_steps.append(bwd_gt)

        sample = None
        greedy = None
        dslp = None
        fused_total = None
        need_greedy = bool(ciw) or (dw > 0 and rr == 'reinforce' and self.reinforce_baseline_type == 'greedy')

        def sampled_with_greedy(**spec):
            """The sampled decode of the step, together with the greedy decode when the step needs one."""
            if need_greedy and greedy is None and self.pair_decodes and not fc_speaker:
                return cg.decode_pair(att_feats, att_masks, spec, dict(mode='greedy', tag='greedy'), att_pre=att_pre)
            return cg.decode(att_feats, att_masks, att_pre=att_pre, **spec), greedy
        if dw > 0:                                                     # DISC loss :455-488
            if rr == 'reinforce':
                sample, greedy = sampled_with_greedy(mode='multinomial', temp=1.0, grad=spk_grad)   # :226-247
                gen = vse.run(fc_feats, decode=sample, only_one_retrieval=oor, slot=2)
                btype = self.reinforce_baseline_type
                if btype == 'greedy':                                  # :250-298
                    if greedy is None:
                        greedy = cg.decode(att_feats, att_masks, 'greedy', att_pre=att_pre, tag='greedy')
                    base = vse.run(fc_feats, decode=greedy, only_one_retrieval=oor, slot=3).loss_rows
                elif btype == 'gt':                                    # :300-310
                    base = vse.run(fc_feats, labels=seq, masks=masks, only_one_retrieval=oor, slot=3).loss_rows
                else:                                                  # :312-319
                    base = torch.zeros(B, device=dev)
                coef = (gen.loss_rows - base).contiguous()
                dslp = cg._buf.get('dslp', (B, T), torch.float32, dev)
                sc = engine.seq_loss(sample.slp, sample.seq, sample.L, coef, 1.0, dw, dslp=dslp)   # :321-325
                terms.append((dw, sc))
                self._loss['retrieval_sc_loss'] = sc.detach()[0]
                self._loss['retrieval_loss'] = gen.loss_rows.sum().detach()
                self._loss['retrieval_loss_greedy'] = base.sum().detach()
            elif rr in ('gumbel', 'multinomial', 'gumbel_softmax', 'multinomial_soft'):   # st_and_ps_methods :343-376
                mode = {'gumbel': 'gumbel', 'multinomial': 'multinomial_st', 'gumbel_softmax': 'gumbel_ps',
                        'multinomial_soft': 'multinomial_ps'}[rr]
                temp = cg.gumbel_temp if rr in ('gumbel', 'gumbel_softmax') else cg.multinomial_temp
                ps_prob = {'gumbel_softmax': cg.prob_gumbel_softmax, 'multinomial_soft': cg.prob_multinomial_soft}.get(rr, 0.0)
                if mode.endswith('_ps'):     # soft-input steps: not row-wise launches, decoded on its own
                    sample = cg.decode(att_feats, att_masks, mode, temp, att_pre=att_pre, grad=spk_grad, ps_prob=ps_prob)
                else:
                    sample, greedy = sampled_with_greedy(mode=mode, temp=temp, grad=spk_grad)
                gen = vse.run(fc_feats, decode=sample, only_one_retrieval=oor, slot=2)
                vse._loss['contrastive'] = gen.loss_sum.detach()[0]
                terms.append((dw, gen.loss_sum))
                d_onehot = cg._buf.get('d_onehot', (T, B, cg.vocab_size + 1), torch.float32, dev) if spk_grad else None

                def bwd_listener(go, gen=gen, d_onehot=d_onehot):
                    vse.run_backward(gen, g_scalar=go.reshape(1), g_scale=dw, param_grads=lst_grad, d_onehot=d_onehot)
                bwd_listener.is_listener = True
                if spk_grad or lst_grad:
                    bwd_steps.append(bwd_listener)
                sample.d_onehot = d_onehot
                if sample.soft is not None and spk_grad:
                    # partial sampling: the CIDEr term below draws its own captions (:378-389), so this decode's
                    # backward (listener gradient only) is queued here
                    def bwd_ps(go, logit_ready=None, ps=sample, d=d_onehot):
                        cg.decode_backward(ps, d_onehot=d, logit_grads_ready=logit_ready)
                    bwd_ps.is_speaker = True
                    bwd_steps.append(bwd_ps)
            else:
                raise ValueError(f'unknown retrieval_reward {rr!r}')
        if ciw:                                                        # CIDEr loss :490-503
            if sample is None or rr in ('multinomial_soft', 'gumbel_softmax'):
                sample, greedy = sampled_with_greedy(mode='multinomial', temp=1.0, grad=spk_grad,
                                                     tag='cider_gen')  # gen_result_for_cider :378-389
            if greedy is None:
                greedy = cg.decode(att_feats, att_masks, 'greedy', att_pre=att_pre, tag='greedy')   # :391-403
            refs, ref_off = self._refs(data, dev)
            rw = rewards.get_self_critical_reward_device(refs, ref_off, sample, greedy)
            coef = rw['scores'][:B].float().contiguous() if self.use_gen_cider_scores else rw['reward']
            fresh = dslp is None
            if fresh:
                dslp = cg._buf.get('dslp', (B, T), torch.float32, dev)
            # the CIDEr term is the last of the step's sum: its kernel writes the step's loss as well (no loss_combine launch)
            lc, fused_total = engine.seq_loss(sample.slp, sample.seq, sample.L, coef, -1.0, ciw, dslp=dslp,
                                              accumulate=not fresh, combine=(list(terms), ciw))
            terms.append((ciw, lc))
            # mean of coef from the reward kernel's own sums (stats = mean sampled score, mean greedy score): no launch here
            st = rw['stats']
            self._loss['avg_reward'] = (lambda st=st: st[0]) if self.use_gen_cider_scores else (lambda st=st: st[0] - st[1])
            self._loss['cider_greedy'] = rw['stats'][1].detach()
            self._loss['loss_cider'] = lc.detach()[0]
        # the decodes of this step (DecodeResult or None), for callers that want the generated tokens (evaluation, tests)
        self.last_decodes = {'sample': sample, 'greedy': greedy}
        if sample is not None and sample.soft is not None:
            sample = None            # a partial-sampling decode without a CIDEr term: already queued above
        if sample is not None and spk_grad and (dslp is not None or getattr(sample, 'd_onehot', None) is not None):
            def bwd_speaker(go, logit_ready=None, sample=sample, dslp=dslp):
                cg.decode_backward(sample, d_onehot=getattr(sample, 'd_onehot', None), dslp=dslp,
                                   dslp_scale=go if dslp is not None else None, logit_grads_ready=logit_ready)
            bwd_speaker.is_speaker = True
            bwd_steps.append(bwd_speaker)

        if not terms:
            return self._zero_loss(dev)
        # sum_i weight_i * term_i: by the last term's own kernel when that is the CIDEr term, else one launch
        loss = fused_total if fused_total is not None else engine.loss_combine(terms)
        anchor = next((p for p in self.parameters() if p.requires_grad), None)
        if anchor is None or not torch.is_grad_enabled() or not bwd_steps:
            return loss.detach()

        # the listener's gradient is final after the last step that runs a listener backward engine
        last_lst = max([i for i, st in enumerate(bwd_steps) if getattr(st, 'is_listener', False)], default=-1)
        # ... and the speaker's logit-layer gradient inside the LAST speaker backward, before its BPTT loop
        last_spk = max([i for i, st in enumerate(bwd_steps) if getattr(st, 'is_speaker', False)], default=-1)

        def backward(go):
            for i, step in enumerate(bwd_steps):
                if i == last_spk and self.speaker_logit_grads_ready is not None:
                    step(go, logit_ready=self.speaker_logit_grads_ready)
                else:
                    step(go)
                if i == last_lst and lst_grad and self.listener_grads_ready is not None:
                    self.listener_grads_ready()
        return engine_loss(loss, anchor, backward)

    def forward(self, fc_feats, seq, masks, data, att_feats, att_masks, is_alternating=False, alternating_turn=None):
        """:433-555."""
        if not is_alternating:
            return self._plain_forward(fc_feats, seq, masks, data, att_feats, att_masks)
        oldVSE, oldMLE, oldCider, oldDISC = self.getLossFlags()
        try:
            if alternating_turn == 'speaker':                          # :508-526
                if self.retrieval_reward == 'reinforce':
                    self.changeModelUpdateStatus({'vseModel': False, 'captionModel': True})
                self.setLossFlages(VSEWeight=0, MLEWeight=oldMLE, ciderFlag=oldCider, DISCWeight=oldDISC)
                return self._plain_forward(fc_feats, seq, masks, data, att_feats, att_masks)
            elif alternating_turn == 'listener':                       # :528-555
                self.changeModelUpdateStatus({'vseModel': True, 'captionModel': False})
                self.setLossFlages(VSEWeight=oldVSE, MLEWeight=0, ciderFlag=0, DISCWeight=0)
                cg = self.caption_generator
                gen = cg.decode(att_feats, att_masks, 'multinomial', 1.0, grad=False)
                self.last_decodes = {'sample': gen, 'greedy': None}
                return self._listener_on_generated(fc_feats, gen)
            raise ValueError(f'unknown alternating_turn {alternating_turn!r}')
        finally:
            self.setLossFlages(VSEWeight=oldVSE, MLEWeight=oldMLE, ciderFlag=oldCider, DISCWeight=oldDISC)

    def _zero_loss(self, device):
        """A zero loss the caller can still .backward() (the reference's `0 * loss` keeps its grad_fn and yields zero
        gradients, e.g. the listener turn with vse_loss_weight 0)."""
        z = torch.zeros((), device=device)
        anchor = next((p for p in self.parameters() if p.requires_grad), None)
        if anchor is None or not torch.is_grad_enabled():
            return z
        return EngineLoss.apply(z, anchor, lambda go: None)

    def _listener_on_generated(self, fc_feats, gen):
        """Listener turn: VSE loss on sampled captions fed as plain indices (:539-551)."""
        vse = self.vse
        vw = self.vse_loss_weight
        if not vw > 0:
            return self._zero_loss(fc_feats.device)
        gen.stv = None                                               # plain index input, no straight-through values
        res = vse.run(fc_feats, decode=gen, only_one_retrieval=self.only_one_retrieval, slot=2)
        self._loss['loss_vse'] = res.loss_sum.detach()[0]
        vse._loss['contrastive'] = res.loss_sum.detach()[0]
        loss = vw * res.loss_sum[0]
        anchor = next((p for p in vse.parameters() if p.requires_grad), None)
        if anchor is None or not torch.is_grad_enabled():
            return loss.detach()
        return engine_loss(loss, anchor, lambda go: vse.run_backward(res, g_scalar=(go * vw).reshape(1).contiguous()))

    def sample(self, fc_feats, att_feats, att_masks, opt={}):
        return self.caption_generator.sample(fc_feats, att_feats, att_masks, opt)

    def loss(self):
        """:562-568."""
        out = {}
        out.update({k: (v() if callable(v) else v) for k, v in self._loss.items()})   # lazily computed logging values
        out.update({'cap_' + k: v for k, v in self.caption_generator._loss.items()})
        if self.vse is not None:
            out.update({'vse_' + k: v for k, v in self.vse._loss.items()})
        return out
