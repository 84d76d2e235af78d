"""Speaker: Att2in2Model with the reference's constructor signature, attributes and
state-dict names (models/AttModel.py:53-94,456-463,492-508,534-539), computing through the
HIP engines of libcic_hip.so instead of torch ops.

The nn.Modules below are parameter containers only (so that ``state_dict()`` has exactly the
reference's keys and the default initialisers draw the same numbers for the same seed);
none of their ``forward`` methods is on the path.
"""
import torch
import torch.nn as nn

from .. import _lib, engine
from ..flat import FlatAgent
from ..noise import NoiseSource
from ..bufcache import BufCache
from ..autograd_glue import engine_loss


class Attention(nn.Module):
    """Parameter container of models/AttModel.py:456-463."""

    def __init__(self, opt):
        super().__init__()
        self.rnn_size = opt.rnn_size
        self.att_hid_size = opt.att_hid_size
        self.h2att = nn.Linear(self.rnn_size, self.att_hid_size)
        self.alpha_net = nn.Linear(self.att_hid_size, 1)


class Att2in2Core(nn.Module):
    """Parameter container of models/AttModel.py:492-508."""

    def __init__(self, opt):
        super().__init__()
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.drop_prob_lm = opt.drop_prob_lm
        self.a2c = nn.Linear(self.rnn_size, 2 * self.rnn_size)
        self.i2h = nn.Linear(self.input_encoding_size, 5 * self.rnn_size)
        self.h2h = nn.Linear(self.rnn_size, 5 * self.rnn_size)
        self.dropout = nn.Dropout(self.drop_prob_lm)
        self.attention = Attention(opt)


MODES = {'greedy': _lib.SAMPLE_GREEDY, 'multinomial': _lib.SAMPLE_MULTINOMIAL,
         'gumbel': _lib.SAMPLE_GUMBEL_ST, 'multinomial_st': _lib.SAMPLE_MULTINOMIAL_ST,
         'teacher': _lib.SAMPLE_TEACHER, 'gumbel_ps': _lib.SAMPLE_GUMBEL_PS,
         'multinomial_ps': _lib.SAMPLE_MULTINOMIAL_PS}


class DecodeResult:
    """What one decode leaves on the device (no host sync): seq/slp/stv are padded to
    seq_length columns, L is the number of columns the reference would have returned."""

    def __init__(self, fwd, mode, dims, params, att_raw, grad):
        self.fwd, self.mode, self.dims, self.params, self.att_raw, self.grad = fwd, mode, dims, params, att_raw, grad
        self.seq, self.slp, self.stv, self.L = fwd['seq'], fwd['slp'], fwd['stv'], fwd['L']
        self.soft = fwd.get('soft')       # [T,B,V+1] partial-sampling caption rows (time-major) or None


class AttModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.vocab_size = opt.vocab_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.num_layers = opt.num_layers
        self.drop_prob_lm = opt.drop_prob_lm
        self.seq_length = opt.seq_length
        self.fc_feat_size = opt.fc_feat_size
        self.att_feat_size = opt.att_feat_size
        self.att_hid_size = opt.att_hid_size
        self.retrieval_reward = opt.retrieval_reward
        self.gumbel_temp = opt.gumbel_temp
        self.multinomial_temp = opt.multinomial_temp
        self.prob_gumbel_softmax = getattr(opt, 'prob_gumbel_softmax', 1)
        self.prob_multinomial_soft = getattr(opt, 'prob_multinomial_soft', 1)
        self.use_bn = getattr(opt, 'use_bn', 0)
        if self.num_layers != 1:
            raise NotImplementedError('att2in2 is a single-layer maxout LSTM (models/AttModel.py:492-531)')
        self.ss_prob = 0.0
        # same construction order as the reference so that the same seed draws the same weights
        self.embed = nn.Sequential(nn.Embedding(self.vocab_size + 2, self.input_encoding_size), nn.ReLU(),
                                   nn.Dropout(self.drop_prob_lm))
        self.relu_dropout = nn.Sequential(nn.ReLU(), nn.Dropout(self.drop_prob_lm))
        _ = nn.Linear(self.fc_feat_size, self.rnn_size)   # fc_embed: created then deleted by Att2in2Model (:538)
        # use_bn: BatchNorm1d over the packed valid region rows in front of the Linear (AttModel.py:82-85): state-dict keys
        # att_embed.0 = the norm (weight, bias, running_mean, running_var, num_batches_tracked), att_embed.1 = the Linear
        self.att_embed = nn.Sequential(*(((nn.BatchNorm1d(self.att_feat_size),) if self.use_bn else ()) +
                                         (nn.Linear(self.att_feat_size, self.rnn_size), nn.ReLU(),
                                          nn.Dropout(self.drop_prob_lm))))
        self.logit = nn.Linear(self.rnn_size, self.vocab_size + 1)
        self.ctx2att = nn.Linear(self.rnn_size, self.att_hid_size)
        self.decoding_constraint = getattr(opt, 'decoding_constraint', 0)
        # 'f32': the reference's arithmetic.  'bf16': the reduced-precision variant of BASELINE configs[1] (cic.h,
        # cic_speaker_dims.compute_dtype): bf16 operands in the batched products, bf16 storage of the region features
        self.compute_dtype = getattr(opt, 'compute_dtype', 'f32') or 'f32'
        if self.use_bn and self.compute_dtype != 'f32':
            raise NotImplementedError('use_bn=1 is built for compute_dtype f32 only')
        self._loss = {}
        self._flat = None
        self.noise = NoiseSource()
        self._ws = {}
        self._buf = BufCache()
        self.timer = None            # engine.KernelTimer: in-step kernel timing of this model's decodes (bench.py)
        # data-parallel runs (optimizer.overlap_gradient_exchange): makes the stream wait for gradient exchanges in flight; called
        # right before the one-launch BPTT loop, which needs every CU for itself
        self.exchange_barrier = None

    # ---- engine plumbing -----------------------------------------------------------------
    def flat(self):
        owner = getattr(self, '_external_owner', None)
        owner_flat = owner.flat() if owner is not None else None   # share_embed: the shared table's storage and gradient view are the listener's
        if self._flat is None:
            # the logit layer last: a contiguous early bucket of the data-parallel gradient exchange (flat.py)
            self._flat = FlatAgent(self, tail=('logit.weight', 'logit.bias'), external=getattr(self, '_external', ()))
        self._flat.ext_owner_flat = owner_flat
        self._flat.ensure()
        return self._flat

    def _dims(self, B, K, T):
        return engine.speaker_dims(B, K, self.att_feat_size, self.rnn_size, self.input_encoding_size,
                                   self.att_hid_size, self.vocab_size, T, self.drop_prob_lm, self.compute_dtype)

    def _check_inputs(self, att_feats):
        if not att_feats.is_cuda:
            raise _lib.CicError('cooperativeimagecaptioning_amd runs on the GPU only: att_feats is on ' +
                                str(att_feats.device) + ' (there is no CPU fallback path)')
        assert att_feats.dim() == 3 and att_feats.shape[2] == self.att_feat_size

    # ---- use_bn: BatchNorm1d folded into att_embed's Linear (csrc/batchnorm.hip) ------------------
    def _bn_buffers(self, dev):
        D, H = self.att_feat_size, self.rnn_size
        g = lambda k, shape: self._buf.get(('bn', k), shape, torch.float32, dev)
        return dict(mean=g('mean', (D,)), var=g('var', (D,)), count=g('count', (1,)), Wf=g('Wf', (H, D)), bf=g('bf', (H,)),
                    dW=g('dW', (H, D)), db=g('db', (H,)))

    def _bn_masks(self, att_feats, att_masks):
        if att_masks is None:
            # the reference feeds a [B,K,D] tensor to BatchNorm1d then (pack_wrapper, AttModel.py:44-51): a shape error
            raise ValueError('use_bn=1 normalises the packed valid region rows: att_masks is required')
        return self._buf.stage('att_masks', att_masks, torch.float32)

    def _bn_stats(self, att_raw, att_masks):
        """Statistics of this batch's valid region rows (training mode) or the running ones (eval), then the fold."""
        bn, lin = self.att_embed[0], self.att_embed[1]
        b = self._bn_buffers(att_raw.device)
        rows = att_raw.shape[0] * att_raw.shape[1]
        if self.training:
            engine.bn_stats(att_raw, att_masks, rows, self.att_feat_size, b['mean'], b['var'], b['count'])
            mean, var = b['mean'], b['var']
        else:
            mean, var = bn.running_mean, bn.running_var
        engine.bn_fold_fwd(lin.weight.data, lin.bias.data, bn.weight.data, bn.bias.data, mean, var, bn.eps, b['Wf'], b['bf'])
        self._bn_used = (mean, var)

    def _bn_tick(self):
        """One reference forward in training mode = one update of the running statistics (every decode of a step runs
        att_embed again there; here they share one evaluation)."""
        if self.use_bn and self.training:
            bn, b = self.att_embed[0], self._bn_buffers(self.att_embed[1].weight.device)
            engine.bn_running_update(b['mean'], b['var'], b['count'], bn.momentum, bn.running_mean, bn.running_var)
            bn.num_batches_tracked += 1

    def _speaker_tensors(self, fl):
        """The engine's parameter pointers under the reference's non-bn names; use_bn: att_embed = the folded Linear."""
        t = fl.tensors()
        if not self.use_bn:
            return t
        b = self._bn_buffers(self.att_embed[1].weight.device)
        t = dict(t)
        t['att_embed.0.weight'], t['att_embed.0.bias'] = b['Wf'], b['bf']
        return t

    def _speaker_grads(self, fl):
        """Gradient targets of one backward; use_bn: the Linear's raw gradients land in zeroed scratch (_bn_backward)."""
        g = fl.grad_tensors()
        if not self.use_bn:
            return g
        b = self._bn_buffers(self.att_embed[1].weight.device)
        self._bn_targets = (g['att_embed.1.weight'], g['att_embed.1.bias'], g['att_embed.0.weight'], g['att_embed.0.bias'])
        g = dict(g)
        if self._bn_targets[0] is None:                  # a frozen speaker: no att_embed gradient at all (null pointers)
            g['att_embed.0.weight'] = g['att_embed.0.bias'] = None
            return g
        b['dW'].zero_(), b['db'].zero_()
        g['att_embed.0.weight'], g['att_embed.0.bias'] = b['dW'], b['db']
        return g

    def _bn_backward(self):
        if not self.use_bn:
            return
        bn, lin = self.att_embed[0], self.att_embed[1]
        b = self._bn_buffers(lin.weight.device)
        dW, dbias, dgamma, dbeta = self._bn_targets
        if dW is None:
            return
        mean, var = self._bn_used
        engine.bn_fold_bwd(b['dW'], b['db'], lin.weight.data, bn.weight.data, bn.bias.data, mean, var, bn.eps, dW, dbias,
                           dgamma, dbeta)

    def att_embed_pre(self, att_feats, att_masks=None):
        """relu(att_embed(att_feats)) before the dropout — computed once per training step and shared
        by all decodes of the step (AttModel.py:315; their dropout masks differ).  att_masks: needed by use_bn only."""
        self._check_inputs(att_feats)
        fl = self.flat()
        B, K, _ = att_feats.shape
        dims = self._dims(B, K, self.seq_length)
        att_raw = self._buf.stage('att_raw', att_feats, torch.float32)
        if self.use_bn:
            self._bn_stats(att_raw, self._bn_masks(att_feats, att_masks))
        params = engine.speaker_params(self._speaker_tensors(fl))
        self._staged_att, self._staged_raw = att_feats, att_raw     # the decodes of this step reuse the staged tensor
        att_pre = self._buf.get('att_pre', (B, K, self.rnn_size), torch.float32, att_raw.device)
        return engine.speaker_att_embed_fwd(dims, params, att_raw, att_pre)

    def decode(self, att_feats, att_masks, mode, temp=1.0, **kw):
        """One AttModel.sample / AttModel.forward pass on the device -> DecodeResult."""
        dims, params, io, (mode, att_raw, grad, ws_key) = self._decode_io(att_feats, att_masks, mode, temp, **kw)
        self.noise.flush()
        engine.speaker_decode_launch(dims, params, io)
        return DecodeResult(io, mode, dims, params, att_raw, grad)

    def decode_pair(self, att_feats, att_masks, spec_a, spec_b, att_pre=None):
        """Two decodes of the same images (spec = dict(mode=..., temp=..., **decode kwargs)) advanced in lock step
        through shared launches: every per-timestep kernel runs once over 2B rows.  Same results as two decode()
        calls, bit for bit.  -> (DecodeResult a, DecodeResult b)."""
        ra = self._decode_io(att_feats, att_masks, att_pre=att_pre, **spec_a)
        rb = self._decode_io(att_feats, att_masks, att_pre=att_pre, **spec_b)
        (dims, params, ia, (ma, att_raw, ga, _)), (dims_b, _, ib, (mb, _, gb, _)) = ra, rb
        assert (dims.B, dims.K, dims.T) == (dims_b.B, dims_b.K, dims_b.T), 'paired decodes share their shapes'
        self.noise.flush()                       # the dropout masks of both decodes: one launch
        engine.speaker_decode_fwd_pair(dims, params, ia, ib)
        self.last_pair_fused = engine.speaker_decode_pair_fused(dims, ia, ib)   # reported by bench.py / train.py
        return DecodeResult(ia, ma, dims, params, att_raw, ga), DecodeResult(ib, mb, dims_b, params, att_raw, gb)

    def _decode_io(self, att_feats, att_masks, mode, temp=1.0, att_pre=None, grad=False, T=None, pick=None,
                   first_token=None, decoding_constraint=None, tag='sample', want_stv=None, ss_prob=0.0, ps_prob=0.0):
        """Stages inputs, draws the noise and fills the C-ABI io struct of one decode (no launch of the loop)."""
        self._check_inputs(att_feats)
        fl = self.flat()
        B, K, _ = att_feats.shape
        T = T or self.seq_length
        dims = self._dims(B, K, T)
        if att_pre is not None and getattr(self, '_staged_att', None) is att_feats:
            att_raw = self._staged_raw                                   # staged by att_embed_pre
        else:
            att_raw = self._buf.stage('att_raw', att_feats, torch.float32)
            self._staged_att = None
        if self.use_bn:
            if att_pre is None:
                self._bn_stats(att_raw, self._bn_masks(att_feats, att_masks))
            self._bn_tick()
        params = engine.speaker_params(self._speaker_tensors(fl))
        if att_pre is None:
            att_pre = engine.speaker_att_embed_fwd(dims, params, att_raw,
                                                   self._buf.get('att_pre', (B, K, self.rnn_size), torch.float32, att_raw.device))
        p = self.drop_prob_lm if self.training else 0.0
        dims.p_drop = p
        ss = mode == 'teacher' and ss_prob > 0.0
        ps = mode in ('gumbel_ps', 'multinomial_ps')
        nz = self.noise.decode_noise(tag, B, K, self.rnn_size, self.input_encoding_size, self.vocab_size + 1, T, p,
                                     need_u=(mode in ('gumbel', 'gumbel_ps')) or
                                     (mode in ('multinomial', 'multinomial_st', 'multinomial_ps') and pick is None) or ss,
                                     device=att_raw.device, need_ss=ss, need_ps=ps and ps_prob > 0.0,
                                     u_in_kernel=not ps)
        ss_pick = None
        if mode == 'teacher':
            ss_pick = nz.get('pick') if ss else None     # recorded scheduled-sampling draws (tests)
        elif pick is None:
            pick = nz.get('pick')
        if want_stv is None:
            want_stv = mode in ('gumbel', 'multinomial_st')
        # a decode whose activations must survive until backward() gets its own workspace
        ws_key = (tag, B, K, T, grad)
        dev = att_raw.device
        out = dict(seq=self._buf.get((ws_key, 'seq'), (B, T), torch.int32, dev, fill=0),
                   slp=self._buf.get((ws_key, 'slp'), (B, T), torch.float32, dev, fill=0),
                   stv=self._buf.get((ws_key, 'stv'), (B, T), torch.float32, dev, fill=1) if want_stv else None,
                   L=self._buf.get((ws_key, 'L'), (1,), torch.int32, dev, fill=0))
        if ps:
            V1 = self.vocab_size + 1
            out.update(soft=self._buf.get((ws_key, 'soft'), (T, B, V1), torch.float32, dev),
                       soft_raw=self._buf.get((ws_key, 'soft_raw'), (T, B, V1), torch.float32, dev),
                       xpre=self._buf.get((ws_key, 'xpre'), (T, B, self.input_encoding_size), torch.float32, dev))
        if att_masks is not None:
            att_masks = self._buf.stage('att_masks', att_masks, torch.float32)
        if pick is not None:
            pick = self._buf.stage((tag, 'pick'), pick.contiguous(), torch.int64)
        if first_token is not None:
            first_token = self._buf.stage((tag, 'first'), first_token.contiguous(), torch.int64)
        fwd = engine.speaker_decode_io(dims, params, att_pre, MODES[mode], temp, att_masks=att_masks,
                                       att_keep=nz.get('att_keep'), x_keep=nz.get('x_keep'), out_keep=nz.get('out_keep'),
                                       U=nz.get('gumbel_u'), pick=pick,
                                       decoding_constraint=self.decoding_constraint if decoding_constraint is None else decoding_constraint,
                                       want_stv=want_stv, ws=self._ws.get(ws_key), first_token=first_token, out=out,
                                       ss_u=nz.get('ss_u') if ss else None, ss_prob=ss_prob if ss else 0.0,
                                       ss_pick=ss_pick, ps_u=nz.get('ps_u') if ps else None,
                                       ps_prob=ps_prob if ps else 0.0, u_stream=nz.get('u_stream'), timer=self.timer)
        self._ws[ws_key] = fwd['ws']
        return dims, params, fwd, (mode, att_raw, grad, ws_key)

    def decode_backward(self, res, d_onehot=None, dslp=None, logit_grads_ready=None, dslp_scale=None, before_loop=None):
        """logit_grads_ready: called once the logit layer's gradient is final and the BPTT loop has run (data-parallel runs start
        the all-reduce of the logit bucket there; only when this decode is the last one writing the logit gradient);
        before_loop: called right before the BPTT loop (data-parallel runs let exchanges in flight land there)."""
        fl = self.flat()
        if dslp is not None:
            dslp = self._buf.stage(('dslp_in', res.dims.T), dslp.contiguous(), torch.float32)
        key = ('bwd', res.dims.B, res.dims.K, res.dims.T)
        if dslp_scale is not None:
            # the upstream gradient of the loss as a device scalar: applied inside the sampler backward (partial-sampling
            # decodes run their sampler backward inside the time loop and take dslp as it is)
            dslp_scale = dslp_scale.reshape(1)
            if res.soft is not None or dslp is None or dslp_scale.dtype != torch.float32:
                dslp = dslp * dslp_scale if dslp is not None else None
                dslp_scale = None
        kw = dict(d_onehot=d_onehot, dslp=dslp, dslp_scale=dslp_scale)
        grads = self._speaker_grads(fl)
        if logit_grads_ready is not None and res.soft is None:
            # data-parallel runs: the logit layer's gradient is final after the first phase.  The BPTT loop is ONE launch that
            # fills every CU (a collective cannot run beside it): exchanges already in flight land first (before_loop), the loop
            # runs, and the logit bucket starts behind it, under the batched weight-gradient products of the last phase.
            self._ws[key] = engine.speaker_decode_bwd(res.dims, res.params, res.fwd, grads, res.att_raw,
                                                      ws_bwd=self._ws.get(key), phase=_lib.BWD_LOGIT, **kw)
            if before_loop is None:
                before_loop = getattr(self, 'exchange_barrier', None)
            if before_loop is not None:
                before_loop()
            engine.speaker_decode_bwd(res.dims, res.params, res.fwd, grads, res.att_raw,
                                      ws_bwd=self._ws[key], phase=_lib.BWD_LOOP, **kw)
            logit_grads_ready()
            engine.speaker_decode_bwd(res.dims, res.params, res.fwd, grads, res.att_raw,
                                      ws_bwd=self._ws[key], phase=_lib.BWD_TAIL, **kw)
            self._bn_backward()
            return
        barrier = before_loop if before_loop is not None else getattr(self, 'exchange_barrier', None)
        if barrier is not None:
            barrier()             # (data-parallel runs: see optimizer.overlap_gradient_exchange)
        self._ws[key] = engine.speaker_decode_bwd(res.dims, res.params, res.fwd, grads, res.att_raw,
                                                  ws_bwd=self._ws.get(key), **kw)
        self._bn_backward()
        if logit_grads_ready is not None:
            logit_grads_ready()

    # ---- reference API ---------------------------------------------------------------------
    def forward(self, fc_feats, att_feats, att_masks, seq, masks):
        """Teacher-forced MLE loss, models/AttModel.py:103-148."""
        B, Lp = seq.shape
        T = Lp - 1
        res = self.decode(att_feats, att_masks, 'teacher', 1.0, grad=True, T=T,
                          pick=seq.t().contiguous().long(), first_token=seq[:, 0].contiguous().long(), tag='mle',
                          decoding_constraint=0, want_stv=False, ss_prob=self.ss_prob if self.training else 0.0)
        dslp = torch.empty(B, T, device=att_feats.device)
        loss = engine.masked_nll(res.slp, masks.float()[:, 1:], 1.0, dslp=dslp)
        self._loss['xe'] = loss.detach()[0]

        def bwd(go):
            self.decode_backward(res, dslp=dslp, dslp_scale=go)
        anchor = next((p for p in self.parameters() if p.requires_grad), None)
        if anchor is None or not torch.is_grad_enabled():
            return loss[0].detach().clone()
        return engine_loss(loss[0], anchor, bwd)

    def sample_beam(self, fc_feats, att_feats, att_masks, opt={}):
        """models/AttModel.py:150-289: beam search over all images at once on the device (the reference decodes image
        by image and merges beams on the host).  Evaluation semantics: no dropout.  Returns (seq, seqLogprobs) as
        the reference does and keeps done_beams[k] = [best beam] (seq, logps, p)."""
        self._check_inputs(att_feats)
        beam_size = opt.get('beam_size', 10)
        dc = opt.get('decoding_constraint', self.decoding_constraint)
        assert beam_size <= self.vocab_size + 1, 'lets assume this for now (models/AttModel.py:164-166)'
        fl = self.flat()
        B, K, _ = att_feats.shape
        dims = self._dims(B, K, self.seq_length)
        dims.p_drop = 0.0
        att_raw = self._buf.stage('att_raw', att_feats, torch.float32)
        self._staged_att = None
        if self.use_bn:
            self._bn_stats(att_raw, self._bn_masks(att_feats, att_masks))
            self._bn_tick()
        params = engine.speaker_params(self._speaker_tensors(fl))
        att_pre = engine.speaker_att_embed_fwd(dims, params, att_raw,
                                               self._buf.get('att_pre', (B, K, self.rnn_size), torch.float32, att_raw.device))
        if att_masks is not None:
            att_masks = self._buf.stage('att_masks', att_masks, torch.float32)
        key = ('beam', B, K, beam_size)
        out = engine.speaker_beam_search(dims, params, att_pre, beam_size, att_masks, dc, ws=self._ws.get(key))
        self._ws[key] = out['ws']
        seq, logps = out['seq'].long(), out['logps']
        self.done_beams = [[{'seq': seq[k], 'logps': logps[k], 'p': out['score'][k]}] for k in range(B)]
        return seq, logps

    def sample(self, fc_feats, att_feats, att_masks, opt={}):
        """models/AttModel.py:291-452 (beam_size 1).  Outputs are detached (evaluation use); the joint
        model differentiates through decode()/decode_backward() directly."""
        use_one_hot = opt.get('use_one_hot', 0)
        sample_max = opt.get('sample_max', 1)
        beam_size = opt.get('beam_size', 1)
        temperature = opt.get('temperature', 1.0)
        if beam_size > 1:
            return self.sample_beam(fc_feats, att_feats, att_masks, opt)
        dc = opt.get('decoding_constraint', self.decoding_constraint)
        plain = self.retrieval_reward == 'reinforce' or not use_one_hot
        if sample_max:
            res = self.decode(att_feats, att_masks, 'greedy', decoding_constraint=dc, tag='greedy')
        elif plain:
            res = self.decode(att_feats, att_masks, 'multinomial', temperature, decoding_constraint=dc)
        elif self.retrieval_reward == 'gumbel':
            res = self.decode(att_feats, att_masks, 'gumbel', self.gumbel_temp, decoding_constraint=dc)
        elif self.retrieval_reward == 'multinomial':
            res = self.decode(att_feats, att_masks, 'multinomial_st', self.multinomial_temp, decoding_constraint=dc)
        elif self.retrieval_reward == 'gumbel_softmax':
            res = self.decode(att_feats, att_masks, 'gumbel_ps', self.gumbel_temp, decoding_constraint=dc,
                              ps_prob=self.prob_gumbel_softmax)
        elif self.retrieval_reward == 'multinomial_soft':
            res = self.decode(att_feats, att_masks, 'multinomial_ps', self.multinomial_temp, decoding_constraint=dc,
                              ps_prob=self.prob_multinomial_soft)
        else:
            raise ValueError(f'unknown retrieval_reward {self.retrieval_reward!r}')
        L = int(res.L.item())                      # the one host sync of a decode (the reference syncs every step)
        if L == 0:
            raise ValueError('every caption ended at the first step (the reference raises here too: '
                             'torch.cat of an empty list, AttModel.py:446)')
        seq = res.seq[:, :L].long()
        slp = res.slp[:, :L].clone()
        stv = res.stv[:, :L].clone() if res.stv is not None else None
        if sample_max or plain:
            return seq, slp
        if res.soft is not None:          # soft rows, with the reference's trailing zero <bos> column (:372-378)
            soft = torch.zeros(seq.shape[0], L, self.vocab_size + 2, device=seq.device)
            soft[:, :, :self.vocab_size + 1] = res.soft[:L].transpose(0, 1)
            return seq, soft, slp
        one_hot = torch.zeros(seq.shape[0], L, self.vocab_size + 2, device=seq.device)
        one_hot.scatter_(2, seq.unsqueeze(2), stv.unsqueeze(2))
        return seq, one_hot, slp


class Att2in2Model(AttModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.core = Att2in2Core(opt)
        self.fc_embed = lambda x: x
