"""FCModel: the fc-feature speaker (models/FCModel.py:12-327), the caption model of the reference's
CPU plumbing configuration (BASELINE configs[0]: MLE, batch 2, fc_feats only).  Same constructor
signature, attributes and state-dict names; the computation runs through the speaker decode engine of
libcic_hip.so in its ``fc_mode`` (image step first, plain token embeddings, no attention, dropped-out
recurrent state).  The nn.Modules are parameter containers only."""
import torch
import torch.nn as nn

from .. import _lib, engine, ops
from ..flat import FlatAgent
from ..noise import NoiseSource
from ..bufcache import BufCache
from ..autograd_glue import engine_loss
from .AttModel import DecodeResult


class LSTMCore(nn.Module):
    """Parameter container of models/FCModel.py:12-22."""

    def __init__(self, opt):
        super().__init__()
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.drop_prob_lm = opt.drop_prob_lm
        self.i2h = nn.Linear(self.input_encoding_size, 5 * self.rnn_size)
        self.h2h = nn.Linear(self.rnn_size, 5 * self.rnn_size)
        self.dropout = nn.Dropout(self.drop_prob_lm)


_FIELDS = {'embed_w': 'embed.weight', 'logit_w': 'logit.weight', 'logit_b': 'logit.bias',
           'i2h_w': 'core.i2h.weight', 'i2h_b': 'core.i2h.bias', 'h2h_w': 'core.h2h.weight', 'h2h_b': 'core.h2h.bias'}


def _params(tensors):
    sp = _lib.SpeakerParams()
    for field, key in _FIELDS.items():
        setattr(sp, field, tensors[key].data_ptr())
    return sp


class FCModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.vocab_size = opt.vocab_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_type = opt.rnn_type
        self.rnn_size = opt.rnn_size
        self.num_layers = opt.num_layers
        self.drop_prob_lm = opt.drop_prob_lm
        self.seq_length = opt.seq_length
        self.fc_feat_size = opt.fc_feat_size
        if self.num_layers != 1 or self.rnn_type != 'lstm':
            raise NotImplementedError('FCModel runs its single-layer maxout LSTMCore (models/FCModel.py:12-43)')
        self.ss_prob = 0.0
        # same construction order as the reference: the same seed draws the same weights (:59-62)
        self.img_embed = nn.Linear(self.fc_feat_size, self.input_encoding_size)
        self.core = LSTMCore(opt)
        self.embed = nn.Embedding(self.vocab_size + 2, self.input_encoding_size)
        self.logit = nn.Linear(self.rnn_size, self.vocab_size + 1)
        self.init_weights()
        self.decoding_constraint = getattr(opt, 'decoding_constraint', 0)
        self._loss = {}
        self.gumbel_temp = opt.gumbel_temp
        self._flat = None
        self.noise = NoiseSource()
        self._ws = {}
        self._buf = BufCache()
        self._step_fc = None                           # the fc features of the joint step in flight (AlternatingJointModel)

    def init_weights(self):                                            # :74-78
        initrange = 0.1
        self.embed.weight.data.uniform_(-initrange, initrange)
        self.logit.bias.data.fill_(0)
        self.logit.weight.data.uniform_(-initrange, initrange)

    def flat(self):
        if self._flat is None:
            self._flat = FlatAgent(self)
        self._flat.ensure()
        return self._flat

    def _dims(self, B, T):
        p = self.drop_prob_lm if self.training else 0.0
        return engine.speaker_dims(B, 0, 1, self.rnn_size, self.input_encoding_size, 0, self.vocab_size, T, p)

    def _decode(self, fc_feats, mode, temp=1.0, T=None, pick=None, first_token=None, grad=False, tag='sample',
                decoding_constraint=0):
        if not fc_feats.is_cuda:
            raise _lib.CicError('cooperativeimagecaptioning_amd runs on the GPU only: fc_feats is on ' +
                                str(fc_feats.device) + ' (there is no CPU fallback path)')
        fl = self.flat()
        tens = fl.tensors()
        B = fc_feats.shape[0]
        T = T or self.seq_length
        dims = self._dims(B, T)
        params = _params(tens)
        dev = fc_feats.device
        fc = self._buf.stage('fc', fc_feats, torch.float32)
        # xt of the image step: img_embed(fc_feats)                       (:99, :276)
        x0 = self._buf.get((tag, 'x0'), (B, self.input_encoding_size), torch.float32, dev)
        ops.gemm(fc, tens['img_embed.weight'], x0, True, True, bias=tens['img_embed.bias'])
        p = dims.p_drop
        keep = None
        ov = self.noise.override.get(tag) if self.noise.override is not None else None
        if ov is not None:
            if p > 0.0 and ov.get('out_keep') is not None:
                keep = torch.as_tensor(ov['out_keep']).to(torch.uint8).to(dev).contiguous()
            if pick is None and ov.get('pick') is not None:
                # the reference draws at loop iteration t = step + 1 (FCModel.py:274-300): drop the image row
                pick = torch.as_tensor(ov['pick']).long()[1:].to(dev).contiguous()
        elif p > 0.0:
            keep = self.noise._get((tag, 'out_keep'), (T + 2, B, self.rnn_size), torch.uint8, dev)
            ops.dropout_keep_(keep, p, self.noise.seed, self.noise._next_offset())
        u_stream = None
        if mode == 'multinomial' and pick is None:      # Gumbel-max draws from a Philox stream, inside the kernels
            u_stream = (self.noise.seed, self.noise._next_offset())
        ws_key = (tag, B, T, grad)
        out = dict(seq=self._buf.get((ws_key, 'seq'), (B, T), torch.int32, dev, fill=0),
                   slp=self._buf.get((ws_key, 'slp'), (B, T), torch.float32, dev, fill=0), stv=None,
                   L=self._buf.get((ws_key, 'L'), (1,), torch.int32, dev, fill=0))
        from .AttModel import MODES
        io = engine.speaker_decode_io(dims, params, None, MODES[mode], temp, out_keep=keep, u_stream=u_stream, pick=pick,
                                      decoding_constraint=decoding_constraint, ws=self._ws.get(ws_key),
                                      first_token=first_token, out=out, fc_x0=x0)
        self._ws[ws_key] = io['ws']
        engine.speaker_decode_launch(dims, params, io)
        res = DecodeResult(io, mode, dims, params, None, grad)
        res.x0, res.fc = x0, fc
        return res

    def _decode_backward(self, res, dslp):
        fl = self.flat()
        grads = fl.grad_tensors()
        B, E = res.x0.shape
        d_x0 = self._buf.get('d_x0', (B, E), torch.float32, res.x0.device)
        key = ('bwd', res.dims.B, res.dims.T)
        self._ws[key] = engine.speaker_decode_bwd(res.dims, res.params, res.fwd, grads, None, dslp=dslp,
                                                  ws_bwd=self._ws.get(key), grad_params=_params(grads), d_x0=d_x0)
        # img_embed backward: dW += d_x0^T fc, db += colsum(d_x0)         (:99)
        ops.gemm(d_x0, res.fc, grads['img_embed.weight'], False, False, accumulate=True, sum_order_free=True)
        ops.colsum(d_x0, grads['img_embed.bias'], accumulate=True)

    # ---- the decode interface AlternatingJointModel drives (same names as AttModel's) ---------------------------
    def att_embed_pre(self, att_feats, att_masks=None):
        return None                                    # no region features in this speaker

    def decode(self, att_feats, att_masks, mode, temp=1.0, att_pre=None, grad=False, T=None, pick=None, first_token=None,
               tag='sample', decoding_constraint=None, want_stv=None, ss_prob=0.0, ps_prob=0.0, fc_feats=None):
        """One FCModel.sample / FCModel.forward pass -> DecodeResult.  FCModel.sample returns (seq, logprobs) only
        (FCModel.py:324-325), so the reference can run it under the MLE, REINFORCE and CIDEr terms: greedy, multinomial
        and teacher-forced decodes; the straight-through modes have no FCModel form."""
        if mode not in ('greedy', 'multinomial', 'teacher'):
            raise NotImplementedError(f"caption_model 'fc' has no {mode!r} decode: FCModel.sample returns (seq, logprobs) "
                                      'only, the straight-through retrieval rewards need the one-hot output of AttModel')
        if ss_prob:
            raise NotImplementedError('scheduled sampling is supported for att2in2 only')
        fc_feats = self._step_fc if fc_feats is None else fc_feats
        dc = self.decoding_constraint if decoding_constraint is None else decoding_constraint
        return self._decode(fc_feats, mode, temp, T=T, pick=pick, first_token=first_token, grad=grad, tag=tag,
                            decoding_constraint=dc)

    def decode_backward(self, res, d_onehot=None, dslp=None, logit_grads_ready=None, dslp_scale=None):
        assert d_onehot is None and dslp is not None, 'the fc speaker receives gradient through its log-probabilities only'
        if dslp_scale is not None:
            dslp = dslp * dslp_scale
        self._decode_backward(res, dslp=dslp.contiguous())
        if logit_grads_ready is not None:
            logit_grads_ready()

    # ---- reference API ---------------------------------------------------------------------
    def forward(self, fc_feats, att_feats, att_masks, seq, masks):
        """Teacher-forced MLE loss, models/FCModel.py:91-131."""
        B, Lp = seq.shape
        T = Lp - 1
        if self.training and self.ss_prob > 0.0:
            raise NotImplementedError('scheduled sampling is supported for att2in2 only')
        res = self._decode(fc_feats, 'teacher', 1.0, T=T, pick=seq.t().contiguous().long(),
                           first_token=seq[:, 0].contiguous().long(), grad=True, tag='mle')
        dslp = torch.empty(B, T, device=fc_feats.device)
        loss = engine.masked_nll(res.slp, masks.float()[:, 1:], 1.0, dslp=dslp)
        self._loss['xe'] = loss.detach()[0]

        def bwd(go):
            self._decode_backward(res, dslp=(dslp * go).contiguous())
        anchor = next((p for p in self.parameters() if p.requires_grad), None)
        if anchor is None or not torch.is_grad_enabled():
            return loss[0].detach().clone()
        return engine_loss(loss[0], anchor, bwd)

    def sample(self, fc_feats, att_feats, att_masks, opt={}):
        """models/FCModel.py:260-327 (beam_size 1; sample_max 1 greedy, 0 multinomial)."""
        sample_max = opt.get('sample_max', 1)
        beam_size = opt.get('beam_size', 1)
        temperature = opt.get('temperature', 1.0)
        dc = opt.get('decoding_constraint', self.decoding_constraint)
        if beam_size > 1:
            raise NotImplementedError('FCModel.sample_beam references undefined names in the reference '
                                      '(FCModel.py:161-163) and is never run by its scripts')
        if sample_max == 1:
            res = self._decode(fc_feats, 'greedy', decoding_constraint=dc, tag='greedy')
        elif sample_max == 0:
            res = self._decode(fc_feats, 'multinomial', temperature, decoding_constraint=dc)
        else:
            raise NotImplementedError('sample_max == 2 (in-place Gumbel argmax, FCModel.py:284-289)')
        L = int(res.L.item())
        if L == 0:
            raise ValueError('every caption ended at the first step (the reference raises here too)')
        return res.seq[:, :L].long(), res.slp[:, :L].clone()
