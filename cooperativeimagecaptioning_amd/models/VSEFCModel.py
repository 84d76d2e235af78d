"""Listener: VSEFCModel with the reference's constructor signature, attributes and state-dict
names (models/VSEFCModel.py:19-241), computing through the HIP listener engine."""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib, engine
from ..flat import FlatAgent
from ..autograd_glue import engine_loss
from ..bufcache import BufCache


class EncoderImage(nn.Module):
    """Parameter container of models/VSEFCModel.py:19-38 (Xavier-uniform fc, zero bias)."""

    def __init__(self, opt):
        super().__init__()
        self.embed_size = opt.vse_embed_size
        self.no_imgnorm = opt.vse_no_imgnorm
        self.use_abs = opt.vse_use_abs
        self.fc_feat_size = opt.fc_feat_size
        self.fc = nn.Linear(self.fc_feat_size, self.embed_size)
        self.init_weights()

    def init_weights(self):
        r = np.sqrt(6.) / np.sqrt(self.fc.in_features + self.fc.out_features)
        self.fc.weight.data.uniform_(-r, r)
        self.fc.bias.data.fill_(0)


class EncoderText(nn.Module):
    """Parameter container of models/VSEFCModel.py:57-81."""

    def __init__(self, opt):
        super().__init__()
        self.use_abs = opt.vse_use_abs
        self.input_encoding_size = opt.input_encoding_size
        self.embed_size = opt.vse_embed_size
        self.num_layers = opt.vse_num_layers
        self.rnn_type = opt.vse_rnn_type
        self.vocab_size = opt.vocab_size
        self.pool_type = getattr(opt, 'vse_pool_type', '')
        if self.rnn_type.lower() != 'gru' or self.num_layers != 1:
            raise NotImplementedError('the MI355X listener path is the single-layer GRU the scripts use')
        self.embed = nn.Embedding(self.vocab_size + 2, self.input_encoding_size)
        self.rnn = nn.GRU(self.input_encoding_size, self.embed_size, self.num_layers, batch_first=True)
        self.init_weights()

    def init_weights(self):
        self.embed.weight.data.uniform_(-0.1, 0.1)


class ContrastiveLoss(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.margin = opt.vse_margin
        self.measure = opt.vse_measure
        if self.measure != 'cosine':
            raise NotImplementedError('only the cosine similarity is supported (as in the reference)')
        self.max_violation = opt.vse_max_violation


class ListenerResult:
    def __init__(self, fwd, dims, params):
        self.fwd, self.dims, self.params = fwd, dims, params
        self.loss_rows, self.loss_sum = fwd['loss_rows'], fwd['loss_sum']


class VSEFCModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.loss_type = opt.vse_loss_type
        self.img_enc = EncoderImage(opt)
        self.txt_enc = EncoderText(opt)
        self.contrastive_loss = ContrastiveLoss(opt)
        self.margin = opt.vse_margin
        self.embed_size = opt.vse_embed_size
        self.vocab_size = opt.vocab_size
        self.seq_length = opt.seq_length
        # 'bf16': the reduced-precision variant (cic.h, cic_listener_dims.compute_dtype); 'f32': the reference's arithmetic
        self.compute_dtype = getattr(opt, 'compute_dtype', 'f32') or 'f32'
        self._loss = {}
        self._flat = None
        self._ws = {}
        self._buf = BufCache()
        self._n = 0

    def flat(self):
        if self._flat is None:
            self._flat = FlatAgent(self)
        self._flat.ensure()
        return self._flat

    def _dims(self, B, Lp):
        e = self.txt_enc
        return engine.listener_dims(B, self.img_enc.fc_feat_size, e.input_encoding_size, self.embed_size,
                                    self.vocab_size, self.seq_length, Lp, self.margin,
                                    self.contrastive_loss.max_violation, self.img_enc.no_imgnorm, self.img_enc.use_abs,
                                    pool=e.pool_type, compute_dtype=self.compute_dtype)

    def run(self, fc_feats, labels=None, masks=None, decode=None, only_one_retrieval='off', slot=0, want_emb=False):
        """Forward on the device.  Captions come either from ground-truth ``labels``/``masks`` or from a
        speaker DecodeResult (generated captions, straight-through values included).  ``slot`` keeps
        the workspaces of several listener passes of one step apart."""
        if not fc_feats.is_cuda:
            raise _lib.CicError('cooperativeimagecaptioning_amd runs on the GPU only (no CPU fallback path)')
        fl = self.flat()
        B = fc_feats.shape[0]
        params = engine.listener_params(fl.tensors())
        fc = self._buf.stage('fc', fc_feats, torch.float32)
        dev = fc.device

        def outs(key, J):
            return dict(loss_rows=self._buf.get((key, 'rows'), (B,), torch.float32, dev),
                        loss_sum=self._buf.get((key, 'sum'), (1,), torch.float32, dev),
                        img_emb=self._buf.get((key, 'img'), (B, J), torch.float32, dev) if want_emb else None,
                        cap_emb=self._buf.get((key, 'cap'), (B, J), torch.float32, dev) if want_emb else None)
        if decode is not None:
            dims = self._dims(B, self.seq_length + 1)
            key = ('gen', B, slot)
            fwd = engine.listener_fwd(dims, params, fc, seq=decode.seq, stv=decode.stv, L=decode.L,
                                      only_one_retrieval=only_one_retrieval, want_emb=want_emb, ws=self._ws.get(key),
                                      out=outs(key, dims.J), soft=getattr(decode, 'soft', None))
        else:
            if labels.dim() > 2:
                raise NotImplementedError('dense one-hot / soft caption input (VSEFCModel.py:102-104) is only '
                                          'supported through the joint model (straight-through token + value)')
            dims = self._dims(B, labels.shape[1])
            key = ('lab', B, labels.shape[1], slot)
            fwd = engine.listener_fwd(dims, params, fc, labels=self._buf.stage((key, 'labels'), labels, torch.int64),
                                      masks=self._buf.stage((key, 'masks'), masks, torch.float32),
                                      only_one_retrieval=only_one_retrieval, want_emb=want_emb, ws=self._ws.get(key),
                                      out=outs(key, dims.J))
        self._ws[key] = fwd['ws']
        return ListenerResult(fwd, dims, params)

    def run_backward(self, res, g_scalar=None, g_rows=None, param_grads=True, d_onehot=None, g_scale=1.0):
        """g_scale: a host factor on the upstream gradient (a loss weight), applied inside the first backward kernel."""
        fl = self.flat()
        if g_scalar is not None:
            g_scalar = self._buf.stage('g_scalar', g_scalar.reshape(1), torch.float32)
        if g_rows is not None:
            g_rows = self._buf.stage('g_rows', g_rows.contiguous(), torch.float32)
        grads = None
        if param_grads:
            # a frozen parameter takes no gradient (share_embed = 1: the shared table in a reinforce listener turn,
            # AlternatingJointModel.py:592-645 - the engine skips a missing pointer)
            grads = fl.grad_tensors()
            if not self.txt_enc.embed.weight.requires_grad:
                grads = {k: v for k, v in grads.items() if k != 'txt_enc.embed.weight'}
        engine.listener_bwd(res.dims, res.params, res.fwd, g_rows=g_rows, g_scalar=g_scalar,
                            grads=grads, d_onehot=d_onehot, g_scale=g_scale)

    def forward(self, fc_feats, att_feats, seq, masks, whole_batch=False, only_one_retrieval='off'):
        """models/VSEFCModel.py:230-241."""
        res = self.run(fc_feats, labels=seq, masks=masks, only_one_retrieval=only_one_retrieval, slot=self._next_slot())
        if whole_batch:
            value = res.loss_rows

            def bwd(go):
                self.run_backward(res, g_rows=go.contiguous())
        else:
            value = res.loss_sum[0]
            self._loss['contrastive'] = value.detach()

            def bwd(go):
                self.run_backward(res, g_scalar=go.reshape(1).contiguous())
        anchor = next((p for p in self.parameters() if p.requires_grad), None)
        if anchor is None or not torch.is_grad_enabled():
            return value.detach().clone()
        return engine_loss(value, anchor, bwd)

    def _next_slot(self):
        self._n = (self._n + 1) % 4
        return 100 + self._n
