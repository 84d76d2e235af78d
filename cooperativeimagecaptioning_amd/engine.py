"""Sequence-level entry points of the C ABI (speaker decode, listener, rewards) wrapped for
torch tensors.  All state lives in caller-visible tensors; the workspace tensor returned by a
forward call is what the matching backward call consumes."""
import ctypes as C

import torch

from . import _lib, status
from ._lib import (lib, check, stream, SpeakerDims, SpeakerParams, DecodeIO, DecodeBwdIO, SPEAKER_PARAM_FIELDS,
                   ListenerDims, ListenerParams, ListenerIO, ListenerBwdIO, LISTENER_PARAM_FIELDS, CiderdArgs)

P = C.c_void_p

lib.cic_speaker_decode_ws_bytes.argtypes = [C.POINTER(SpeakerDims)]
lib.cic_speaker_decode_ws_bytes.restype = C.c_size_t
lib.cic_speaker_att_embed_fwd.argtypes = [C.POINTER(SpeakerDims), C.POINTER(SpeakerParams), P, P, P]
lib.cic_speaker_att_embed_fwd.restype = C.c_int
lib.cic_speaker_decode_fwd.argtypes = [C.POINTER(SpeakerDims), C.POINTER(SpeakerParams), C.POINTER(DecodeIO), P,
                                       C.c_size_t, P]
lib.cic_speaker_decode_fwd.restype = C.c_int


lib.cic_speaker_decode_bwd_ws_bytes.argtypes = [C.POINTER(SpeakerDims)]
lib.cic_speaker_decode_bwd_ws_bytes.restype = C.c_size_t
lib.cic_speaker_decode_bwd.argtypes = [C.POINTER(SpeakerDims), C.POINTER(SpeakerParams), C.POINTER(DecodeIO),
                                       C.POINTER(DecodeBwdIO), P, C.c_size_t, P, C.c_size_t, P]
lib.cic_speaker_decode_bwd.restype = C.c_int
lib.cic_listener_ws_bytes.argtypes = [C.POINTER(ListenerDims)]
lib.cic_listener_ws_bytes.restype = C.c_size_t
lib.cic_listener_fwd.argtypes = [C.POINTER(ListenerDims), C.POINTER(ListenerParams), C.POINTER(ListenerIO), P,
                                 C.c_size_t, P]
lib.cic_listener_fwd.restype = C.c_int
lib.cic_listener_bwd.argtypes = [C.POINTER(ListenerDims), C.POINTER(ListenerParams), C.POINTER(ListenerIO),
                                 C.POINTER(ListenerBwdIO), P, C.c_size_t, P]
lib.cic_listener_bwd.restype = C.c_int

lib.cic_ciderd_ws_bytes.argtypes = [C.c_int, C.c_int]
lib.cic_ciderd_ws_bytes.restype = C.c_size_t
lib.cic_ciderd_reward.argtypes = [C.POINTER(CiderdArgs), P, C.c_size_t, P]
lib.cic_ciderd_reward.restype = C.c_int
lib.cic_seq_loss.argtypes = [P, P, P, P, C.c_float, C.c_float, C.c_int, C.c_int, P, P, C.c_int, P]
lib.cic_seq_loss.restype = C.c_int
lib.cic_seq_loss_total.argtypes = [P, P, P, P, C.c_float, C.c_float, C.c_int, C.c_int, P, P, C.c_int,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.c_int, C.c_float, P, P]
lib.cic_seq_loss_total.restype = C.c_int
lib.cic_loss_combine.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.c_int, P, P]
lib.cic_loss_combine.restype = C.c_int
lib.cic_masked_nll.argtypes = [P, P, C.c_int, C.c_float, C.c_int, C.c_int, P, P, P]
lib.cic_masked_nll.restype = C.c_int
lib.cic_clamp_adam.argtypes = [P, P, P, P, C.c_int64] + [C.c_double] * 6 + [C.c_int, C.c_double, P]
lib.cic_clamp_adam.restype = C.c_int
lib.cic_clamp_adam_zero.argtypes = [P, P, P, P, C.c_int64] + [C.c_double] * 6 + [C.c_int, C.c_double, C.c_int, P]
lib.cic_clamp_adam_zero.restype = C.c_int
lib.cic_clamp_adam_guarded.argtypes = [P, P, P, P, C.c_int64] + [C.c_double] * 6 + [C.c_int, C.c_double, C.c_int, P, P]
lib.cic_clamp_adam_guarded.restype = C.c_int

ONLY_ONE = {'off': 0, 'image': 1, 'caption': 2}
# several processes computing on ONE GPU (the multi-rank rehearsals on a one-GPU box: bench.py / train.py / tools with
# --same-device; CIC_SHARED_DEVICE=1): launches that need all their workgroups resident together are not used then
DEVICE_SHARED = [__import__('os').environ.get('CIC_SHARED_DEVICE', '0') == '1']


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'cic engine: CUDA-contiguous tensors only'
    return t.data_ptr()


def speaker_dims(B, K, D, H, E, A, V, T, p_drop, compute_dtype='f32'):
    d = SpeakerDims()
    d.B, d.K, d.D, d.H, d.E, d.A, d.V, d.T, d.p_drop = B, K, D, H, E, A, V, T, float(p_drop)
    d.compute_dtype = {'f32': 0, 'bf16': 1}[compute_dtype]
    return d


def speaker_params(tensors):
    """tensors: dict keyed by the reference state-dict names -> struct of device pointers."""
    sp = SpeakerParams()
    for field, key in SPEAKER_PARAM_FIELDS:
        t = tensors[key]
        assert t is None or t.dtype == torch.float32
        setattr(sp, field, _p(t))
    return sp


def speaker_att_embed_fwd(dims, params, att_raw, att_pre=None):
    if att_pre is None:
        att_pre = torch.empty(dims.B, dims.K, dims.H, device=att_raw.device)
    check(lib.cic_speaker_att_embed_fwd(C.byref(dims), C.byref(params), _p(att_raw), _p(att_pre), stream()),
          'cic_speaker_att_embed_fwd')
    return att_pre


lib.cic_speaker_decode_fwd_pair.argtypes = [C.POINTER(SpeakerDims), C.POINTER(SpeakerParams), C.POINTER(DecodeIO), P,
                                            C.c_size_t, C.POINTER(DecodeIO), P, C.c_size_t, P]
lib.cic_speaker_decode_fwd_pair.restype = C.c_int
lib.cic_speaker_decode_pair_fused.argtypes = [C.POINTER(SpeakerDims), C.POINTER(DecodeIO), C.POINTER(DecodeIO)]
lib.cic_speaker_decode_pair_fused.restype = C.c_int


def speaker_decode_fwd(dims, params, att_pre, mode, temp=1.0, *args, **kw):
    return speaker_decode_launch(dims, params, speaker_decode_io(dims, params, att_pre, mode, temp, *args, **kw))


def speaker_decode_launch(dims, params, out):
    ws = out['ws']
    check(lib.cic_speaker_decode_fwd(C.byref(dims), C.byref(params), C.byref(out['io']), ws.data_ptr(), ws.numel(),
                                     stream()), 'cic_speaker_decode_fwd')
    return out


def speaker_decode_fwd_pair(dims, params, a, b):
    """a, b: dicts from speaker_decode_io (two decodes of the same images and weights).  One launch per
    per-timestep kernel over the rows of both; results identical to two speaker_decode_fwd calls."""
    check(lib.cic_speaker_decode_fwd_pair(C.byref(dims), C.byref(params), C.byref(a['io']), a['ws'].data_ptr(),
                                          a['ws'].numel(), C.byref(b['io']), b['ws'].data_ptr(), b['ws'].numel(),
                                          stream()), 'cic_speaker_decode_fwd_pair')
    return a, b


def speaker_decode_pair_fused(dims, a, b):
    """True when the pair goes through shared launches (else the library runs the two decodes one after the other)."""
    return bool(lib.cic_speaker_decode_pair_fused(C.byref(dims), C.byref(a['io']), C.byref(b['io'])))


def speaker_decode_io(dims, params, att_pre, mode, temp=1.0, att_masks=None, att_keep=None, x_keep=None,
                      out_keep=None, U=None, pick=None, decoding_constraint=0, want_stv=False, ws=None,
                      first_token=None, out=None, ss_u=None, ss_prob=0.0, ss_pick=None, ps_u=None, ps_prob=0.0,
                      fc_x0=None, u_stream=None, timer=None):
    """-> dict(seq i32[B,T], slp f32[B,T], stv f32[B,T]|None, L i32[1], ws); partial-sampling modes add
    soft f32[T,B,V+1] (the caption rows handed to the listener) and the saved soft_raw / xpre."""
    dev = att_pre.device if att_pre is not None else fc_x0.device
    B, T = dims.B, dims.T
    nbytes = lib.cic_speaker_decode_ws_bytes(C.byref(dims))
    if ws is None or ws.numel() < nbytes:
        # zero-filled ONCE: a decode that stops early (every caption ended) leaves the slabs of its remaining steps as
        # they were, and the backward pass multiplies them by zero gradients - they have to be finite
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    if out is None:
        out = dict(seq=torch.zeros(B, T, dtype=torch.int32, device=dev), slp=torch.zeros(B, T, device=dev),
                   stv=torch.ones(B, T, device=dev) if want_stv else None,
                   L=torch.zeros(1, dtype=torch.int32, device=dev))
    out['ws'] = ws
    io = DecodeIO()
    if mode in (_lib.SAMPLE_GUMBEL_PS, _lib.SAMPLE_MULTINOMIAL_PS):
        V1 = dims.V + 1
        for k, shape in (('soft', (T, B, V1)), ('soft_raw', (T, B, V1)), ('xpre', (T, B, dims.E))):
            if out.get(k) is None:
                out[k] = torch.empty(shape, device=dev)
        io.soft_out, io.soft_raw, io.xpre = _p(out['soft']), _p(out['soft_raw']), _p(out['xpre'])
        io.ps_u, io.ps_prob = _p(ps_u), float(ps_prob)
    io.mode, io.temp, io.decoding_constraint = mode, float(temp), int(decoding_constraint)
    io.att_pre, io.att_masks = _p(att_pre), _p(att_masks)
    if fc_x0 is not None:          # FCModel decode (dims.K == 0)
        io.fc_mode, io.x0 = 1, _p(fc_x0)
    io.att_keep, io.x_keep, io.out_keep = _p(att_keep), _p(x_keep), _p(out_keep)
    io.U, io.pick, io.first_token = _p(U), _p(pick), _p(first_token)
    if u_stream is not None:      # (seed, offset): the kernels draw the Gumbel uniforms themselves (no [T+1,B,V+1] slab)
        assert U is None
        io.u_philox, io.u_seed, io.u_offset = 1, int(u_stream[0]), int(u_stream[1])
    if timer is not None:
        io.timer = timer.handle
    io.ss_u, io.ss_prob, io.ss_pick = _p(ss_u), float(ss_prob), _p(ss_pick)
    io.device_shared = 1 if DEVICE_SHARED[0] else 0
    io.status = status.ptr(dev)           # hand-off time-outs of the one-launch loops land in the sticky status word
    io.seq, io.slp, io.stv, io.L = _p(out['seq']), _p(out['slp']), _p(out['stv']), _p(out['L'])
    out['io'] = io
    out['_keep'] = (att_pre, att_masks, att_keep, x_keep, out_keep, U, pick, first_token, ss_u, ss_pick, ps_u, fc_x0)   # alive until the backward call
    return out


def speaker_decode_bwd(dims, params, fwd, grads, att_raw, d_onehot=None, dslp=None, ws_bwd=None, grad_params=None,
                       d_x0=None, phase=0, dslp_scale=None):
    """Accumulates parameter gradients of one decode into `grads` (dict of tensors keyed like
    the parameters).  fwd: the dict returned by speaker_decode_fwd."""
    nbytes = lib.cic_speaker_decode_bwd_ws_bytes(C.byref(dims))
    if ws_bwd is None or ws_bwd.numel() < nbytes:
        ws_bwd = torch.zeros(nbytes, dtype=torch.uint8, device=fwd['ws'].device)
    bio = DecodeBwdIO()
    gp = grad_params if grad_params is not None else speaker_params(grads)
    bio.d_onehot, bio.dslp, bio.att_raw, bio.d_x0 = _p(d_onehot), _p(dslp), _p(att_raw), _p(d_x0)
    bio.grads = C.pointer(gp)
    bio.phase = int(phase)
    bio.device_shared = 1 if DEVICE_SHARED[0] else 0
    bio.dslp_scale = _p(dslp_scale)      # f32[1] on the device: dslp is multiplied by it inside the sampler backward
    ws = fwd['ws']
    check(lib.cic_speaker_decode_bwd(C.byref(dims), C.byref(params), C.byref(fwd['io']), C.byref(bio),
                                     ws.data_ptr(), ws.numel(), ws_bwd.data_ptr(), ws_bwd.numel(), stream()),
          'cic_speaker_decode_bwd')
    return ws_bwd


lib.cic_speaker_beam_ws_bytes.argtypes = [C.POINTER(SpeakerDims), C.c_int]
lib.cic_speaker_beam_ws_bytes.restype = C.c_size_t
lib.cic_speaker_beam_search.argtypes = [C.POINTER(SpeakerDims), C.POINTER(SpeakerParams), C.POINTER(_lib.BeamIO), P, C.c_size_t, P]
lib.cic_speaker_beam_search.restype = C.c_int


def speaker_beam_search(dims, params, att_pre, beam, att_masks=None, decoding_constraint=0, ws=None):
    """AttModel.sample_beam on the device -> dict(seq i32[B,T], logps f32[B,T], score f32[B], ws)."""
    dev = att_pre.device
    nbytes = lib.cic_speaker_beam_ws_bytes(C.byref(dims), int(beam))
    if nbytes == 0:
        raise _lib.CicError(f'beam_size {beam} is outside 1..16')
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = dict(seq=torch.zeros(dims.B, dims.T, dtype=torch.int32, device=dev), logps=torch.zeros(dims.B, dims.T, device=dev),
               score=torch.zeros(dims.B, device=dev), ws=ws)
    io = _lib.BeamIO()
    io.beam, io.decoding_constraint = int(beam), int(decoding_constraint)
    io.att_pre, io.att_masks = _p(att_pre), _p(att_masks)
    io.seq, io.logps, io.score = _p(out['seq']), _p(out['logps']), _p(out['score'])
    check(lib.cic_speaker_beam_search(C.byref(dims), C.byref(params), C.byref(io), ws.data_ptr(), ws.numel(), stream()),
          'cic_speaker_beam_search')
    return out


def listener_dims(B, F, E, J, V, T, Lp, margin=0.2, max_violation=1, no_imgnorm=0, use_abs=0, pool='last', compute_dtype='f32'):
    d = ListenerDims()
    d.B, d.F, d.E, d.J, d.V, d.T, d.Lp = B, F, E, J, V, T, Lp
    d.margin, d.max_violation, d.no_imgnorm, d.use_abs = float(margin), int(max_violation), int(no_imgnorm), int(use_abs)
    d.pool = {'mean': 1, 'max': 2}.get(pool, 0)
    d.compute_dtype = {'f32': 0, 'bf16': 1}[compute_dtype]
    return d


def listener_params(tensors):
    lp = ListenerParams()
    for field, key in LISTENER_PARAM_FIELDS:
        t = tensors.get(key)
        setattr(lp, field, _p(t) if t is not None else None)
    return lp


def listener_fwd(dims, params, fc_feats, labels=None, masks=None, seq=None, stv=None, L=None,
                 only_one_retrieval='off', want_emb=False, ws=None, out=None, soft=None):
    """-> dict(loss_rows f32[B], loss_sum f32[1], img_emb, cap_emb, ws, io)."""
    dev = fc_feats.device
    nbytes = lib.cic_listener_ws_bytes(C.byref(dims))
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if out is None:
        out = dict(loss_rows=torch.empty(dims.B, device=dev), loss_sum=torch.empty(1, device=dev),
                   img_emb=torch.empty(dims.B, dims.J, device=dev) if want_emb else None,
                   cap_emb=torch.empty(dims.B, dims.J, device=dev) if want_emb else None)
    out['ws'] = ws
    io = ListenerIO()
    io.fc_feats, io.labels, io.masks = _p(fc_feats), _p(labels), _p(masks)
    io.seq, io.stv, io.L, io.soft = _p(seq), _p(stv), _p(L), _p(soft)
    io.only_one_retrieval = ONLY_ONE[only_one_retrieval]
    io.loss_rows, io.loss_sum = _p(out['loss_rows']), _p(out['loss_sum'])
    io.img_emb_out, io.cap_emb_out = _p(out['img_emb']), _p(out['cap_emb'])
    io.device_shared = 1 if DEVICE_SHARED[0] else 0
    io.status = status.ptr(dev)
    check(lib.cic_listener_fwd(C.byref(dims), C.byref(params), C.byref(io), ws.data_ptr(), ws.numel(), stream()),
          'cic_listener_fwd')
    out['io'] = io
    out['_keep'] = (fc_feats, labels, masks, seq, stv, L, soft)     # keep inputs alive for the backward call
    return out


def listener_bwd(dims, params, fwd, g_rows=None, g_scalar=None, grads=None, d_onehot=None, g_scale=1.0):
    bio = ListenerBwdIO()
    bio.g_rows, bio.g_scalar = _p(g_rows), _p(g_scalar)
    bio.g_scale = float(g_scale)
    gp = listener_params(grads) if grads is not None else None
    bio.grads = C.pointer(gp) if gp is not None else None
    bio.d_onehot = _p(d_onehot)
    ws = fwd['ws']
    check(lib.cic_listener_bwd(C.byref(dims), C.byref(params), C.byref(fwd['io']), C.byref(bio), ws.data_ptr(),
                               ws.numel(), stream()), 'cic_listener_bwd')


def pack_refs(gts, device):
    """data['gts'] (list of int arrays [ncap, Tr]) -> (refs i32[R,Tr], ref_off i32[n_images+1]) on device."""
    import numpy as np
    off = np.zeros(len(gts) + 1, np.int32)
    off[1:] = np.cumsum([len(g) for g in gts])
    refs = np.concatenate([np.asarray(g) for g in gts], 0).astype(np.int32)
    ref_off = torch.from_numpy(off).to(device)
    ref_off.max_refs = int(max(len(g) for g in gts))      # known here, on the host: saves the reward a fallback launch
    return torch.from_numpy(np.ascontiguousarray(refs)).to(device), ref_off


CIDERD_MAX_VOCAB = 32766      # n-gram keys pack 15 bits per token (ids 0 .. V+1)


def ciderd_reward(gen, L_gen, greedy, L_greedy, refs, ref_off, spi=None, debug=False, ws=None,
                  vocab_size=CIDERD_MAX_VOCAB):
    """get_self_critical_reward on the GPU.  gen/greedy: i32[B,T]; refs/ref_off from pack_refs.
    -> dict(scores f64[2B], reward f32[B], stats f64[2] = (mean sampled score, cider_greedy)).
    vocab_size: the captions' vocabulary (token ids 0 .. V+1); the call is refused (CicError) for V > 32766 and a
    token outside the declared range turns the scores into NaN instead of aliasing another word."""
    dev = gen.device
    B, T = gen.shape
    n_images = ref_off.numel() - 1
    spi = spi or B // n_images
    R, Tr = refs.shape
    nbytes = lib.cic_ciderd_ws_bytes(B, R)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = dict(scores=torch.empty(2 * B, dtype=torch.float64, device=dev), reward=torch.empty(B, device=dev),
               stats=torch.empty(2, dtype=torch.float64, device=dev), ws=ws)
    a = CiderdArgs()
    a.B, a.T, a.n_images, a.spi, a.R, a.Tr = B, T, n_images, spi, R, Tr
    a.gen, a.L_gen, a.greedy, a.L_greedy = _p(gen), _p(L_gen), _p(greedy), _p(L_greedy)
    a.refs, a.ref_off = _p(refs), _p(ref_off)
    a.scores, a.reward, a.stats = _p(out['scores']), _p(out['reward']), _p(out['stats'])
    a.vocab_size = int(vocab_size)
    a.max_refs_per_image = int(getattr(ref_off, 'max_refs', 0))
    if debug:
        S = 2 * B + R
        out['dbg_keys'] = torch.zeros(S, 64, dtype=torch.int64, device=dev)
        out['dbg_cnt'] = torch.zeros(S, 64, dtype=torch.int32, device=dev)
        out['dbg_df'] = torch.zeros(S, 64, dtype=torch.int32, device=dev)
        out['dbg_nuniq'] = torch.zeros(S, dtype=torch.int32, device=dev)
        a.dbg_keys, a.dbg_cnt, a.dbg_df, a.dbg_nuniq = (_p(out['dbg_keys']), _p(out['dbg_cnt']), _p(out['dbg_df']),
                                                      _p(out['dbg_nuniq']))
    check(lib.cic_ciderd_reward(C.byref(a), ws.data_ptr(), ws.numel(), stream()), 'cic_ciderd_reward')
    return out


def seq_loss(slp, seq, L, coef, coef_sign, weight, dslp=None, accumulate=False, loss_out=None, combine=None):
    """combine = (weighted_terms, self_weight): this term is the LAST of the step's loss sum; the kernel then also writes
    sum_i w_i * term_i + self_weight * (this term) - what loss_combine would, without its launch - and the call returns
    (loss_out, total) with total a 0-dim tensor."""
    B, T = slp.shape
    if loss_out is None:
        loss_out = torch.empty(1, device=slp.device)
    if combine is None:
        check(lib.cic_seq_loss(_p(slp), _p(seq), _p(L), _p(coef), float(coef_sign), float(weight), B, T, _p(loss_out),
                               _p(dslp), int(accumulate), stream()), 'cic_seq_loss')
        return loss_out
    prior, w_self = combine
    k = len(prior)
    assert all(t.dtype == torch.float32 and t.is_cuda for _, t in prior)
    ptrs = (C.c_void_p * max(k, 1))(*[_p(t) for _, t in prior])
    ws = (C.c_float * max(k, 1))(*[float(w) for w, _ in prior])
    total = torch.empty(1, device=slp.device)
    check(lib.cic_seq_loss_total(_p(slp), _p(seq), _p(L), _p(coef), float(coef_sign), float(weight), B, T, _p(loss_out),
                                 _p(dslp), int(accumulate), ptrs, ws, k, float(w_self), _p(total), stream()),
          'cic_seq_loss_total')
    out = total[0]
    out._cic_fresh = True
    return loss_out, out


def loss_combine(weighted_terms):
    """[(weight, device scalar tensor f32[>=1])] -> 0-dim tensor sum_i weight_i * term_i[0], one launch."""
    k = len(weighted_terms)
    terms = [t for _, t in weighted_terms]
    assert all(t.dtype == torch.float32 and t.is_cuda for t in terms)
    total = torch.empty(1, device=terms[0].device)
    ptrs = (C.c_void_p * k)(*[_p(t) for t in terms])
    ws = (C.c_float * k)(*[float(w) for w, _ in weighted_terms])
    check(lib.cic_loss_combine(ptrs, ws, k, _p(total), stream()), 'cic_loss_combine')
    out = total[0]
    out._cic_fresh = True          # autograd_glue.EngineLoss need not copy it
    return out


def masked_nll(slp, mask, weight, dslp=None, loss_out=None):
    B, T = slp.shape
    assert mask.stride(1) == 1
    if loss_out is None:
        loss_out = torch.empty(1, device=slp.device)
    check(lib.cic_masked_nll(_p(slp), mask.data_ptr(), mask.stride(0), float(weight), B, T, _p(loss_out), _p(dslp),
                             stream()), 'cic_masked_nll')
    return loss_out


def clamp_adam(p, g, m, v, lr, step, grad_clip=0.1, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0,
               zero_grad=False, guarded=True):
    """guarded: the update reads the device's sticky status word (status.py) and leaves p, m, v and g untouched while it is
    set - a gradient poisoned by a timed-out hand-off never reaches the weights."""
    check(lib.cic_clamp_adam_guarded(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay,
                                     grad_clip, int(step), grad_scale, 1 if zero_grad else 0,
                                     status.ptr(p.device) if guarded else None, stream()), 'cic_clamp_adam_guarded')


# ---- use_bn: BatchNorm1d folded into att_embed's Linear (cic.h, csrc/batchnorm.hip) ----------------------------------
lib.cic_bn_stats.argtypes = [P, P, C.c_int, C.c_int, P, P, P, P]
lib.cic_bn_running_update.argtypes = [P, P, P, C.c_float, C.c_int, P, P, P]
lib.cic_bn_fold_fwd.argtypes = [P] * 6 + [C.c_float, C.c_int, C.c_int, P, P, P]
lib.cic_bn_fold_bwd.argtypes = [P] * 7 + [C.c_float, C.c_int, C.c_int, P, P, P, P, P]
for _f in (lib.cic_bn_stats, lib.cic_bn_running_update, lib.cic_bn_fold_fwd, lib.cic_bn_fold_bwd):
    _f.restype = C.c_int


def bn_stats(x, masks, rows, D, mean, var, count):
    assert x.is_contiguous() and masks.is_contiguous() and masks.numel() == rows and x.numel() == rows * D
    check(lib.cic_bn_stats(_p(x), _p(masks), rows, D, _p(mean), _p(var), _p(count), stream()), 'cic_bn_stats')


def bn_running_update(mean, var, count, momentum, running_mean, running_var):
    check(lib.cic_bn_running_update(_p(mean), _p(var), _p(count), float(momentum), mean.numel(), _p(running_mean),
                                    _p(running_var), stream()), 'cic_bn_running_update')


def bn_fold_fwd(W, bias, gamma, beta, mean, var, eps, W_folded, bias_folded):
    H, D = W.shape
    check(lib.cic_bn_fold_fwd(_p(W), _p(bias), _p(gamma), _p(beta), _p(mean), _p(var), float(eps), H, D, _p(W_folded),
                              _p(bias_folded), stream()), 'cic_bn_fold_fwd')


def bn_fold_bwd(dW_raw, db_raw, W, gamma, beta, mean, var, eps, dW, dbias, dgamma, dbeta):
    H, D = W.shape
    check(lib.cic_bn_fold_bwd(_p(dW_raw), _p(db_raw), _p(W), _p(gamma), _p(beta), _p(mean), _p(var), float(eps), H, D,
                              _p(dW), _p(dbias), _p(dgamma), _p(dbeta), stream()), 'cic_bn_fold_bwd')


TIMED_IDS = {'attn_fwd': 0, 'logit_gemm': 1, 'attn_bwd': 2, 'sampler': 3}
lib.cic_timer_create.restype = C.c_void_p
lib.cic_timer_destroy.argtypes = [C.c_void_p]
lib.cic_timer_destroy.restype = None
lib.cic_timer_reset.argtypes = [C.c_void_p]
lib.cic_timer_collect.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
lib.cic_timer_bracket_overhead.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_void_p]


class KernelTimer:
    """A caller-owned cic_timer: decodes that are handed it (speaker_decode_io(..., timer=...)) bracket the launches of
    their attention / logit / sampler kernels with HIP events on their own stream, forward and backward."""

    def __init__(self):
        self.handle = lib.cic_timer_create()
        if not self.handle:
            raise _lib.CicError('cic_timer_create failed')

    def reset(self):
        check(lib.cic_timer_reset(self.handle), 'cic_timer_reset')

    def collect(self):
        """-> {kernel: dict(ms=total elapsed, n=launches)} since reset(); synchronises the recorded events."""
        out = {}
        for name, i in TIMED_IDS.items():
            ms, n = C.c_double(0.0), C.c_int(0)
            check(lib.cic_timer_collect(self.handle, i, C.byref(ms), C.byref(n)), 'cic_timer_collect')
            out[name] = dict(ms=ms.value, n=n.value)
        return out

    def bracket_overhead_us(self, pairs=200):
        """Average elapsed time of an event pair with nothing between its events: subtract it from a bracketed launch."""
        us = C.c_double(0.0)
        check(lib.cic_timer_bracket_overhead(self.handle, int(pairs), C.byref(us), stream()), 'cic_timer_bracket_overhead')
        return us.value

    def __del__(self):
        h, self.handle = getattr(self, 'handle', None), None
        if h:
            lib.cic_timer_destroy(h)
