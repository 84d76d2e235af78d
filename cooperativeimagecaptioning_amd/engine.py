"""Sequence-level entry points of the C ABI (speaker decode, listener, rewards) wrapped for
torch tensors.  All state lives in caller-visible tensors; the workspace tensor returned by a
forward call is what the matching backward call consumes."""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check, stream, SpeakerDims, SpeakerParams, DecodeIO, SPEAKER_PARAM_FIELDS

P = C.c_void_p

lib.cic_speaker_decode_ws_bytes.argtypes = [C.POINTER(SpeakerDims)]
lib.cic_speaker_decode_ws_bytes.restype = C.c_size_t
lib.cic_speaker_att_embed_fwd.argtypes = [C.POINTER(SpeakerDims), C.POINTER(SpeakerParams), P, P, P]
lib.cic_speaker_att_embed_fwd.restype = C.c_int
lib.cic_speaker_decode_fwd.argtypes = [C.POINTER(SpeakerDims), C.POINTER(SpeakerParams), C.POINTER(DecodeIO), P,
                                       C.c_size_t, P]
lib.cic_speaker_decode_fwd.restype = C.c_int


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'cic engine: CUDA-contiguous tensors only'
    return t.data_ptr()


def speaker_dims(B, K, D, H, E, A, V, T, p_drop):
    d = SpeakerDims()
    d.B, d.K, d.D, d.H, d.E, d.A, d.V, d.T, d.p_drop = B, K, D, H, E, A, V, T, float(p_drop)
    return d


def speaker_params(tensors):
    """tensors: dict keyed by the reference state-dict names -> struct of device pointers."""
    sp = SpeakerParams()
    for field, key in SPEAKER_PARAM_FIELDS:
        t = tensors[key]
        assert t.dtype == torch.float32
        setattr(sp, field, _p(t))
    return sp


def speaker_att_embed_fwd(dims, params, att_raw, att_pre=None):
    if att_pre is None:
        att_pre = torch.empty(dims.B, dims.K, dims.H, device=att_raw.device)
    check(lib.cic_speaker_att_embed_fwd(C.byref(dims), C.byref(params), _p(att_raw), _p(att_pre), stream()),
          'cic_speaker_att_embed_fwd')
    return att_pre


def speaker_decode_fwd(dims, params, att_pre, mode, temp=1.0, att_masks=None, att_keep=None, x_keep=None,
                       out_keep=None, U=None, pick=None, decoding_constraint=0, want_stv=False, ws=None):
    """-> dict(seq i32[B,T], slp f32[B,T], stv f32[B,T]|None, L i32[1], ws)."""
    dev = att_pre.device
    B, T = dims.B, dims.T
    nbytes = lib.cic_speaker_decode_ws_bytes(C.byref(dims))
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = dict(seq=torch.zeros(B, T, dtype=torch.int32, device=dev), slp=torch.zeros(B, T, device=dev),
               stv=torch.ones(B, T, device=dev) if want_stv else None,
               L=torch.zeros(1, dtype=torch.int32, device=dev), ws=ws)
    io = DecodeIO()
    io.mode, io.temp, io.decoding_constraint = mode, float(temp), int(decoding_constraint)
    io.att_pre, io.att_masks = _p(att_pre), _p(att_masks)
    io.att_keep, io.x_keep, io.out_keep = _p(att_keep), _p(x_keep), _p(out_keep)
    io.U, io.pick = _p(U), _p(pick)
    io.seq, io.slp, io.stv, io.L = _p(out['seq']), _p(out['slp']), _p(out['stv']), _p(out['L'])
    check(lib.cic_speaker_decode_fwd(C.byref(dims), C.byref(params), C.byref(io), ws.data_ptr(), ws.numel(),
                                     stream()), 'cic_speaker_decode_fwd')
    return out
