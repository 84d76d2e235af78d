"""Thin tensor-level wrappers over the C ABI (include/cic.h): torch tensors in, kernels
launched on torch's current HIP stream.  No computation happens in Python here."""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check, ptr, stream, GemmArgs, SamplerArgs

P, I, F, L64, U64 = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint64


def _sig(name, argtypes):
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = C.c_int
    return fn


_uniform = _sig('cic_uniform_f32', [P, L64, U64, U64, P])
_keep = _sig('cic_dropout_keep_u8', [P, L64, F, U64, U64, P])
_keep_multi = _sig('cic_dropout_keep_u8_multi', [C.POINTER(P), C.POINTER(L64), C.POINTER(U64), I, F, U64, P])
_gemm = _sig('cic_gemm_f32', [C.POINTER(GemmArgs), P])
# every entry point gets its signature here or in engine.py: ctypes would pass an undeclared Python int as a 32-bit int
_sig('cic_gemm_logit_parts', [C.POINTER(GemmArgs)])
_sig('cic_gemm_split_ok', [C.POINTER(GemmArgs)])
_split3 = _sig('cic_split_bf16x3', [P, L64, P, P])
_round1 = _sig('cic_round_bf16', [P, L64, P, P])
_sig('cic_gemm_f32_timed', [C.POINTER(GemmArgs), I, C.POINTER(C.c_double), P])
_sig('cic_logit_partials', [P, I, I, I, P, I, P])
_sig('cic_attn_fwd_timed', [P] * 7 + [I] * 5 + [P, L64, C.POINTER(C.c_double), P])
_colsum = _sig('cic_colsum_f32', [P, I, I, I, P, I, P])
_attn_fwd = _sig('cic_attn_fwd', [P] * 9 + [I] * 4 + [P])
_cell_fwd = _sig('cic_cell_fwd', [P, P, P, F, P, P, P, I, I, P])
_embed_fwd = _sig('cic_embed_fwd', [P, P, P, F, P, I, I, P])
_apply_keep = _sig('cic_apply_keep', [P, P, F, P, L64, P])
_lss = _sig('cic_logsoftmax_sample', [C.POINTER(SamplerArgs), P])
_finalize_len = _sig('cic_finalize_len', [P, I, P, P])


def _dev(t):
    assert t.is_cuda, 'cic ops run on the GPU only (no CPU fallback)'
    return t


def uniform_(out, seed, offset=0):
    check(_uniform(ptr(_dev(out)), out.numel(), seed, offset, stream()), 'cic_uniform_f32')
    return out


def dropout_keep_(keep, p, seed, offset=0):
    assert keep.dtype == torch.uint8
    check(_keep(ptr(_dev(keep)), keep.numel(), float(p), seed, offset, stream()), 'cic_dropout_keep_u8')
    return keep


def dropout_keep_multi_(keeps, p, seed, offsets):
    """Several keep masks in one launch; mask i is what dropout_keep_(keeps[i], p, seed, offsets[i]) writes."""
    k = len(keeps)
    assert k == len(offsets) and all(t.dtype == torch.uint8 for t in keeps)
    ptrs = (P * k)(*[ptr(_dev(t)) for t in keeps])
    ns = (L64 * k)(*[t.numel() for t in keeps])
    offs = (U64 * k)(*offsets)
    check(_keep_multi(ptrs, ns, offs, k, float(p), seed, stream()), 'cic_dropout_keep_u8_multi')
    return keeps


def split_bf16x3_(x, parts):
    """parts (int16 / uint16 view, 3 * x.numel() elements) <- the three bf16 parts of f32 x."""
    check(_split3(ptr(_dev(x)), x.numel(), ptr(_dev(parts)), stream()), 'cic_split_bf16x3')
    return parts


def round_bf16_(x, packed):
    """packed (int16 / uint16 view, x.numel() elements) <- bf16(x), round to nearest even."""
    check(_round1(ptr(_dev(x)), x.numel(), ptr(_dev(packed)), stream()), 'cic_round_bf16')
    return packed


def gemm(A, B, C_, a_kc=True, b_kc=True, bias=None, accumulate=False, relu=False, A2=None, B2=None,
         M=None, N=None, K=None, K2=None, sum_order_free=False, precision=0):
    """C = op(A) op(B) (+ op(A2) op(B2)) (+bias) (+C).  2-D tensors; rows may be strided views
    (stride(1) == 1).  See cic_gemm_f32 in include/cic.h."""
    def ld(t):
        assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32
        return t.stride(0)
    g = GemmArgs()
    if M is None:
        M = A.shape[0] if a_kc else A.shape[1]
    if K is None:
        K = A.shape[1] if a_kc else A.shape[0]
    if N is None:
        N = B.shape[0] if b_kc else B.shape[1]
    g.M, g.N, g.K = M, N, K
    g.A, g.lda, g.a_kc = _dev(A).data_ptr(), ld(A), int(a_kc)
    g.B, g.ldb, g.b_kc = _dev(B).data_ptr(), ld(B), int(b_kc)
    if A2 is not None:
        g.K2 = K2 if K2 is not None else (A2.shape[1] if a_kc else A2.shape[0])
        g.A2, g.lda2 = A2.data_ptr(), ld(A2)
        g.B2, g.ldb2 = B2.data_ptr(), ld(B2)
    else:
        g.K2 = 0
    g.C, g.ldc = _dev(C_).data_ptr(), ld(C_)
    g.bias = bias.data_ptr() if bias is not None else None
    g.accumulate, g.relu = int(accumulate), int(relu)
    g.sum_order_free = int(sum_order_free)
    g.precision = int(precision)      # 0 f32 accuracy, 1 f32-input MFMA only, 2 bf16 operands (cic.h)
    check(_gemm(C.byref(g), stream()), 'cic_gemm_f32')
    return C_


def colsum(X, out, accumulate=False):
    assert X.dim() == 2 and X.stride(1) == 1
    check(_colsum(ptr_any(X), X.shape[0], X.shape[1], X.stride(0), ptr(out), int(accumulate), stream()),
          'cic_colsum_f32')
    return out


def ptr_any(t):
    return t.data_ptr() if t is not None else None


def attn_fwd(att_h, p_att, att, w_alpha, b_alpha, masks, att_res, alpha, dot=None):
    B, K, A = p_att.shape
    H = att.shape[2]
    check(_attn_fwd(ptr(att_h), ptr(p_att), ptr(att), ptr(w_alpha), ptr(b_alpha), ptr(masks), ptr(att_res),
                    ptr(alpha), ptr(dot), B, K, A, H, stream()), 'cic_attn_fwd')


def cell_fwd(pre, c_prev, keep, p_drop, h_new, c_new, out):
    B, H = c_prev.shape
    check(_cell_fwd(ptr(pre), ptr(c_prev), ptr(keep), float(p_drop), ptr(h_new), ptr(c_new), ptr(out), B, H,
                    stream()), 'cic_cell_fwd')


def embed_fwd(E, it, keep, p_drop, x):
    B, Ed = x.shape
    assert it.dtype == torch.int32
    check(_embed_fwd(ptr(E), ptr(it), ptr(keep), float(p_drop), ptr(x), B, Ed, stream()), 'cic_embed_fwd')


def apply_keep(x, keep, p_drop, y):
    check(_apply_keep(ptr(x), ptr(keep), float(p_drop), ptr(y), x.numel(), stream()), 'cic_apply_keep')


def logsoftmax_sample(logits, mode, temp=1.0, U=None, pick=None, decoding_constraint=0, step=1,
                      unfinished=None, it_next=None, seq=None, slp=None, stv=None, any_unfinished=None):
    a = SamplerArgs()
    a.logits, a.B, a.V1, a.ld = logits.data_ptr(), logits.shape[0], logits.shape[1], logits.stride(0)
    a.mode, a.temp = mode, float(temp)
    a.U, a.ldu = (U.data_ptr(), U.stride(0)) if U is not None else (None, 0)
    a.pick = ptr_any(pick)
    a.ss_u, a.ss_prob, a.ss_pick = None, 0.0, None
    a.decoding_constraint = int(decoding_constraint)
    a.step = step
    a.unfinished, a.it_next = ptr_any(unfinished), ptr_any(it_next)
    a.seq, a.slp, a.stv = ptr_any(seq), ptr_any(slp), ptr_any(stv)
    a.seq_ld = seq.stride(0) if seq is not None else 0
    a.any_unfinished = ptr_any(any_unfinished)
    check(_lss(C.byref(a), stream()), 'cic_logsoftmax_sample')


def finalize_len(any_unfinished, T, L):
    check(_finalize_len(ptr(any_unfinished), T, ptr(L), stream()), 'cic_finalize_len')
