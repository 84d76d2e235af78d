"""ctypes binding of libcic_hip.so (the C ABI declared in include/cic.h).

The library is the product: there is no eager/CPU fallback.  If the shared object is
missing or a symbol is absent, importing this module raises.
"""
import ctypes as C
import os
import re

import torch  # noqa: F401  (loads the HIP runtime the library links against first)

_HERE = os.path.dirname(os.path.abspath(__file__))
# CIC_HIP_LIB: tools/ point this at the development build (libcic_hip_dev.so, tools/_devlib.py); unset = the product
_LIB_PATH = os.environ.get('CIC_HIP_LIB') or os.path.join(_HERE, 'libcic_hip.so')
_HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'cic.h')


class CicError(RuntimeError):
    pass


def _load():
    if not os.path.exists(_LIB_PATH):
        raise CicError(
            f'{_LIB_PATH} not found: build it with `python -m cooperativeimagecaptioning_amd.build` '
            f'(hipcc --offload-arch=gfx950).  There is no fallback path.')
    return C.CDLL(_LIB_PATH)


lib = _load()

c_f32p = C.c_void_p   # device pointers travel as integers
c_ptr = C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [('M', C.c_int), ('N', C.c_int), ('K', C.c_int),
                ('A', c_ptr), ('lda', C.c_int), ('a_kc', C.c_int),
                ('B', c_ptr), ('ldb', C.c_int), ('b_kc', C.c_int),
                ('K2', C.c_int),
                ('A2', c_ptr), ('lda2', C.c_int),
                ('B2', c_ptr), ('ldb2', C.c_int),
                ('C', c_ptr), ('ldc', C.c_int),
                ('bias', c_ptr), ('accumulate', C.c_int), ('relu', C.c_int),
                ('sum_order_free', C.c_int), ('c_is_zero', C.c_int), ('colsum_A', c_ptr), ('colsum_A2', c_ptr), ('rows_blk', C.c_int), ('A_b', c_ptr), ('A2_b', c_ptr), ('C_b', c_ptr),
                ('n_split', C.c_int), ('B2_tail', c_ptr), ('ldb2_tail', C.c_int), ('bias_tail', c_ptr),
                ('C_tail', c_ptr), ('C_tail_b', c_ptr), ('ldc_tail', C.c_int),
                ('epi', c_ptr),        # fused vocabulary epilogue (the decode engine's logit product); NULL here
                ('precision', C.c_int),   # 0 f32 accuracy (default), 1 f32-input MFMA only, 2 bf16 operands
                ('B_parts', c_ptr),       # B pre-split by cic_split_bf16x3 (the logit weights), or NULL
                ('live', c_ptr), ('live_b', c_ptr), ('live_min', C.c_int),   # decode-loop early stop (engines only); NULL here
                ('B2_parts', c_ptr), ('B2_tail_parts', c_ptr)]               # bf16 images of B2 / B2_tail (CIC_PRECISION_BF16) or NULL


class SamplerArgs(C.Structure):
    _fields_ = [('logits', c_ptr), ('B', C.c_int), ('V1', C.c_int), ('ld', C.c_int),
                ('mode', C.c_int), ('temp', C.c_float),
                ('U', c_ptr), ('ldu', C.c_int),
                ('pick', c_ptr), ('ss_u', c_ptr), ('ss_prob', C.c_float), ('ss_pick', c_ptr),
                ('soft', c_ptr), ('ld_soft', C.c_int), ('ps_u', c_ptr), ('ps_prob', C.c_float),
                ('decoding_constraint', C.c_int), ('step', C.c_int),
                ('unfinished', c_ptr), ('it_next', c_ptr), ('seq', c_ptr), ('slp', c_ptr),
                ('stv', c_ptr), ('seq_ld', C.c_int), ('any_unfinished', c_ptr),
                ('emb_w', c_ptr), ('emb_x', c_ptr), ('emb_keep', c_ptr), ('emb_scale', C.c_float), ('emb_dim', C.c_int),
                ('emb_plain', C.c_int)]


class SpeakerDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('B', 'K', 'D', 'H', 'E', 'A', 'V', 'T')] + [('p_drop', C.c_float), ('compute_dtype', C.c_int)]


SPEAKER_PARAM_FIELDS = [
    # (struct field, reference state-dict key)
    ('embed_w', 'embed.0.weight'),
    ('att_embed_w', 'att_embed.0.weight'), ('att_embed_b', 'att_embed.0.bias'),
    ('logit_w', 'logit.weight'), ('logit_b', 'logit.bias'),
    ('ctx2att_w', 'ctx2att.weight'), ('ctx2att_b', 'ctx2att.bias'),
    ('a2c_w', 'core.a2c.weight'), ('a2c_b', 'core.a2c.bias'),
    ('i2h_w', 'core.i2h.weight'), ('i2h_b', 'core.i2h.bias'),
    ('h2h_w', 'core.h2h.weight'), ('h2h_b', 'core.h2h.bias'),
    ('h2att_w', 'core.attention.h2att.weight'), ('h2att_b', 'core.attention.h2att.bias'),
    ('alpha_w', 'core.attention.alpha_net.weight'), ('alpha_b', 'core.attention.alpha_net.bias'),
]


class SpeakerParams(C.Structure):
    _fields_ = [(f, c_ptr) for f, _ in SPEAKER_PARAM_FIELDS]


class DecodeIO(C.Structure):
    _fields_ = [('mode', C.c_int), ('temp', C.c_float), ('decoding_constraint', C.c_int),
                ('att_pre', c_ptr), ('att_masks', c_ptr), ('att_keep', c_ptr), ('x_keep', c_ptr),
                ('out_keep', c_ptr), ('U', c_ptr), ('pick', c_ptr),
                ('ps_u', c_ptr), ('ps_prob', C.c_float), ('soft_raw', c_ptr), ('xpre', c_ptr), ('soft_out', c_ptr),
                ('ss_u', c_ptr), ('ss_prob', C.c_float),
                ('ss_pick', c_ptr), ('fc_mode', C.c_int), ('x0', c_ptr), ('first_token', c_ptr),
                ('seq', c_ptr), ('slp', c_ptr), ('stv', c_ptr), ('L', c_ptr),
                ('u_philox', C.c_int), ('u_seed', C.c_uint64), ('u_offset', C.c_uint64), ('timer', c_ptr),
                ('device_shared', C.c_int), ('status', c_ptr)]


class DecodeBwdIO(C.Structure):
    _fields_ = [('d_onehot', c_ptr), ('dslp', c_ptr), ('grads', C.POINTER(SpeakerParams)), ('att_raw', c_ptr),
                ('d_x0', c_ptr), ('phase', C.c_int), ('device_shared', C.c_int), ('dslp_scale', c_ptr)]


BWD_ALL, BWD_LOGIT, BWD_REST, BWD_LOOP, BWD_TAIL = 0, 1, 2, 3, 4


class BeamIO(C.Structure):
    _fields_ = [('beam', C.c_int), ('decoding_constraint', C.c_int), ('att_pre', c_ptr), ('att_masks', c_ptr),
                ('seq', c_ptr), ('logps', c_ptr), ('score', c_ptr)]


class CiderdArgs(C.Structure):
    _fields_ = [('B', C.c_int), ('T', C.c_int), ('n_images', C.c_int), ('spi', C.c_int), ('R', C.c_int),
                ('Tr', C.c_int), ('gen', c_ptr), ('L_gen', c_ptr), ('greedy', c_ptr), ('L_greedy', c_ptr),
                ('refs', c_ptr), ('ref_off', c_ptr), ('scores', c_ptr), ('reward', c_ptr), ('stats', c_ptr),
                ('dbg_keys', c_ptr), ('dbg_cnt', c_ptr), ('dbg_df', c_ptr), ('dbg_nuniq', c_ptr),
                ('vocab_size', C.c_int), ('max_refs_per_image', C.c_int)]


class ListenerDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('B', 'F', 'E', 'J', 'V', 'T', 'Lp')] + [
        ('margin', C.c_float), ('max_violation', C.c_int), ('no_imgnorm', C.c_int), ('use_abs', C.c_int),
        ('pool', C.c_int), ('compute_dtype', C.c_int)]


LISTENER_PARAM_FIELDS = [
    ('img_fc_w', 'img_enc.fc.weight'), ('img_fc_b', 'img_enc.fc.bias'),
    ('embed_w', 'txt_enc.embed.weight'),
    ('w_ih', 'txt_enc.rnn.weight_ih_l0'), ('w_hh', 'txt_enc.rnn.weight_hh_l0'),
    ('b_ih', 'txt_enc.rnn.bias_ih_l0'), ('b_hh', 'txt_enc.rnn.bias_hh_l0'),
]


class ListenerParams(C.Structure):
    _fields_ = [(f, c_ptr) for f, _ in LISTENER_PARAM_FIELDS]


class ListenerIO(C.Structure):
    _fields_ = [('fc_feats', c_ptr), ('labels', c_ptr), ('masks', c_ptr), ('seq', c_ptr), ('stv', c_ptr),
                ('L', c_ptr), ('soft', c_ptr), ('only_one_retrieval', C.c_int), ('loss_rows', c_ptr),
                ('loss_sum', c_ptr),
                ('img_emb_out', c_ptr), ('cap_emb_out', c_ptr), ('device_shared', C.c_int), ('status', c_ptr)]


class ListenerBwdIO(C.Structure):
    _fields_ = [('g_rows', c_ptr), ('g_scalar', c_ptr), ('grads', C.POINTER(ListenerParams)),
                ('d_onehot', c_ptr), ('g_scale', C.c_float)]


SAMPLE_NONE, SAMPLE_GREEDY, SAMPLE_MULTINOMIAL, SAMPLE_GUMBEL_ST, SAMPLE_MULTINOMIAL_ST, SAMPLE_TEACHER = range(6)
SAMPLE_GUMBEL_PS, SAMPLE_MULTINOMIAL_PS = 6, 7


def declared_symbols():
    """Every function name declared in include/cic.h."""
    with open(_HEADER) as f:
        src = f.read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cic_[a-z0-9_]+)\s*\(', src)))


def check_exports():
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise CicError('libcic_hip.so lacks symbols declared in cic.h: ' + ', '.join(missing))


lib.cic_last_error.restype = C.c_char_p
lib.cic_version.restype = C.c_int


def check(rc, what=''):
    if rc != 0:
        raise CicError(f'{what} failed (rc={rc}): {lib.cic_last_error().decode()}')


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), 'cic: tensor must be contiguous'
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
