"""Mirror of the retrieval-rank part of the reference's eval_utils.py (i2t :545-596, t2i :598-720, cosine
measure) on the GPU: same signatures and return values, numpy in / numpy out.  The similarity matrix is one f32
MFMA product and the ranks come from counting kernels (cic_retrieval_ranks) instead of one np.argsort per query."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, stream

lib.cic_retrieval_ws_bytes.argtypes = [C.c_int, C.c_int]
lib.cic_retrieval_ws_bytes.restype = C.c_size_t
lib.cic_retrieval_ranks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]
lib.cic_retrieval_ranks.restype = C.c_int


def _metrics(ranks):
    r1 = 100.0 * len(np.where(ranks < 1)[0]) / len(ranks)
    r5 = 100.0 * len(np.where(ranks < 5)[0]) / len(ranks)
    r10 = 100.0 * len(np.where(ranks < 10)[0]) / len(ranks)
    medr = np.floor(np.median(ranks)) + 1
    meanr = ranks.mean() + 1
    return (r1, r5, r10, medr, meanr)


def retrieval_ranks(ims, caps, cpi, want_i2t=True, want_t2i=True):
    """ims [N,J], caps [N*cpi,J] device tensors -> dict of int32 device tensors (ranks / top1 per direction)."""
    assert ims.is_cuda and caps.is_cuda, 'cic: the retrieval evaluation runs on the GPU only'
    ims, caps = ims.float().contiguous(), caps.float().contiguous()
    N, J = ims.shape
    assert caps.shape == (N * cpi, J)
    ws = torch.empty(lib.cic_retrieval_ws_bytes(N, cpi), dtype=torch.uint8, device=ims.device)
    out = {}
    if want_i2t:
        out['ranks_i2t'] = torch.empty(N, dtype=torch.int32, device=ims.device)
        out['top1_i2t'] = torch.empty(N, dtype=torch.int32, device=ims.device)
    if want_t2i:
        out['ranks_t2i'] = torch.empty(N * cpi, dtype=torch.int32, device=ims.device)
        out['top1_t2i'] = torch.empty(N * cpi, dtype=torch.int32, device=ims.device)
    p = lambda k: out[k].data_ptr() if k in out else None   # noqa: E731
    check(lib.cic_retrieval_ranks(ims.data_ptr(), caps.data_ptr(), N, cpi, J, p('ranks_i2t'), p('top1_i2t'), p('ranks_t2i'),
                                  p('top1_t2i'), ws.data_ptr(), ws.numel(), stream()), 'cic_retrieval_ranks')
    return out


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def i2t(images, captions, npts=None, measure='cosine', return_ranks=False):
    """eval_utils.py:545-596.  images: (5N, K) with every image repeated 5 times, captions: (5N, K)."""
    if measure != 'cosine':
        raise NotImplementedError("only the 'cosine' measure (the one the scripts use)")
    if npts is None:
        npts = images.shape[0] // 5
    out = retrieval_ranks(_dev(images[0:5 * npts:5]), _dev(captions[:5 * npts]), 5, want_t2i=False)
    ranks = out['ranks_i2t'].cpu().numpy().astype(np.float64)
    top1 = out['top1_i2t'].cpu().numpy().astype(np.float64)
    r = _metrics(ranks)
    return (r, (ranks, top1)) if return_ranks else r


def t2i(images, captions, images_data=None, npts=None, measure='cosine', return_ranks=False, useGenSent=False):
    """eval_utils.py:598-720 (ranks and metrics; the per-image ranking dictionary with ids / file paths is built
    from images_data when given, as the reference does)."""
    if measure != 'cosine':
        raise NotImplementedError("only the 'cosine' measure (the one the scripts use)")
    cpi = 1 if useGenSent else 5
    if npts is None:
        npts = images.shape[0] // cpi
    dev = retrieval_ranks(_dev(images[0:cpi * npts:cpi]), _dev(captions[:cpi * npts]), cpi, want_i2t=False)
    ranks = dev['ranks_t2i'].cpu().numpy().astype(np.float64)
    top1 = dev['top1_t2i'].cpu().numpy().astype(np.float64)
    r = _metrics(ranks)
    if useGenSent:
        print('\n validation rank stats for generated captions: \n r1 {} \n r5 {} \n r10 {} \n medr {} \n meanr {} \n \n'
              .format(*r))
    if not return_ranks:
        return r
    images_ranking = {}
    if images_data is not None:
        for index in range(npts):
            for i in range(cpi):
                entry = {'image_id': images_data[index]['id'], 'rank_correct_im': ranks[cpi * index + i],
                         'file_path': images_data[index]['file_path']}
                if useGenSent:
                    images_ranking[index] = entry
                else:
                    images_ranking.setdefault(index, {})['caption' + str(i)] = entry
    return r, (ranks, top1), images_ranking
