"""Mirror of the reference's eval_utils.py: eval_split (:89-280), encode_data (:283-412), evalrank (:415-542), i2t (:545-596), t2i (:598-720, cosine
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


# ---------------------------------------------------------------------------------------------------------------
# The evaluation drivers around the device engines (eval_utils.py:89-542).  language_eval (:19-86: PTB tokenizer,
# METEOR, SPICE - Java) stays out of scope.
# ---------------------------------------------------------------------------------------------------------------
def _to_dev(x, rows=None):
    """A batch entry of the loader (numpy array or tensor, host or device) as a device tensor; rows: keep these."""
    if x is None:
        return None
    t = torch.as_tensor(x)
    if rows is not None:
        t = t[torch.as_tensor(rows)]
    return t.cuda(non_blocking=True)


def _budget(data, n, num_images):
    """-> (ix0, ix1): position in the split and the number of images to evaluate (eval_utils.py:205-209,387-391)."""
    ix1 = data['bounds']['it_max']
    if num_images != -1:
        ix1 = min(ix1, num_images)
    return data['bounds']['it_pos_now'], ix1


def eval_split(model, loader, eval_kwargs={}, annFile=None, useGenSent=False):
    """eval_utils.py:89-280: validation losses, one generated caption per image (greedy or beam search on the device:
    model.sample -> cic_speaker_decode_fwd / cic_speaker_beam_search) and, with rank_eval, the retrieval ranks.
    Returns (losses, predictions, lang_stats) as the reference does."""
    from .misc import utils
    verbose = eval_kwargs.get('verbose', True)
    num_images = eval_kwargs.get('num_images', eval_kwargs.get('val_images_use', -1))
    split = eval_kwargs.get('split', 'val')
    rank_eval = eval_kwargs.get('rank_eval', 0)
    phase = eval_kwargs.get('phase', 0)
    use_att = eval_kwargs.get('use_att', True)
    if eval_kwargs.get('language_eval', 0) == 1:
        raise NotImplementedError('language_eval runs the Java PTB tokenizer / METEOR / SPICE of coco-caption '
                                  '(eval_utils.py:19-86): out of scope here; score `predictions` with that tool')
    model.eval()
    np.random.seed(123)
    loader.reset_iterator(split)
    n, loss_evals, losses, predictions = 0, 1e-8, {}, []
    with torch.no_grad():
        while True:
            data = loader.get_batch(split)
            n += loader.batch_size
            att_masks = _to_dev(data.get('att_masks')) if use_att else None
            if data.get('labels', None) is not None:                              # the model's loss on the batch
                loss = model(_to_dev(data['fc_feats']), _to_dev(data['labels']), _to_dev(data['masks']), data,
                             _to_dev(data['att_feats']) if use_att else None, att_masks)
                loss = float(loss)
                for k, v in model.loss().items():
                    losses[k] = losses.get(k, 0) + float(v)
                loss_evals += 1
            rows = np.arange(loader.batch_size) * loader.seq_per_img              # one feature row per image (:160-170)
            fc = _to_dev(data['fc_feats'], rows)
            att = _to_dev(data['att_feats'], rows) if use_att else None
            am = _to_dev(data.get('att_masks'), rows) if use_att else None
            seq, _ = model.sample(fc, att, am, opt=eval_kwargs)
            sents = utils.decode_sequence(loader.get_vocab(), seq)
            for k, sent in enumerate(sents):
                entry = {'image_id': data['infos'][k]['id'], 'caption': sent}
                if eval_kwargs.get('dump_path', 0) == 1:
                    entry['file_name'] = data['infos'][k]['file_path']
                predictions.append(entry)
                if verbose:
                    print('image %s: %s' % (entry['image_id'], entry['caption']))
            ix0, ix1 = _budget(data, n, num_images)
            for _ in range(n - ix1):
                predictions.pop()
            if verbose:
                print('evaluating validation preformance... %d/%d (%f)' % (ix0 - 1, ix1, loss))
            if data['bounds']['wrapped'] or (num_images >= 0 and n >= num_images):
                break
    lang_stats = {}
    ranks, gt_ranks = {}, {}
    if useGenSent:
        if rank_eval:
            ranks = evalrank(model, loader, eval_kwargs, useGenSent)
            if not annFile:
                gt_ranks = evalrank(model, loader, eval_kwargs, False)
    elif rank_eval:
        if phase == 1:
            old_split = eval_kwargs.get('split')
            for split_rank in ['val', 'test']:
                eval_kwargs['split'] = split_rank
                ranks[split_rank] = evalrank(model, loader, eval_kwargs, useGenSent)
            eval_kwargs['split'] = old_split
        else:
            ranks = evalrank(model, loader, eval_kwargs, useGenSent)
    model.train()
    losses = {k: v / loss_evals for k, v in losses.items()}
    losses.update(ranks)
    if useGenSent and not annFile:
        losses['gt_ranks'] = gt_ranks
    return losses, predictions, lang_stats


def encode_data(model, loader, eval_kwargs={}, useGenSent=False):
    """eval_utils.py:283-412: the listener's embeddings of a split - image and ground-truth captions (5 per image), or
    image and ONE greedily generated caption per image - through cic_listener_fwd (want_emb).  Returns
    (img_embs, cap_embs, images_data) as stacked numpy arrays, what i2t / t2i consume."""
    num_images = eval_kwargs.get('num_images', eval_kwargs.get('val_images_use', -1))
    split = eval_kwargs.get('split', 'val')
    model.eval()
    loader_seq_per_img = loader.seq_per_img
    loader.seq_per_img = 5 if (not useGenSent and loader.dataset in ['coco', 'flickr8k', 'flickr30k']) else 1
    loader.reset_iterator(split)
    n, img_embs, cap_embs, images_data = 0, [], [], []
    with torch.no_grad():
        while True:
            data = loader.get_batch(split)
            n += loader.batch_size
            if not useGenSent:                                                    # ground-truth captions
                res = model.vse.run(_to_dev(data['fc_feats']), labels=_to_dev(data['labels']),
                                    masks=_to_dev(data['masks']), want_emb=True, slot=7)
            else:                                                                  # one greedy caption per image
                rows = np.arange(loader.batch_size) * loader.seq_per_img
                fc = _to_dev(data['fc_feats'], rows)
                gen = model.caption_generator.decode(_to_dev(data['att_feats'], rows), _to_dev(data.get('att_masks'), rows),
                                                     'greedy', tag='greedy')
                gen.stv = None                                                     # plain token input: <bos> + seq, masks
                res = model.vse.run(fc, decode=gen, want_emb=True, slot=7)       # [1, 1, (seq > 0)[:, :-1]] (:360-368)
            img_emb, cap_emb = res.fwd['img_emb'], res.fwd['cap_emb']
            ix0, ix1 = _budget(data, n, num_images)
            if n > ix1:
                keep = (ix1 - n) * loader.seq_per_img
                img_emb, cap_emb = img_emb[:keep], cap_emb[:keep]
                images_data += data['infos'][:(ix1 - n)]
            else:
                images_data += data['infos']
            img_embs.append(img_emb.cpu().numpy().copy())
            cap_embs.append(cap_emb.cpu().numpy().copy())
            if data['bounds']['wrapped'] or (num_images >= 0 and n >= num_images):
                break
            print('%d/%d' % (n, ix1))
    img_embs, cap_embs = np.vstack(img_embs), np.vstack(cap_embs)
    assert img_embs.shape[0] == ix1 * loader.seq_per_img
    loader.seq_per_img = loader_seq_per_img
    return img_embs, cap_embs, images_data


def _evalrank_fold5(img_embs, cap_embs, images_data, useGenSent):
    """fold5 = 1 (eval_utils.py:450-487,503-540): the MSCOCO 5k test set as five folds of 1000 images, metrics averaged.
    The reference's own fold5 branch cannot execute - it calls t2i() without its required `images_data` argument (TypeError)
    and returns an `images_ranking` it never assigned - so there is nothing to record from it: this follows its evident intent
    (per fold i2t + t2i on rows [i*5000, (i+1)*5000) - 1000 images x 5 caption rows; with generated captions 1000 rows of
    each -, the thirteen numbers [r1 r5 r10 medr meanr] x 2 + ar + ari + rsum averaged over the folds) through the same
    device rank kernels as the full evaluation.  Parity: by construction on i2t / t2i, which are pinned (tests/golden/retrieval_*)."""
    per = 1000 * (1 if useGenSent else 5)
    nfold = img_embs.shape[0] // per
    if nfold < 1:
        raise ValueError(f'fold5 needs at least {per} embedded rows (1000 images), got {img_embs.shape[0]}')
    results, first_ranking = [], None
    for i in range(min(5, nfold)):
        sl = slice(i * per, (i + 1) * per)
        infos = images_data[i * 1000:(i + 1) * 1000]
        ri, rti, ranking = t2i(img_embs[sl], cap_embs[sl], infos, measure='cosine', return_ranks=True, useGenSent=useGenSent)
        if useGenSent:
            r = (0.0,) * 5                      # one caption per image: the reference ranks text -> image only there
        else:
            r, rt = i2t(img_embs[sl], cap_embs[sl], measure='cosine', return_ranks=True)
            print('Image to text: %.1f, %.1f, %.1f, %.1f, %.1f' % r)
        print('Text to image: %.1f, %.1f, %.1f, %.1f, %.1f' % ri)
        ar, ari = (r[0] + r[1] + r[2]) / 3, (ri[0] + ri[1] + ri[2]) / 3
        rsum = r[0] + r[1] + r[2] + ri[0] + ri[1] + ri[2]
        print('rsum: %.1f ar: %.1f ari: %.1f' % (rsum, ar, ari))
        results.append(list(r) + list(ri) + [ar, ari, rsum])
        if i == 0:
            first_ranking = ranking
    m = tuple(np.array(results).mean(axis=0).flatten())
    print('-----------------------------------')
    print('Mean metrics: ')
    print('rsum: %.1f' % m[12])
    if not useGenSent:
        print('Average i2t Recall: %.1f' % m[10])
        print('Image to text: %.1f %.1f %.1f %.1f %.1f' % m[:5])
    print('Average t2i Recall: %.1f' % m[11])
    print('Text to image: %.1f %.1f %.1f %.1f %.1f' % m[5:10])
    out = {'rsum': m[12], 't2i_ar': m[11], 't2i_r1': m[5], 't2i_r5': m[6], 't2i_r10': m[7], 't2i_medr': m[8], 't2i_meanr': m[9],
           'folds': len(results)}
    if useGenSent:
        out['images_ranking'] = first_ranking
    else:
        out.update({'i2t_ar': m[10], 'i2t_r1': m[0], 'i2t_r5': m[1], 'i2t_r10': m[2], 'i2t_medr': m[3], 'i2t_meanr': m[4],
                    'gt_images_ranking': first_ranking})
    return out


def evalrank(model, loader, eval_kwargs={}, useGenSent=False):
    """eval_utils.py:415-542: the full evaluation, or with eval_kwargs['fold5'] the five 1000-image folds (_evalrank_fold5)."""
    if eval_kwargs.get('fold5', 0):
        img_embs, cap_embs, images_data = encode_data(model, loader, eval_kwargs, useGenSent) if useGenSent \
            else encode_data(model, loader, eval_kwargs)
        print('Images: %d, Captions: %d' % (img_embs.shape[0] / (1 if useGenSent else 5), cap_embs.shape[0]))
        return _evalrank_fold5(img_embs, cap_embs, images_data, useGenSent)
    if not useGenSent:
        print('Computing results useGenSent = False...')
        img_embs, cap_embs, images_data = encode_data(model, loader, eval_kwargs)
        print('Images: %d, Captions: %d' % (img_embs.shape[0] / 5, cap_embs.shape[0]))
        r, rt = i2t(img_embs, cap_embs, measure='cosine', return_ranks=True)
        ri, rti, images_ranking = t2i(img_embs, cap_embs, images_data, measure='cosine', return_ranks=True)
        ar, ari = (r[0] + r[1] + r[2]) / 3, (ri[0] + ri[1] + ri[2]) / 3
        rsum = r[0] + r[1] + r[2] + ri[0] + ri[1] + ri[2]
        print('rsum: %.1f' % rsum)
        print('Average i2t Recall: %.1f' % ar)
        print('Image to text: %.1f %.1f %.1f %.1f %.1f' % r)
        print('Average t2i Recall: %.1f' % ari)
        print('Text to image: %.1f %.1f %.1f %.1f %.1f' % ri)
        return {'rsum': rsum, 'i2t_ar': ar, 't2i_ar': ari, 'i2t_r1': r[0], 'i2t_r5': r[1], 'i2t_r10': r[2], 'i2t_medr': r[3],
                'i2t_meanr': r[4], 't2i_r1': ri[0], 't2i_r5': ri[1], 't2i_r10': ri[2], 't2i_medr': ri[3], 't2i_meanr': ri[4],
                'gt_images_ranking': images_ranking}
    print('Computing results for generated samples...')
    img_embs, cap_embs, images_data = encode_data(model, loader, eval_kwargs, useGenSent)
    print('Images: %d, Captions: %d' % (img_embs.shape[0], cap_embs.shape[0]))
    ri, rti, images_ranking = t2i(img_embs, cap_embs, images_data, measure='cosine', return_ranks=True, useGenSent=useGenSent)
    ari = (ri[0] + ri[1] + ri[2]) / 3
    rsum = ri[0] + ri[1] + ri[2]
    print('rsum: %.1f' % rsum)
    print('Average t2i Recall: %.1f' % ari)
    print('Text to image: %.1f %.1f %.1f %.1f %.1f' % ri)
    return {'rsum': rsum, 't2i_ar': ari, 't2i_r1': ri[0], 't2i_r5': ri[1], 't2i_r10': ri[2], 't2i_medr': ri[3],
            't2i_meanr': ri[4], 'images_ranking': images_ranking}
