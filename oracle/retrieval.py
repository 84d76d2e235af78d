"""Oracle (test infrastructure): retrieval-rank evaluation of the listener, numpy restatement of
/root/reference/eval_utils.py:545-596 (i2t) and :598-720 (t2i), cosine measure.

images: (cpi*N, K) — every image embedding repeated once per caption, as encode_data stacks them; captions:
(cpi*N, K).  Ranks are positions in ``np.argsort(d)[::-1]``."""
import numpy as np


def metrics(ranks):
    """eval_utils.py:586-590 / :705-713."""
    r1 = 100.0 * len(np.where(ranks < 1)[0]) / len(ranks)
    r5 = 100.0 * len(np.where(ranks < 5)[0]) / len(ranks)
    r10 = 100.0 * len(np.where(ranks < 10)[0]) / len(ranks)
    medr = np.floor(np.median(ranks)) + 1
    meanr = ranks.mean() + 1
    return (r1, r5, r10, medr, meanr)


def i2t(images, captions, npts=None):
    """Image -> text (eval_utils.py:545-596): rank of the best-placed of the image's 5 captions."""
    if npts is None:
        npts = images.shape[0] // 5
    ranks = np.zeros(npts)
    top1 = np.zeros(npts)
    for index in range(npts):
        im = images[5 * index].reshape(1, images.shape[1])
        d = np.dot(im, captions.T).flatten()
        inds = np.argsort(d)[::-1]
        rank = 1e20
        for i in range(5 * index, 5 * index + 5):
            tmp = np.where(inds == i)[0][0]
            if tmp < rank:
                rank = tmp
        ranks[index] = rank
        top1[index] = inds[0]
    return metrics(ranks), (ranks, top1)


def t2i(images, captions, cpi=5, npts=None):
    """Text -> image (eval_utils.py:598-720): rank of the caption's own image among the N images."""
    if npts is None:
        npts = images.shape[0] // cpi
    ims = np.array([images[i] for i in range(0, len(images), cpi)])
    ranks = np.zeros(cpi * npts)
    top1 = np.zeros(cpi * npts)
    for index in range(npts):
        queries = captions[cpi * index:cpi * index + cpi]
        d = np.dot(queries, ims.T)
        for i in range(len(d)):
            inds = np.argsort(d[i])[::-1]
            ranks[cpi * index + i] = np.where(inds == index)[0][0]
            top1[cpi * index + i] = inds[0]
    return metrics(ranks), (ranks, top1)
