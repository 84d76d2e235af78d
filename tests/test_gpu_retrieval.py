"""GPU: retrieval-rank evaluation (eval_utils.i2t / t2i) — one similarity GEMM + counting kernels —
against the ranks recorded from the reference's numpy implementation (integer ranks and top-1 indices bit-exact)
and, at the evaluation size (1000 images x 5 captions), against the oracle."""
import os

import numpy as np
import pytest
import torch

import golden_util as GU

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['retrieval_5cap', 'retrieval_gen_1cap'])
def test_ranks_match_reference(name):
    from cooperativeimagecaptioning_amd import eval_utils as E
    z = dict(np.load(os.path.join(GU.GOLDEN, name + '.npz')))
    cpi = int(z['cpi'])
    if cpi == 5:
        r, (ranks, top1) = E.i2t(z['images'], z['captions'], return_ranks=True)
        np.testing.assert_array_equal(ranks, z['i2t_ranks'])
        np.testing.assert_array_equal(top1, z['i2t_top1'])
        np.testing.assert_allclose(np.array(r), z['i2t_r'])
    data = [{'id': i, 'file_path': str(i)} for i in range(z['images'].shape[0] // cpi)]
    ri, (ranks_i, top1_i), ranking = E.t2i(z['images'], z['captions'], data, return_ranks=True, useGenSent=(cpi == 1))
    np.testing.assert_array_equal(ranks_i, z['t2i_ranks'])
    np.testing.assert_array_equal(top1_i, z['t2i_top1'])
    np.testing.assert_allclose(np.array(ri), z['t2i_r'])
    assert len(ranking) == len(data)


def test_ranks_at_evaluation_size_vs_oracle():
    from cooperativeimagecaptioning_amd import eval_utils as E
    from oracle import retrieval as R
    rs = np.random.RandomState(7)
    N, K = 1000, 1024
    im = rs.randn(N, K).astype(np.float32)
    im /= np.linalg.norm(im, axis=1, keepdims=True)
    cap = 0.15 * np.repeat(im, 5, 0) + rs.randn(5 * N, K).astype(np.float32) / np.sqrt(K)
    cap /= np.linalg.norm(cap, axis=1, keepdims=True)
    images = np.repeat(im, 5, 0)
    r, (ranks, top1) = E.i2t(images, cap, return_ranks=True)
    ro, (ranks_o, top1_o) = R.i2t(images, cap)
    ri, (ranks_i, top1_i), _ = E.t2i(images, cap, None, return_ranks=True)
    rio, (ranks_io, top1_io) = R.t2i(images, cap)
    # f32 MFMA dot vs numpy's blocked f32 dot: a pair of near-equal scores may swap; allow a handful of off-by-one ranks
    assert (ranks != ranks_o).mean() < 0.01 and np.abs(ranks - ranks_o).max() <= 2
    assert (ranks_i != ranks_io).mean() < 0.01 and np.abs(ranks_i - ranks_io).max() <= 2
    assert (top1 != top1_o).mean() < 0.01 and (top1_i != top1_io).mean() < 0.01
    np.testing.assert_allclose(np.array(r), np.array(ro), atol=0.5)
    np.testing.assert_allclose(np.array(ri), np.array(rio), atol=0.5)
