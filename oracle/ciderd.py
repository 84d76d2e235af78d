"""Oracle (test infrastructure): self-critical CIDEr-D reward on integer tokens.

Follows /root/reference/misc/rewards.py:26-72 and
/root/reference/cider/pyciderevalcap/ciderD/ciderD_scorer.py:13-215 (df mode
"corpus", n=4, sigma=6).  The reference turns token rows into space-joined
strings and counts n-grams of the split words; that is the same as counting
n-grams of the integer tokens, which is what this file does (dict of int tuples,
same insertion order, so the fp64 sums are formed in the same order).
"""
import math
from collections import defaultdict

import numpy as np

N = 4
SIGMA = 6.0


def row_to_tokens(arr):
    """array_to_str, rewards.py:26-32: tokens up to AND INCLUDING the first 0."""
    out = []
    for v in arr:
        out.append(int(v))
        if int(v) == 0:
            break
    return out


def precook(words, n=N):
    """precook, ciderD_scorer.py:13-30: counts of all 1..n-grams."""
    counts = defaultdict(int)
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return counts


def compute_doc_freq(crefs):
    """compute_doc_freq, ciderD_scorer.py:106-118: one count per entry whose ref set has the n-gram."""
    df = defaultdict(float)
    for refs in crefs:
        for ngram in set(ng for ref in refs for ng in ref.keys()):
            df[ngram] += 1
    return df


def counts2vec(cnts, df, ref_len, n=N):
    """counts2vec, ciderD_scorer.py:121-146 (length counts BIGRAMS, :143-144)."""
    vec = [defaultdict(float) for _ in range(n)]
    length = 0
    norm = [0.0 for _ in range(n)]
    for (ngram, tf) in cnts.items():
        d = np.log(max(1.0, df[ngram]))
        k = len(ngram) - 1
        vec[k][ngram] = float(tf) * (ref_len - d)
        norm[k] += pow(vec[k][ngram], 2)
        if k == 1:
            length += tf
    norm = [np.sqrt(x) for x in norm]
    return vec, norm, length


def sim(vec_hyp, vec_ref, norm_hyp, norm_ref, length_hyp, length_ref, n=N, sigma=SIGMA):
    """sim, ciderD_scorer.py:148-175: clipped cosine + Gaussian length penalty."""
    delta = float(length_hyp - length_ref)
    val = np.array([0.0 for _ in range(n)])
    for k in range(n):
        for (ngram, _) in vec_hyp[k].items():
            val[k] += min(vec_hyp[k][ngram], vec_ref[k][ngram]) * vec_ref[k][ngram]
        if (norm_hyp[k] != 0) and (norm_ref[k] != 0):
            val[k] /= (norm_hyp[k] * norm_ref[k])
        assert not math.isnan(val[k])
        val[k] *= np.e ** (-(delta ** 2) / (2 * sigma ** 2))
    return val


def ciderd_scores(hyps, refs_per_hyp, n=N, sigma=SIGMA):
    """CiderScorer.compute_score in "corpus" mode, ciderD_scorer.py:177-215.

    hyps: list of token lists; refs_per_hyp: list (same length) of lists of token lists.
    Returns (mean, scores f64[len(hyps)])."""
    ctest = [precook(h, n) for h in hyps]
    crefs = [[precook(r, n) for r in refs] for refs in refs_per_hyp]
    df = compute_doc_freq(crefs)
    ref_len = np.log(float(len(crefs)))                                # :178-179
    scores = []
    for test, refs in zip(ctest, crefs):                               # :184-201
        vec, norm, length = counts2vec(test, df, ref_len, n)
        score = np.array([0.0 for _ in range(n)])
        for ref in refs:
            vr, nr, lr = counts2vec(ref, df, ref_len, n)
            score += sim(vec, vr, norm, nr, length, lr, n, sigma)
        score_avg = np.mean(score)
        score_avg /= len(refs)
        score_avg *= 10.0
        scores.append(score_avg)
    return np.mean(np.array(scores)), np.array(scores)


def get_self_critical_reward(gts, gen_result, greedy_res, return_gen_scores=False):
    """get_self_critical_reward, rewards.py:34-72.

    gts: list (one per image) of int arrays [ncap, seq_len]; gen_result/greedy_res:
    int arrays [B, L] / [B, L'] (B = len(gts) * seq_per_img)."""
    gen_result = np.asarray(gen_result)
    greedy_res = np.asarray(greedy_res)
    B = gen_result.shape[0]
    seq_per_img = B // len(gts)
    hyps = [row_to_tokens(gen_result[i]) for i in range(B)] + \
           [row_to_tokens(greedy_res[i]) for i in range(B)]            # :43-46
    g = [[row_to_tokens(gts[i][j]) for j in range(len(gts[i]))] for i in range(len(gts))]  # :48-51
    refs = [g[i % B // seq_per_img] for i in range(2 * B)]             # :55
    _, scores = ciderd_scores(hyps, refs)
    cider_gen = scores[:B]
    cider_greedy = scores[B:].mean()
    diff = scores[:B] - scores[B:]                                     # :66
    if not return_gen_scores:
        return diff, cider_greedy
    return cider_gen, diff, cider_greedy


def ngram_count_table(tokens, n=N):
    """Flat, sorted (ngram tuple, count) list of one sentence — the integer table the
    GPU kernel's counts are compared with bit for bit."""
    c = precook(tokens, n)
    return sorted(c.items())
