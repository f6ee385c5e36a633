"""CPU restatement of the training-time losses and of the MEET expert sampling (the
parts named there: "weighted CE (BETA_LOSS ...), MEET per-group CE with ... expert sampling").

TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else; the product path is veto_amd/csrc/losses.hip behind
veto_ce_loss / veto_meet_sample.

PARITY PINNED: tests/golden/train_*.npz hold, from the reference predictor run in training mode here
(tests/golden/make_golden.py::run_train_losses): the classifier logits it produced, the labels, the losses it
returned and, for MEET, its expert sampling drawn from Python's `random` seeded with 1.
"""
import numpy as np


def weighted_ce(logits, labels, weight=None):
    """nn.CrossEntropyLoss(weight=w)(logits, labels), reduction 'mean' (roi_relation_predictors.py:4067-4068,4133):
    sum_i w[y_i] * (logsumexp(z_i) - z_i[y_i]) / sum_i w[y_i].  Returns (loss, dlogits) in float64."""
    z = np.asarray(logits, dtype=np.float64)
    y = np.asarray(labels, dtype=np.int64)
    w = np.ones(z.shape[1]) if weight is None else np.asarray(weight, dtype=np.float64)
    m = z.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(z - m).sum(1))
    wi = w[y]
    loss = float((wi * (lse - z[np.arange(len(y)), y])).sum() / wi.sum())
    p = np.exp(z - lse[:, None])
    p[np.arange(len(y)), y] -= 1.0
    return loss, p * (wi / wi.sum())[:, None]


class PyRandomStream:
    """The exact stream of Python's `random` module (MT19937, CPython's random_random / getrandbits /
    _randbelow_with_getrandbits) on top of 32-bit words, so that it can be replayed from a block of raw words."""

    def __init__(self, words):
        self.words, self.pos = words, 0

    def _u32(self):
        v = int(self.words[self.pos])
        self.pos += 1
        return v

    def random(self):
        a, b = self._u32() >> 5, self._u32() >> 6
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0)

    def randint0(self, n):          # random.randint(0, n - 1)
        k = n.bit_length()
        r = self._u32() >> (32 - k)
        while r >= n:
            r = self._u32() >> (32 - k)
        return r


def meet_sampling(labels, incre_idx_list, sample_rate_matrix, num_groups, stream):
    """VETOPredictor_MEET.forward, training branch, ZERO_LABEL_PADDING_MODE 'rand_insert'
    (roi_relation_predictors.py:3940-3969): per relation, in order: a background label goes to ONE random group
    (random.randint); a foreground label with group id g = incre_idx_list[label] draws u = random.random() and walks
    a = G .. 1: the first a with u <= sample_rate_matrix[a-1][label] or a < g puts the relation into groups 0..a-1.
    Returns the list of per-group row index lists (cur_chosen_matrix)."""
    chosen = [[] for _ in range(num_groups)]
    for i, lab in enumerate(labels):
        lab = int(lab)
        if lab == 0:
            chosen[stream.randint0(num_groups)].append(i)
            continue
        g = incre_idx_list[lab]
        u = stream.random()
        for j in range(num_groups):
            a = num_groups - j
            if u <= sample_rate_matrix[a - 1][lab] or a < g:
                for k in range(a):
                    chosen[k].append(i)
                break
    return chosen


def meet_group_labels(labels, rows, incre_idx_list, k):
    """Ensemble.forward :3812-3821: inside group k own classes map to 1.. (their position in the group's class
    list + 1), every other foreground class to g_k + 1, background stays 0."""
    own = [c for c, x in enumerate(incre_idx_list) if x == k + 1]
    out = []
    for i in rows:
        lab = int(labels[i])
        out.append(0 if lab == 0 else (own.index(lab) + 1 if lab in own else len(own) + 1))
    return np.array(out, dtype=np.int64)
