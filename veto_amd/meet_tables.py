"""MEET group tables as data.

The reference's splits (SHA_GCL_extra/group_chosen_function.py:6-94) are contiguous ranges of the
frequency-ordered predicate ids 1..50 (VG) / 1..100 (GQA); only the group SIZES matter to the
predictor: head k is Linear(576, size_k + 2) (roi_relation_predictors.py:3726-3728) and
incre_idx_list[c] = 1-based group of class c, 0 for background
(SHA_GCL_extra/extra_function_utils.py:39-77)."""

GROUP_SIZES = {
    ("VG", "divide3"): [3, 3, 8, 6, 20, 10],
    ("VG", "divide4"): [4, 6, 9, 19, 12],
    ("VG", "divide5"): [4, 8, 10, 28],
    ("VG", "divide7new"): [2, 4, 5, 6, 8, 10, 15],
    ("VG", "average"): [10, 10, 10, 10, 10],
    ("GQA", "divide3"): [4, 4, 11, 16, 31, 34],
    ("GQA", "divide4"): [5, 10, 20, 65],
    ("GQA", "divide5"): [7, 14, 28, 51],
    ("GQA", "average"): [20, 20, 20, 20, 20],
}

NUM_CLASSES = {"VG": (151, 51), "GQA": (201, 101)}  # (objects, predicates) incl. background


def group_sizes(dataset, split):
    try:
        return list(GROUP_SIZES[(dataset, split)])
    except KeyError:
        raise ValueError("unknown MEET group split %r for dataset %r" % (split, dataset))


def incre_idx_list(sizes):
    out = [0]
    for g, n in enumerate(sizes):
        out.extend([g + 1] * n)
    return out
