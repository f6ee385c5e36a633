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


# Frequency-ordered predicate counts (class 0 = background count placeholder) that the reference hard-codes for its
# median-based under-sampling rates (SHA_GCL_extra/extra_function_utils.py:185-203).  Dataset statistics, kept as data.
PREDICATE_COUNTS = {
    "VG": [3024465, 109355, 67144, 47326, 31347, 21748, 15300, 10011, 11059, 10764, 6712, 5086, 4810, 3757, 4260, 3167,
           2273, 1829, 1603, 1413, 1225, 793, 809, 676, 352, 663, 752, 565, 504, 644, 601, 551, 460, 394, 379, 397, 429,
           364, 333, 299, 270, 234, 171, 208, 163, 157, 151, 71, 114, 44, 4],
    "GQA": [200000, 64218, 47205, 32126, 25203, 21104, 15890, 15676, 7688, 6966, 6596, 6044, 5250, 4260, 4180, 4131, 2859,
            2559, 2368, 2351, 2134, 1673, 1532, 1373, 1273, 1175, 1139, 1123, 1077, 941, 916, 849, 835, 808, 782, 767, 628,
            603, 569, 540, 494, 416, 412, 412, 398, 395, 394, 390, 345, 327, 302, 301, 292, 275, 270, 267, 267, 264, 258,
            251, 233, 233, 229, 224, 215, 214, 209, 204, 198, 195, 192, 191, 185, 181, 176, 158, 158, 154, 151, 148, 143,
            136, 131, 130, 130, 128, 127, 125, 124, 124, 121, 118, 112, 112, 106, 105, 104, 103, 102, 52, 52],
}


def sample_rate_matrix(dataset, sizes):
    """generate_sample_rate_vector_sep2 (extra_function_utils.py:185-257), one row per group g (classes up to the
    cumulative bound hi_g): with med = median of the counts of the group's own classes (lo_g, hi_g], a class whose count
    exceeds med keeps the fraction med / count (at least 0.01; the background entry 10x that), every other class 1.0.
    Own classes and the classes of earlier groups compare against med; later classes against the largest count the
    group's own list holds (which includes the background count, so they all get 1.0 in practice)."""
    import statistics
    counts = PREDICATE_COUNTS[dataset]
    bounds, total = [], 0
    for n in sizes:
        total += n
        bounds.append(total)
    out, prev = [], 0
    for hi in bounds:
        row = [0.0] * len(counts)
        own = [counts[0]] + counts[prev + 1:hi + 1]
        med = float(statistics.median(own[1:]))

        def rate(c, background=False):
            r = med / c
            if background:
                r *= 10.0
            return max(r, 0.01)

        for j, c in enumerate(own):
            idx = 0 if j == 0 else j + prev
            row[idx] = rate(c, background=(j == 0)) if c > med else 1.0
        for j in range(1, prev + 1):                      # classes of the earlier groups
            row[j] = rate(counts[j]) if counts[j] > med else 1.0
        for j in range(hi + 1, len(counts)):              # classes of the later groups
            c = counts[j]
            row[j] = rate(c, background=(j == hi + 1)) if c > max(own) else 1.0
        out.append(row)
        prev = hi
    return out
