"""ADAPT gradient-ranking helpers — same names and observable behaviour as
ref:openvqe/common_files/sorted_gradient.py:5-108 (they define the tie-breaking of the operator
selection): exact zeros are dropped, values are ordered by decreasing
magnitude (a negative entry ahead of an equal-magnitude positive one), equal values map back to
pool indices in ascending order, duplicates removed.
"""


def index_without_0(my_list):
    """indices of the entries that are not exactly zero"""
    return [i for i, v in enumerate(my_list) if v != 0]


def value_without_0(my_list):
    """entries that are not exactly zero, original order"""
    return [v for v in my_list if v != 0]


def occurence(my_list):
    """(multiplicity of every value, negative values whose positive twin is also present)"""
    counts = {}
    for v in my_list:
        counts[v] = counts.get(v, 0) + 1
    both = [v for v in counts if v < 0 and -v in counts]
    return counts, both


def abs_sort_desc(my_list):
    """Sort by decreasing |value| keeping signs; the sort is done in place on ``my_list`` like the
    reference (callers pass a fresh list) and the sorted list is returned."""
    negatives = {}
    for v in my_list:
        if v < 0:
            negatives[-v] = negatives.get(-v, 0) + 1
    mags = sorted((abs(v) for v in my_list), reverse=True)
    out = []
    for m in mags:
        if negatives.get(m, 0) > 0:
            negatives[m] -= 1
            out.append(-m)
        else:
            out.append(m)
    my_list[:] = out
    return my_list


def corresponding_index(new_list, new_list_index, sorted_new):
    """pool indices in the order of ``sorted_new``; ties (exactly equal floats) resolve to ascending
    position in ``new_list``; each index reported once."""
    seen = set()
    res = []
    for target in sorted_new:
        for j, v in enumerate(new_list):
            if v == target:
                idx = new_list_index[j]
                if idx not in seen:
                    seen.add(idx)
                    res.append(idx)
    return res


def duplicates(my_list, item):
    return [i for i, x in enumerate(my_list) if x == item]
