"""Generates tests/golden/sumtree_ref.json from the REFERENCE SumTree<float>
(sum_tree/sum_tree/include/sum_tree.h compiled into oracle/_ref/libsumtree_ref.so by
oracle/Makefile). Runs in this container only; only the vectors are committed.

Leaf values are dyadic (k/16) so every float sum in the tree is exact and the reference's
running-difference internals (sum_tree.h:80-90) equal an exact rebuild; the quantile grids
include 0, 1.0 and the bucket edges.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle_py import RefTree  # noqa: E402


def main():
    rng = np.random.default_rng(7)
    cases = []
    for cap in (4, 8, 7, 9, 1024):
        t = RefTree(cap)
        case = dict(requested_capacity=cap, capacity=int(t.cap), ops=[])
        n_leaves = t.cap
        for rnd in range(3):
            n = min(n_leaves, [n_leaves, max(1, n_leaves // 2), 3][rnd])
            idx = rng.permutation(n_leaves)[:n].astype(np.int64)  # unique within a call (OpenMP loop is racy otherwise)
            val = (rng.integers(0, 200, n) / 16.0).astype(np.float32)
            t.update(idx, val)
            grid = np.concatenate([np.linspace(0, 1, 33), rng.random(31)]).astype(np.float32)
            sidx, sval = t.sample(grid)
            case["ops"].append(dict(update_idx=idx.tolist(), update_val=val.tolist(), total=t.total(),
                                    leaves=t.get(np.arange(n_leaves)).tolist(), quantiles=grid.tolist(),
                                    sample_idx=sidx.tolist(), sample_val=sval.tolist()))
        cases.append(case)
    # the gtest known answers (sum_tree/sum_tree/src/test_sum_tree.cc:105-117), re-measured on the reference build
    t = RefTree(4)
    t.update([0, 1, 2, 3], [1.0, 2.0, 3.0, 4.0])
    qs = [0.0, 0.099, 0.1, 0.299, 0.3, 0.599, 0.6, 1.0]
    gtest = dict(quantiles=qs, indices=[t.get_index(q) for q in qs], total=t.total())
    # C-8: key 1.0 lands on the last leaf even when it is empty
    t8 = RefTree(8)
    t8.update([0, 1, 2, 3], [0.6] * 4)
    c8 = dict(index_of_1=t8.get_index(1.0))
    with open(os.path.join(HERE, "sumtree_ref.json"), "w") as f:
        json.dump(dict(cases=cases, gtest=gtest, c8=c8), f)
    print("wrote sumtree_ref.json", gtest, c8)


if __name__ == "__main__":
    main()
