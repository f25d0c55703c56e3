#!/usr/bin/env python3
"""CPU side of tools/pars_time.py: the oracle's restatement of the reference's Fitch loops (oracle/pars_oracle.c: GetFitchPartials,
GetParsFP, the candidate loops) timed on one core on the same synthetic shape.   python tests/pars_cpu_time.py [ntaxa npat nstates]
Test infrastructure (it runs the oracle); the reported baseline beside the device numbers, not a target."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrbayes_amd import parsimony as mp                      # noqa: E402  (operation lists only: no device call)
from mrbayes_amd import tree as mbtree                       # noqa: E402
from tests import oracle_lib as ol                           # noqa: E402


def main():
    ntaxa, npat, nstates = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (500, 20000, 4)))
    rng = np.random.default_rng(1)
    t = mbtree.random_tree(ntaxa, 3)
    base = rng.integers(0, nstates, size=npat)
    states = np.where(rng.random((ntaxa, npat)) < 0.15, rng.integers(0, nstates, size=(ntaxa, npat)), base[None, :])
    sets = np.zeros((t.n_nodes, npat), dtype=np.uint64)
    sets[:ntaxa] = mp.tip_sets(states, nstates)
    w = np.ones(npat, dtype=np.float32)
    dops, fops = mp.down_pass_ops(t, t.root_left), mp.final_pass_ops(t, t.root_left)
    nodes = [n for n in t.all_down_pass if t.anc[n] >= 0]
    tuples = [[n, t.anc[n], nodes[(7 * n) % len(nodes)], t.anc[nodes[(7 * n) % len(nodes)]]] for n in nodes[:200]]
    t0 = time.perf_counter(); ol.pars_down(sets, dops, w); t1 = time.perf_counter()
    ol.pars_final(sets, fops, npat); t2 = time.perf_counter()
    ol.pars_score(sets, tuples, w); t3 = time.perf_counter()
    print("%d taxa x %d patterns, %d states -- oracle (the reference's loops on BitsLong sets, one core): down-pass %.2f ms (%.3g node-pattern "
          "updates/s), final pass %.2f ms, 200 candidates %.2f ms" % (ntaxa, npat, nstates, (t1 - t0) * 1e3, len(dops) * npat / (t1 - t0),
                                                                     (t2 - t1) * 1e3, (t3 - t2) * 1e3))


if __name__ == "__main__":
    main()
