#!/usr/bin/env python3
"""Time the device parsimony passes (mbamdPars*):   python tools/pars_time.py [ntaxa npat nstates]
(default 500 x 20 000 DNA, configs[1]'s shape).  A ParsSPR1 move of the reference = 2 down-passes + 2 final passes over
(together) the whole tree + the candidate loop.  The CPU side of the comparison -- the oracle's restatement of the
reference's host loops on one core -- is tests/pars_cpu_time.py (test infrastructure may use the oracle, tools do not);
results are checked here against a numpy restatement of the down-pass."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrbayes_amd import beagle as bg                         # noqa: E402
from mrbayes_amd import parsimony as mp                      # noqa: E402
from mrbayes_amd import tree as mbtree                       # noqa: E402


def main():
    ntaxa, npat, nstates = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (500, 20000, 4)))
    rng = np.random.default_rng(1)
    t = mbtree.random_tree(ntaxa, 3)
    base = rng.integers(0, nstates, size=npat)
    states = np.where(rng.random((ntaxa, npat)) < 0.15, rng.integers(0, nstates, size=(ntaxa, npat)), base[None, :])
    sets = np.zeros((t.n_nodes, npat), dtype=np.uint64)
    sets[:ntaxa] = mp.tip_sets(states, nstates)
    w = np.ones(npat, dtype=np.float32)
    inst = mp.ParsimonyInstance(t.n_nodes, npat, nstates, lib=bg.library())
    for i in range(ntaxa):
        inst.set_sets(i, sets[i])
    inst.set_pattern_weights(w)
    dops, fops = mp.down_pass_ops(t, t.root_left), mp.final_pass_ops(t, t.root_left)
    nodes = [n for n in t.all_down_pass if t.anc[n] >= 0]
    tuples = [[n, t.anc[n], nodes[(7 * n) % len(nodes)], t.anc[nodes[(7 * n) % len(nodes)]]] for n in nodes[:200]]
    # numpy restatement of the Fitch down-pass (one vector operation per tree operation) as the check
    ref = sets.copy()
    total = 0.0
    for d, a, b, _ in dops:
        x = ref[a] & ref[b]
        empty = x == 0
        ref[d] = np.where(empty, ref[a] | ref[b], x)
        total += float(w[empty].sum())
    ok = inst.down_pass(dops) == total and np.array_equal(inst.get_sets(t.root_left), ref[t.root_left])
    want = np.array([w[((ref[a] | ref[b]) & (ref[c] | ref[d])) == 0].sum() for a, b, c, d in tuples], dtype=np.float64)
    ok = ok and np.array_equal(inst.score(tuples), want)
    inst.final_pass(fops)
    if not ok and not os.environ.get("PARS_TIME_NOCHECK"):        # (ablation builds compute nonsense on purpose)
        raise SystemExit("device and the numpy restatement disagree")
    reps = 20
    res = {}
    for name, fn in (("down_pass", lambda: inst.down_pass(dops, want_length=False)), ("final_pass", lambda: inst.final_pass(fops)),
                     ("score_200", lambda: inst.score(tuples)),
                     ("move (down, final, score)", lambda: (inst.down_pass(dops, want_length=False), inst.final_pass(fops), inst.score(tuples)))):
        fn(); inst.score(tuples[:1])
        a = time.perf_counter()
        for _ in range(reps):
            fn()
        inst.score(tuples[:1])                                   # (drains the stream)
        res[name] = (time.perf_counter() - a) / reps
    print("%d taxa x %d patterns, %d states: node-pattern updates per pass %.3g" % (ntaxa, npat, nstates, len(dops) * npat))
    for k, v in res.items():
        print("  device %-28s %.3f ms" % (k, v * 1e3))
    print("  down-pass: %.3g node-pattern updates/s on the device" % (len(dops) * npat / res["down_pass"]))


if __name__ == "__main__":
    main()
