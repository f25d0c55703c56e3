#!/usr/bin/env python3
"""Time the device parsimony passes (mbamdPars*) next to the oracle's restatement of the reference's
host loops on one core:   python tools/pars_time.py [ntaxa npat nstates]   (default 500 x 20 000 DNA, configs[1]'s shape).
A ParsSPR1 move of the reference = 2 down-passes + 2 final passes over (together) the whole tree + the candidate loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrbayes_amd import beagle as bg                         # noqa: E402
from mrbayes_amd import parsimony as mp                      # noqa: E402
from mrbayes_amd import tree as mbtree                       # noqa: E402
from tests import oracle_lib as ol                           # noqa: E402  (the checker / CPU baseline, not the product)


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
    ref = sets.copy()
    t0 = time.perf_counter(); total, _ = ol.pars_down(ref, dops, w); t1 = time.perf_counter()
    ol.pars_final(ref, fops, npat); t2 = time.perf_counter()
    sc = ol.pars_score(ref, tuples, w); t3 = time.perf_counter()
    ok = inst.down_pass(dops) == total
    inst.final_pass(fops)
    ok = ok and np.array_equal(inst.score(tuples), sc) and np.array_equal(inst.get_sets(t.root_left), ref[t.root_left])
    if not ok and not os.environ.get("PARS_TIME_NOCHECK"):        # (ablation builds compute nonsense on purpose)
        raise SystemExit("device and oracle disagree")
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
    print("  oracle (reference loops, 1 core): down %.2f ms, final %.2f ms, 200 candidates %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    for k, v in res.items():
        print("  device %-28s %.3f ms" % (k, v * 1e3))
    print("  down-pass: %.3g node-pattern updates/s on the device, %.3g on the host" % (len(dops) * npat / res["down_pass"], len(dops) * npat / (t1 - t0)))


if __name__ == "__main__":
    main()
