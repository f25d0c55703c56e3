"""How noisy is a two-point generations/s figure?  Walls of the unmodified MrBayes on the engine (bench.py's fixed-topology DNA case:
500 x 20 000 GTR+G4) at several chain lengths, each twice; optional env assignments NAME=VALUE on the command line; the word
`pars` = the default move mix on the binary with the device-parsimony binding instead."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
env = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
pars = "pars" in sys.argv[1:]
lens = [int(a) for a in sys.argv[1:] if "=" not in a and a != "pars"] or [2000, 12000, 42000]
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bench_c2.json")) as fh:
    gold = json.load(fh)
sy = gold["synthetic"]
st = mbdata.synthetic_states(sy["ntaxa"], sy["nsites"], 4, sy["seed"], sy["p_mut"], sy["p_gap"])
tr = mbtree.parse_newick(gold["newick"])
walls = {}
for rep in range(2):
    for n in lens:
        _, wall = refrun.run_mb(refrun.REF_MB_AMD_PARS if pars else refrun.REF_MB_AMD,
                                refrun.mcmc_nexus(st, tr, n, beagle="dynamic", nchains=1, fixed_topology=not pars), env=env)
        walls.setdefault(n, []).append(wall)
print("default mix, device parsimony" if pars else "fixed topology", env, {n: ["%.3f" % w for w in ws] for n, ws in walls.items()})
lo = lens[0]
for n in lens[1:]:
    rates = [(n - lo) / (b - a) for a in walls[lo] for b in walls[n]]
    print("  %d -> %d: %s gen/s (min wall: %.0f)" % (lo, n, ", ".join("%.0f" % r for r in rates), (n - lo) / (min(walls[n]) - min(walls[lo]))))
