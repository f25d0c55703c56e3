"""A/B of whole-MCMC rates on one box: the MrBayes binary `which` (amd | pars | full) on DNA 500 x 20 000, default moves (or `fixed`),
walls of two window lengths for each environment setting, alternating.  usage: mcmc_ab.py <which> <fixed|mix> <ngen_lo> <ngen_hi> ENV=1 [ENV2=1 ...]
Every setting is run against the empty environment; MBAMD_STATS of the long window of each is printed."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
which, mode, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
settings = [{}] + [dict([a.split("=", 1)]) for a in sys.argv[5:]]
binary = {"amd": refrun.REF_MB_AMD, "pars": refrun.REF_MB_AMD_PARS, "full": refrun.REF_MB_AMD_FULL}[which]
st = mbdata.synthetic_states(500, 20000, 4, 7, 0.15, 0.0)
tr = mbtree.random_tree(500, 3, brlen=0.05)
walls = {i: {lo: [], hi: []} for i in range(len(settings))}
stats = {}
for rep in range(2):
    for i, env in enumerate(settings):
        for n in (lo, hi):
            e = dict(env)
            if n == hi and rep == 1:
                e["MBAMD_STATS"] = "1"
            out, wall = refrun.run_mb(binary, refrun.mcmc_nexus(st, tr, n, beagle="dynamic", fixed_topology=(mode == "fixed")), env=e)
            walls[i][n].append(refrun.analysis_seconds(out) or wall)
            if "MBAMD_STATS" in e:
                stats[i] = "\n".join(l for l in out.splitlines() if "[mbamd]" in l)
for i, env in enumerate(settings):
    rate = (hi - lo) / (min(walls[i][hi]) - min(walls[i][lo]))
    print("%-28s %8.0f gen/s   walls %s" % (env or "(default)", rate, {k: ["%.3f" % x for x in v] for k, v in walls[i].items()}))
    print(stats.get(i, ""))
