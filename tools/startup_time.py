#!/usr/bin/env python3
"""Start-up time of MrBayes (read the data, compress the site patterns, set up the chains, one generation) on the bench
alignments, with the reference's pattern search and with the hash-table binding (integration/mrbayes/mbamd_compress_glue.c):
    python tools/startup_time.py [c2 c4]          native CPU kernels (usebeagle=no): no GPU needed"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                               # noqa: E402
from mrbayes_amd import data as mbdata, tree as mbtree     # noqa: E402
from tools import refrun                                   # noqa: E402

binary = os.environ.get("MB_BINARY", os.path.join(ROOT, "oracle", "_ref", "mb_amd_pars"))
for cfg in (sys.argv[1:] or ["c2", "c4"]):
    with open(os.path.join(bench.GOLD, bench.CONFIGS[cfg][0] + ".json")) as fh:
        gold = json.load(fh)
    sy = gold["synthetic"]
    st = mbdata.synthetic_states(sy["ntaxa"], sy["nsites"], 4, sy["seed"], sy["p_mut"], sy["p_gap"])
    tr = mbtree.parse_newick(gold["newick"])
    nex = refrun.mcmc_nexus(st, tr, 1, beagle=None, fixed_topology=True).replace("lset nst=6", "set usebeagle=no;\n  lset nst=6")
    row = {}
    for name, env in (("reference search", {"MBAMD_HASH_COMPRESS": "0"}), ("hash table", {})):
        out, wall = refrun.run_mb(binary, nex, env=env)
        assert "Analysis completed" in out, out[-500:]
        row[name] = wall
    print("%s (%d taxa x %d columns): start-up + 1 generation  %.1f s with the reference's search, %.1f s with the hash table"
          % (cfg, sy["ntaxa"], sy["nsites"], row["reference search"], row["hash table"]))
