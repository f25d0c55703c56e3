#!/usr/bin/env python3
"""Full-size in-process parity check of the device parsimony scorer on the GPU: the reference + binding (oracle/_ref/mb_amd_pars)
on the 500 x 20 000 bench alignment, default move mix, with MBAMD_PARS_CHECK=1 -- every GetParsDP / GetParsFP / candidate
loop also runs in the reference's own host functions and is compared word for word inside the process.
    python tools/pars_fullsize_check.py [ngen]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                               # noqa: E402
from mrbayes_amd import data as mbdata, tree as mbtree     # noqa: E402
from tools import refrun                                   # noqa: E402

ngen = int(sys.argv[1]) if len(sys.argv) > 1 else 400
with open(os.path.join(bench.GOLD, bench.CONFIGS["c2"][0] + ".json")) as fh:
    gold = json.load(fh)
sy = gold["synthetic"]
st = mbdata.synthetic_states(sy["ntaxa"], sy["nsites"], 4, sy["seed"], sy["p_mut"], sy["p_gap"])
tr = mbtree.parse_newick(gold["newick"])
out, wall = refrun.run_mb(refrun.REF_MB_AMD_PARS, refrun.mcmc_nexus(st, tr, ngen, beagle="dynamic"),
                          env={"MBAMD_PARS_CHECK": "1", "MBAMD_COMPRESS_CHECK": "1"})
ok = "Analysis completed" in out
print("completed" if ok else out[-2000:])
for line in out.split("\n"):
    if "parsimony check" in line or "compression check" in line or "Impl Name" in line:
        print(line.strip())
m = re.search(r"parsimony check: (\d+) comparisons", out)
sys.exit(0 if ok and m and int(m.group(1)) > 1000 else 1)
