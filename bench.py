#!/usr/bin/env python3
"""Throughput of the hot path on MI355X: one "step" = one full-tree log-likelihood evaluation of one chain
(transition matrices of all 2N-3 branches, N-2 conditional-likelihood updates with rescaling, root
integration) issued through the C ABI exactly as MrBayes' src/mbbeagle.c issues it for a generation whose
move dirties the whole tree (reference src/proposal.c:17682, src/mbbeagle.c:400-537).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--no-cpu-baseline]

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns one heated chain of the same
shape on its own GPU (chain-parallel MCMCMC, the reference's MPI strategy, src/mcmc.c:18331-18384); the only
exchange is the per-generation swap attempt: the two ranks that own the drawn chains send each other (lnL,
lnPrior, heat) over RCCL, everybody else carries on (src/mcmc.c:653-668).  Weak scaling.

Prints ONE JSON line on rank 0.  Unit of work = node-pattern update: one interior-node
conditional-likelihood update of one unique site pattern over all categories, rescale included.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (kind, ntaxa, npatterns, alignment seed, tree seed, description)
    "c2": ("gtr", 500, 20000, 7, 3, "synthetic DNA 500 taxa x 20000 unique patterns, GTR+G4, 1 chain per GPU"),
    "c3": ("wag", 200, 10000, 5, 9, "synthetic amino-acid 200 taxa x 10000 patterns, WAG+G4"),
    "c4": ("gtr", 1000, 50000, 6, 10, "synthetic DNA 1000 taxa x 50000 patterns, GTR+G4, 1 chain per GPU"),
    "c5": ("m3", 100, 5000, 6, 10, "synthetic codon M3 (61 states, 3 omega classes) 100 taxa x 5000 patterns"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3       # fp32 MFMA = fp32 vector peak


def algorithmic_bytes_per_eval(S, K, P, N):
    """each interior CL written once and read once (fp32), node scaler, tips read as 1-byte codes."""
    return P * ((N - 2) * (2 * K * S * 4 + 4) + N * 1)


def flops_per_eval(S, K, P, N):
    return P * (N - 2) * K * (2 * S * S * 2 + S)


def cpu_baseline(kind, ntaxa, seed, tree_seed, budget_patterns):
    """The reference CPU likelihood on this box's host cores, on a bounded sample of the same workload."""
    from mrbayes_amd import data as mbdata
    from mrbayes_amd import tree as mbtree
    from tools import refrun
    if kind == "gtr" and refrun.reference_available():
        npat = budget_patterns
        st = mbdata.synthetic_states(ntaxa, npat, 4, seed, 0.15, 0.0)
        tr = mbtree.random_tree(ntaxa, tree_seed, brlen=0.05)
        lo, hi = 10, 110
        r = refrun.time_reference_dna(st, tr, lo, hi)
        p = r.get("npatterns", npat)
        ups = (ntaxa - 2) * p / r["sec_per_eval"]
        return {"value": ups / 1e6, "unit": "M updates/s", "cores": 1, "kind": "reference",
                "sample": "oracle/_ref/mb (%s kernels), same tree, first %d patterns, fixed-tree two-point CPU time "
                          "ngen=%d vs %d: %.4f s per full-tree evaluation" % (r["calculator"], p, lo, hi, r["sec_per_eval"])}
    # port: the plain-C oracle (scalar restatement), one core
    from mrbayes_amd.division import synthetic_division
    from tests import oracle_lib
    npat = max(64, budget_patterns // (8 if kind == "gtr" else 40 if kind == "wag" else 400))
    div = synthetic_division(kind, ntaxa, npat, seed=seed, tree_seed=tree_seed,
                             golden_dir=os.path.join(ROOT, "tests", "golden"))
    orc = oracle_lib.load()
    t0 = time.time()
    reps = 0
    while time.time() - t0 < 8.0 or reps < 1:
        orc.tree_loglike(div, use_shortcuts=True)
        reps += 1
    dt = (time.time() - t0) / reps
    return {"value": (ntaxa - 2) * npat / dt / 1e6, "unit": "M updates/s", "cores": 1, "kind": "port",
            "sample": "oracle/mb_oracle.c (scalar), same tree, first %d patterns, %d evaluations" % (npat, reps)}


def mcmc_gen_per_s(ntaxa, npat, seed, tree_seed, nchains=1):
    """Secondary metric: generations/s of the UNMODIFIED MrBayes (default move mix, GTR+G4) driving this engine
    (oracle/_ref/mb_amd) next to the same binary's native CPU kernels (oracle/_ref/mb), two-point differenced
    so that parsing / pattern compression / set-up cancel.  Only where the reference binaries were built."""
    from mrbayes_amd import data as mbdata
    from mrbayes_amd import tree as mbtree
    from tools import refrun
    if not (os.path.exists(refrun.REF_MB_AMD) and os.path.exists(refrun.REF_MB)):
        return None
    st = mbdata.synthetic_states(ntaxa, npat, 4, seed, 0.15, 0.0)
    tr = mbtree.random_tree(ntaxa, tree_seed, brlen=0.05)
    out = {"nchains": nchains, "note": "default mix: 11.5% of the moves are ParsSPR/ParsTBR, whose O(taxa x patterns) parsimony "
           "scoring runs on the host in both builds (SURVEY 8(f) item 4); fixed_topology = branch-length and "
           "substitution-parameter moves only (prset topologypr=fixed)"}
    for mix, fixed in (("default_moves", False), ("fixed_topology", True)):
        res = {}
        for tag, binary, beagle, lo, hi in (("engine", refrun.REF_MB_AMD, "dynamic", 2000 if fixed else 500, 22000 if fixed else 2500),
                                            ("reference_cpu", refrun.REF_MB, None, 20, 120 if fixed else 80)):
            walls = []
            for ngen in (lo, hi):
                _, wall = refrun.run_mb(binary, refrun.mcmc_nexus(st, tr, ngen, beagle=beagle, nchains=nchains,
                                                                  fixed_topology=fixed))
                walls.append(wall)
            res[tag] = (hi - lo) / max(walls[1] - walls[0], 1e-9)
            res[tag + "_ngen"] = [lo, hi]
        res["speedup"] = res["engine"] / res["reference_cpu"]
        out[mix] = res
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-patterns", type=int, default=4000)
    ap.add_argument("--mcmc", action="store_true", help="also time whole MCMC generations of the unmodified MrBayes "
                    "binary on this engine vs its native CPU kernels (adds minutes)")
    ap.add_argument("--emulate", action="store_true",
                    help="TEST ONLY: run the control flow on the CPU (host-emulation engine, gloo, tiny workload); "
                         "the JSON line is marked invalid")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    import torch
    emulate = args.emulate
    if not emulate and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    device = torch.device("cpu") if emulate else torch.device("cuda", local_rank)
    if not emulate:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emulate:
            dist_.init_process_group("gloo")
        else:
            dist_.init_process_group("nccl", device_id=device)
        dist = dist_

    from mrbayes_amd import beagle as bg
    from mrbayes_amd import likelihood as lk
    from mrbayes_amd.division import synthetic_division

    kind, ntaxa, npat, seed, tree_seed, desc = CONFIGS[args.config]
    if emulate:
        ntaxa, npat, desc = 16, 200, "EMULATED (invalid as a measurement): " + desc
    div = synthetic_division(kind, ntaxa, npat, seed=seed, tree_seed=tree_seed,
                             golden_dir=os.path.join(ROOT, "tests", "golden"))
    if world > 1:                      # every chain has its own state: perturb the branch lengths per rank
        import random
        rng = random.Random(1000 + rank)
        div.tree.length = [l * (0.8 + 0.4 * rng.random()) for l in div.tree.length]
    if emulate:
        from tests.hostemu import build_emu
        lib = bg.library(build_emu.build())
    else:
        lib = bg.library()
    bd = lk.BeagleDivision(div, lib, nchains=1, scaling=lk.MB_BEAGLE_SCALE_ALWAYS, resource=0 if emulate else local_rank)
    impl = bd.inst.details.implName.decode()
    lnl0 = bd.LogLike(0)
    bd.AcceptMove(0)
    evals = [lk.record_evaluation(bd), lk.record_evaluation(bd)]      # the two alternating buffer-flip states
    S, K, P, N = div.nstates, div.ncat * div.n_cijk_parts, div.npatterns, div.ntaxa
    units_per_step = (N - 2) * P

    from mrbayes_amd import chains as mbchains
    exchange = mbchains.ChainExchange(world, dist=dist, device=device) if dist is not None else None
    if exchange is not None:
        exchange.warm_up()             # open the pairwise connections outside the timed region

    def step(i):
        rc, lnl = evals[i & 1].run()
        if rc != 0:
            raise RuntimeError("evaluation failed with code %d" % rc)
        if exchange is not None:       # per-generation swap attempt: the two ranks that own the chains exchange states (RCCL)
            exchange.swap_generation({rank: lnl})
        return lnl

    def fence():
        if dist is not None:
            dist.barrier()
        if not emulate:
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    bd.inst.kernel_timing(True)
    bd.inst.get_kernel_timing(reset=True)
    fence()
    t0 = time.perf_counter()
    lnl = None
    for i in range(args.steps):
        lnl = step(i)
    fence()
    dt = time.perf_counter() - t0
    kms, klaunches = bd.inst.get_kernel_timing(reset=True)
    bd.inst.kernel_timing(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert abs(lnl - lnl0) <= 1e-9 * abs(lnl0), (lnl, lnl0)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = units_per_step * world * args.steps / dt / 1e6
        k_ms = kms / max(args.steps, 1)               # partials pass (dominant kernel) per evaluation
        abytes = algorithmic_bytes_per_eval(S, K, P, N)
        aflops = flops_per_eval(S, K, P, N)
        if S == 61:
            roof = {"bound": "mfma", "achieved": aflops / (k_ms * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS,
                    "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": abytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["traffic"] = None            # HBM bytes per launch from committed PMC passes of this workload, if any
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                pmc = json.load(fh).get(args.config)
            if pmc and not emulate:
                roof["traffic"] = pmc["traffic_bytes"]
                roof["traffic_source"] = pmc["source"]
                if "mfma_busy_fraction" in pmc:       # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x CUs), same PMC passes
                    roof["mfma_busy_fraction"] = pmc["mfma_busy_fraction"]
        except (OSError, ValueError):
            pass
        roof["kernel"] = impl
        roof["kernel_ms_per_step"] = k_ms
        roof["launches_per_step"] = klaunches / max(args.steps, 1)
        roof["algorithmic_bytes_per_step"] = abytes
        roof["flops_per_step"] = aflops
        if S == 20:
            roof["secondary_tflops"] = aflops / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "site-pattern lnL (node-pattern conditional-likelihood) updates/sec",
            "value": value, "unit": "M updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if not emulate else "INVALID: emulated control-flow test",
            "config": {"workload": desc, "states": S, "categories": K, "patterns": P, "taxa": N,
                       "chains": world, "parallelism": "chain-parallel (1 chain per GPU)" if world > 1 else "1 chain",
                       "units_per_step": units_per_step, "lnL": lnl},
            "pattern_lnl_per_s": P * world * args.steps / dt,
            "full_tree_evals_per_s": world * args.steps / dt,
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1 and not emulate:
            try:
                out["cpu_baseline"] = cpu_baseline(kind, ntaxa, seed, tree_seed, args.cpu_sample_patterns)
                out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            except Exception as exc:          # the baseline is a report, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "M updates/s", "cores": 1, "kind": "port",
                                       "sample": "failed: %r" % (exc,)}
        if args.mcmc and world == 1 and not emulate and kind == "gtr":
            try:
                out["mcmc_gen_per_s"] = mcmc_gen_per_s(ntaxa, npat, seed, tree_seed)
            except Exception as exc:
                out["mcmc_gen_per_s"] = {"error": repr(exc)}
        print(json.dumps(out))
    bd.finalize()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
