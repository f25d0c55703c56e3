#!/usr/bin/env python3
"""Throughput of the hot path on MI355X: one "step" = one full-tree log-likelihood evaluation of one chain
(transition matrices of all 2N-3 branches, N-2 conditional-likelihood updates with rescaling, root
integration) issued through the C ABI exactly as MrBayes' src/mbbeagle.c issues it for a generation whose
move dirties the whole tree (reference src/proposal.c:17682, src/mbbeagle.c:400-537).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--no-cpu-baseline] [--no-also] [--no-mcmc]

Default workload: BASELINE configs[3]'s shape, synthetic DNA 1000 taxa x 50 000 patterns GTR+G4 (the shape the
>=100x target is quoted on; it fits one GPU), one chain per GPU.  At N=1 the line also carries, under "also", the
same measurement on the other three workloads (configs[1] DNA 500 x 20 000, configs[2] protein 200 x 10 000 WAG+G4,
configs[4]'s codon M3 100 x 5 000) and, under "mcmc_gen_per_s", whole-MCMC generations/s of the
unmodified MrBayes binary on this engine next to its native CPU kernels (short windows; --no-also / --no-mcmc skip
them).  Every workload is a committed golden case (tests/golden/bench_c*.json): alignment seeds, tree and parameters
are those the REAL reference was run on by tools/gen_golden.py, and the lnL printed in "config" is asserted against
the reference's value before anything is timed.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns one heated chain of the same
shape on its own GPU (chain-parallel MCMCMC, the reference's MPI strategy, src/mcmc.c:18331-18384); the only
exchange is the per-generation swap attempt: the two ranks that own the drawn chains send each other (lnL,
lnPrior, heat) over RCCL, everybody else carries on (src/mcmc.c:653-668).  Weak scaling.  (A scaling harness: every
rank replays full evaluations of its own chain state; the swap decision is made but does not change what is replayed.)

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
GOLD = os.path.join(ROOT, "tests", "golden")

CONFIGS = {
    # name: (golden case, model kind, description)
    "c2": ("bench_c2", "gtr", "synthetic DNA 500 taxa x 20000 unique patterns, GTR+G4, 1 chain per GPU"),
    "c3": ("bench_c3", "wag", "synthetic amino-acid 200 taxa x 10000 patterns, WAG+G4"),
    "c4": ("bench_c4", "gtr", "synthetic DNA 1000 taxa x 50000 patterns, GTR+G4, 1 chain per GPU"),
    "c5": ("bench_c5", "m3", "synthetic codon M3 (61 states, 3 omega classes) 100 taxa x 5000 patterns"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3       # fp32 MFMA = fp32 vector peak
REL_FP64 = 2e-6                # lnL vs the reference's fp64 build (tests/engine_checks.py)
ROOFLINE_DEFINITION = (
    "frac = max(HBM bytes moved / 8 TB/s, matrix-core flops issued / 157.3 TFLOP/s) / device time of ALL kernels of one "
    "evaluation (transition matrices + partials + integration, HIP events on the engine's stream: all_kernels_ms_per_step). "
    "Bytes and issued flops per evaluation are PMC counts (rocprofv3 --pmc, profiles/pmc_traffic.json: deterministic for a "
    "workload, taken in their own profiling runs and combined with THIS run's time).  traffic_floor_bytes: the write-once model "
    "(every interior result and its exponents stored once, children read from LDS, tips one byte per pattern); "
    "algorithmic_bytes_per_step: SURVEY 8(d)'s model (every interior CL written once AND read once), about 1.9x what the "
    "kernels move, kept as a count only")


def algorithmic_bytes_per_eval(S, K, P, N):
    """each interior CL written once and read once (fp32), node scaler, tips read as 1-byte codes."""
    return P * ((N - 2) * (2 * K * S * 4 + 4) + N * 1)


def flops_per_eval(S, K, P, N):
    return P * (N - 2) * K * (2 * S * S * 2 + S)


def cpu_baseline(kind, gold, sample_patterns, whole=False):
    """The reference CPU likelihood (oracle/_ref/mb, the real reference's FMA/SSE kernels, one core) on this box's
    host cores, on a bounded sample of the SAME alignment and tree: its first `sample_patterns` columns (cache-resident on
    the host: flattering for the CPU) -- or, whole=True, on the WHOLE alignment with a short two-point window (what "the
    reference CPU likelihood on the same alignment" means; the reference's O(P^2) pattern compression runs twice beside it)."""
    from mrbayes_amd import data as mbdata
    from mrbayes_amd import tree as mbtree
    from tools import refrun
    sy = gold["synthetic"]
    ntaxa = sy["ntaxa"]
    tr = mbtree.parse_newick(gold["newick"])
    if whole and not refrun.reference_available():
        return None
    if refrun.reference_available():
        st = mbdata.synthetic_states(ntaxa, sy["nsites"], sy["nstates"], sy["seed"], sy["p_mut"], sy["p_gap"])
        if whole:
            sample_patterns = st.shape[1]
            lo, hi = {"gtr": (2, 8), "wag": (2, 8), "m3": (1, 4)}[kind]
        else:
            st = st[:, :sample_patterns]
            lo, hi = {"gtr": (10, 70), "wag": (4, 28), "m3": (2, 14)}[kind]
        r = refrun.time_reference(kind, st, tr, lo, hi)
        p = r.get("npatterns", sample_patterns)
        ups = (ntaxa - 2) * p / r["sec_per_eval"]
        return {"value": ups / 1e6, "unit": "M updates/s", "cores": 1, "kind": "reference",
                "sample": "oracle/_ref/mb (%s kernels, single-threaded like the reference), same tree, the first %d of the %d "
                          "patterns of the same alignment, fixed-tree two-point CPU time ngen=%d vs %d: %.4f s per full-tree evaluation"
                          % (r["calculator"], p, sy["nsites"], lo, hi, r["sec_per_eval"])}
    # port: the plain-C oracle (scalar restatement), one core
    from mrbayes_amd.division import division_from_golden
    from tests import oracle_lib
    div = division_from_golden(GOLD, gold["case"])
    npat = max(64, sample_patterns // (8 if kind == "gtr" else 10))
    div.weights = div.weights[:npat]
    div.tip_states = [s[:npat].copy() for s in div.tip_states]
    orc = oracle_lib.load()
    t0 = time.time()
    reps = 0
    while time.time() - t0 < 8.0 or reps < 1:
        orc.tree_loglike(div, use_shortcuts=True)
        reps += 1
    dt = (time.time() - t0) / reps
    return {"value": (ntaxa - 2) * npat / dt / 1e6, "unit": "M updates/s", "cores": 1, "kind": "port",
            "sample": "oracle/mb_oracle.c (scalar), same tree, first %d patterns, %d evaluations" % (npat, reps)}


def mcmc_gen_per_s(gold, nchains=1, quick=True):
    """Secondary metric: generations/s of the UNMODIFIED MrBayes (GTR+G4) driving this engine (oracle/_ref/mb_amd)
    next to the same binary's native CPU kernels (oracle/_ref/mb), two-point differenced so that parsing / pattern
    compression / set-up cancel.  Only where the reference binaries were built."""
    from mrbayes_amd import data as mbdata
    from mrbayes_amd import tree as mbtree
    from tools import refrun
    if not (os.path.exists(refrun.REF_MB_AMD) and os.path.exists(refrun.REF_MB)):
        return None
    sy = gold["synthetic"]
    st = mbdata.synthetic_states(sy["ntaxa"], sy["nsites"], 4, sy["seed"], sy["p_mut"], sy["p_gap"])
    tr = mbtree.parse_newick(gold["newick"])
    out = {"workload": gold["case"], "nchains": nchains,
           "note": "default_moves: MrBayes' default proposal mix; 11.5% of its moves are ParsSPR1/ParsTBR1, whose O(taxa x "
                   "patterns) parsimony scoring runs on the host in the unmodified binary (`engine`, `reference_cpu`) and bounds "
                   "its rate; `engine_device_parsimony` = a PATCHED binary: the same sources with src/proposal.c and src/model.c edited on "
                   "the fly by integration/mrbayes/patches/patch_pars.py to call the device-parsimony binding (SURVEY 8(f) item 4, integration/mrbayes/; "
                   "same proposals and chain as the unmodified binary); "
                   "fixed_topology = branch-length and substitution-parameter moves only (prset topologypr=fixed)"}
    # (round 5: a fixed-topology generation is ~70 us and the 7.6 s of start-up in front of it vary by +-0.3 s from run to run -- a
    #  window of 10 000 generations gave anything from 8 400 to 67 000 generations/s on one box (profiles/r05_path4.txt); the engine's
    #  windows are long enough for +-5 %, and each point is the faster of two runs)
    windows = {(False, "engine"): (300, 1300) if quick else (500, 2500), (True, "engine"): (2000, 42000) if quick else (2000, 82000),
               (False, "reference_cpu"): (10, 40) if quick else (20, 80), (True, "reference_cpu"): (20, 70) if quick else (20, 120)}
    for mix, fixed in (("default_moves", False), ("fixed_topology", True)):
        res = {}
        runs = [("engine", refrun.REF_MB_AMD, "dynamic"), ("reference_cpu", refrun.REF_MB, None)]
        if not fixed and os.path.exists(refrun.REF_MB_AMD_PARS):
            runs.insert(1, ("engine_device_parsimony", refrun.REF_MB_AMD_PARS, "dynamic"))
            windows[(False, "engine_device_parsimony")] = (2000, 27000) if quick else (2000, 52000)
        for tag, binary, beagle in runs:
            lo, hi = windows[(fixed, tag)]
            walls = []
            # (the same number of repeats and the same reduction -- the faster of two runs per window point -- for the engine and for
            #  the reference it is divided by: the start-up in front of a window varies by +-0.3 s from run to run)
            repeats = 2 if (fixed or tag in ("engine_device_parsimony", "reference_cpu")) else 1
            for ngen in (lo, hi):
                walls.append(min(refrun.run_mb(binary, refrun.mcmc_nexus(st, tr, ngen, beagle=beagle, nchains=nchains,
                                                                         fixed_topology=fixed))[1] for _ in range(repeats)))
            res[tag] = (hi - lo) / max(walls[1] - walls[0], 1e-9)
            res[tag + "_ngen"] = [lo, hi]
        res["speedup"] = res["engine"] / res["reference_cpu"]
        if "engine_device_parsimony" in res:
            res["speedup_device_parsimony"] = res["engine_device_parsimony"] / res["reference_cpu"]
        out[mix] = res
    # codon M3 (BASELINE configs[4]'s alignment), fixed topology: a third of the moves change the rate matrices -- the unpatched
    # binary decomposes 3 x 61-state systems on the host (GetEigens), `engine_all_bindings` (a PATCHED binary, oracle/Makefile
    # ref-amd-full: patch_eigen.py + integration/mrbayes/mbamd_eigen_glue.c) on the device, warm-started
    try:
        with open(os.path.join(GOLD, "bench_c5.json")) as fh:
            g5 = json.load(fh)
        s5 = g5["synthetic"]
        st5 = mbdata.synthetic_states(s5["ntaxa"], s5["nsites"], 61, s5["seed"], s5["p_mut"], s5["p_gap"])
        tr5 = mbtree.parse_newick(g5["newick"])
        res = {}
        runs = [("engine", refrun.REF_MB_AMD, "dynamic", (500, 3000)), ("reference_cpu", refrun.REF_MB, None, (3, 10))]
        if os.path.exists(refrun.REF_MB_AMD_FULL):
            runs.insert(1, ("engine_all_bindings", refrun.REF_MB_AMD_FULL, "dynamic", (500, 4500)))
        for tag, binary, beagle, (lo, hi) in runs:
            walls = []
            for ngen in (lo, hi):
                o, wall = refrun.run_mb(binary, refrun.model_nexus("m3", st5, tr5, ngen=ngen, beagle=beagle, fixed_topology=True))
                if "Analysis completed" not in o:
                    raise RuntimeError(o[-800:])
                walls.append(wall)
            res[tag] = (hi - lo) / max(walls[1] - walls[0], 1e-9)
            res[tag + "_ngen"] = [lo, hi]
        res["speedup"] = res["engine"] / res["reference_cpu"]
        out["codon_m3_fixed_topology"] = res
    except Exception as exc:
        out["codon_m3_fixed_topology"] = {"error": repr(exc)[:400]}
    return out


def mpi_mcmc(nranks, emulate=False, full=False):
    """The reference's own chain-parallel design measured on the real binary: MrBayes' MPI build (chains spread over ranks,
    src/mcmc.c:18331-18384; rank r computes on GPU (instance + r) mod #GPUs, src/mbbeagle.c:201-207) on the single-node MPI shim
    (integration/mpi_shim), started as `mbamd_mpirun -n <ranks> mb_amd_mpi_pars` -- the binary with the device-parsimony binding
    (a PATCHED binary: integration/mrbayes/patches/patch_pars.py), MrBayes' default move mix.  Two analyses, generations/s by two-point differencing:
      * BASELINE configs[3]: DNA GTR+G4, nchains=8 (alignment: configs[1]'s 500 x 20 000 unless --mpi-full asks for 1000 x 50 000,
        whose start-up alone takes minutes);
      * BASELINE configs[4]: codon M3 100 x 5 000, nruns=2 x nchains=4.
    Only where the reference binaries were built.  emulate: CPU control-flow test (host-emulation engine, tiny data)."""
    from mrbayes_amd import data as mbdata
    from mrbayes_amd import tree as mbtree
    from tools import refrun
    ref = os.path.join(ROOT, "oracle", "_ref")
    launcher = os.path.join(ref, "mbamd_mpirun")
    binary = os.path.join(ref, "mb_emu_mpi" if emulate else "mb_amd_mpi_pars")
    if not (os.path.exists(launcher) and os.path.exists(binary)):
        return None

    devices = {}                                          # MPI rank -> PCI bus ids of the GPUs its engine instances were created on

    def wall(text):
        out, w = refrun.run_mb(binary, text, timeout=600, argv_prefix=[launcher, "-n", str(nranks)], env={"MBAMD_REPORT_DEVICE": "1"})
        if "Analysis completed" not in out:
            raise RuntimeError(out[-1500:])
        for line in out.splitlines():
            if line.startswith("[mbamd] instance on device"):
                f = line.split()
                devices.setdefault(f[8], set()).add(f[6])
        return w

    def rate(nex_of, lo, hi):
        a, b = wall(nex_of(lo)), wall(nex_of(hi))
        return (hi - lo) / max(b - a, 1e-9)

    cases = []
    if emulate:
        st = mbdata.synthetic_states(8, 60, 4, 11, 0.15, 0.0)
        tr = mbtree.random_tree(8, 12, brlen=0.05)
        dna = ("EMULATED tiny DNA, nchains=%d" % max(4, nranks), st, tr, max(4, nranks), (20, 60))
        cod = None
    else:
        with open(os.path.join(GOLD, ("bench_c4" if full else "bench_c2") + ".json")) as fh:
            gold = json.load(fh)
        sy = gold["synthetic"]
        st = mbdata.synthetic_states(sy["ntaxa"], sy["nsites"], 4, sy["seed"], sy["p_mut"], sy["p_gap"])
        dna = ("configs[3]: DNA %d x %d GTR+G4, nchains=8, default move mix" % (sy["ntaxa"], sy["nsites"]), st, mbtree.parse_newick(gold["newick"]), 8,
               (200, 1200) if not full else (100, 400))
        with open(os.path.join(GOLD, "bench_c5.json")) as fh:
            g5 = json.load(fh)
        s5 = g5["synthetic"]
        cod = ("configs[4]: codon M3 %d x %d, nruns=2 x nchains=4, default move mix" % (s5["ntaxa"], s5["nsites"]),
               mbdata.synthetic_states(s5["ntaxa"], s5["nsites"], 61, s5["seed"], s5["p_mut"], s5["p_gap"]), mbtree.parse_newick(g5["newick"]), (100, 500))
    label, st, tr, nchains, (lo, hi) = dna
    r = rate(lambda ngen: refrun.mcmc_nexus(st, tr, ngen, beagle="dynamic", nchains=nchains), lo, hi)
    cases.append({"workload": label, "chains": nchains, "generations_per_s": r, "ngen": [lo, hi]})
    if cod is not None:
        label, st5, tr5, (lo, hi) = cod
        r = rate(lambda ngen: refrun.model_nexus("m3", st5, tr5, ngen=ngen, beagle="dynamic").replace("nchains=1 nruns=1", "nchains=4 nruns=2"), lo, hi)
        cases.append({"workload": label, "chains": 8, "generations_per_s": r, "ngen": [lo, hi]})
    return {"ranks": nranks, "launcher": "oracle/_ref/mbamd_mpirun -n %d oracle/_ref/%s" % (nranks, os.path.basename(binary)),
            "gpu_of_rank": "(instance + rank) mod visible GPUs (reference src/mbbeagle.c:201-207)",
            "devices_of_rank": {r: sorted(v) for r, v in sorted(devices.items())},
            "distinct_devices": len(set(x for v in devices.values() for x in v)), "cases": cases,
            "unit": "generations/s (all chains advance one generation)"}


def measure(args, cfg, steps, warmup, rank, local_rank, world, dist, device, emulate, lib, want_cpu_baseline, verbose=True):
    """One workload: build the division from its golden case, check the lnL against the reference's, time `steps`
    full-tree evaluations.  Returns the JSON object (rank 0) or None."""
    import torch
    from mrbayes_amd import likelihood as lk
    from mrbayes_amd.division import division_from_golden, synthetic_division

    case, kind, desc = CONFIGS[cfg]
    with open(os.path.join(GOLD, case + ".json")) as fh:
        gold = json.load(fh)
    if emulate:
        desc = "EMULATED (invalid as a measurement): " + desc
        div = synthetic_division(kind, 16, 200, seed=3, tree_seed=4, golden_dir=GOLD)
    else:
        div = division_from_golden(GOLD, case)
    if world > 1:                      # every chain has its own state: perturb the branch lengths per rank
        import random
        rng = random.Random(1000 + rank)
        div.tree.length = [l * (0.8 + 0.4 * rng.random()) for l in div.tree.length]
    bd = lk.BeagleDivision(div, lib, nchains=1, scaling=lk.MB_BEAGLE_SCALE_ALWAYS, resource=0 if emulate else local_rank)
    impl = bd.inst.details.implName.decode()
    inst_devices = [lib.pci_bus_id(r) for r in bd.inst.devices()]      # one per child engine (MBAMD_SHARD: one per GPU)
    lnl0 = bd.LogLike(0)
    ref_lnl = gold["lnL"]["fp64"]
    ablation = bool(os.environ.get("MBAMD_BENCH_NO_ASSERT"))      # timing experiments with deliberately wrong kernels (tools/exp_*.sh)
    pinned = world == 1 and not emulate and not ablation
    if pinned:                         # the workload IS the golden case: its lnL is the reference's, or nothing is timed
        assert abs(lnl0 - ref_lnl) <= REL_FP64 * abs(ref_lnl), (cfg, lnl0, ref_lnl)
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
        if rc != 0 and not ablation:
            raise RuntimeError("evaluation failed with code %d" % rc)
        if exchange is not None:       # per-generation swap attempt: the two ranks that own the chains exchange states (RCCL)
            exchange.swap_generation({rank: lnl})
        return lnl

    def fence():
        if dist is not None:
            dist.barrier()
        if not emulate:
            torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    bd.inst.kernel_timing(True)
    bd.inst.get_kernel_timing(reset=True)
    bd.inst.get_step_timing(reset=True)
    fence()
    t0 = time.perf_counter()
    lnl = None
    for i in range(steps):
        lnl = step(i)
    fence()
    dt = time.perf_counter() - t0
    kms, klaunches = bd.inst.get_kernel_timing(reset=True)
    span_ms, spans = bd.inst.get_step_timing(reset=True)
    bd.inst.kernel_timing(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert ablation or abs(lnl - lnl0) <= 1e-9 * abs(lnl0), (lnl, lnl0)
    bd.finalize()
    if rank != 0:
        return None

    ms_per_step = dt / steps * 1e3
    value = units_per_step * world * steps / dt / 1e6
    k_ms = kms / max(steps, 1)                # the partials kernels per evaluation, HIP events on the engine's stream
    # every kernel of a step (transition matrices, partials, integration) and the gaps between them, HIP events on the
    # engine's stream; the host's wait is not in it (mbamdGetStepTiming)
    all_ms = span_ms / spans if spans > 0 else ms_per_step
    abytes = algorithmic_bytes_per_eval(S, K, P, N)
    aflops = flops_per_eval(S, K, P, N)
    pmc = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh).get(cfg)
    except (OSError, ValueError):
        pass
    if emulate:
        pmc = None
    # What the hardware actually has to do per evaluation: HBM bytes (PMC) and matrix-core flops ISSUED (PMC; a compact
    # tip's factor is a table gather, not an MFMA).  Without a PMC pass of the workload: the write-once model (every
    # interior result and its exponents stored once, children read from LDS) and no matrix-core figure.
    if pmc:
        traffic = float(pmc["traffic_bytes"])
        issued = pmc.get("mfma_issued_gflop", 0.0) * 1e9
    else:
        traffic = float(P) * (N - 2) * (K * S * 4 + K)
        issued = 0.0
    t_hbm = traffic / (HBM_PEAK_GBS * 1e9)
    t_mfma = issued / (FP32_PEAK_TFLOPS * 1e12)
    t_all = all_ms * 1e-3
    if t_mfma > t_hbm:
        roof = {"bound": "mfma", "achieved": issued / t_all / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": traffic / t_all / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["traffic"] = traffic if pmc else None
    if verbose:
        roof["definition"] = ROOFLINE_DEFINITION
    # the least HBM traffic the path can have: every interior result and its exponents written once, children read from LDS,
    # tips as one byte per pattern (the write-once model); traffic_over_floor = what the counters saw over that
    floor = float(P) * ((N - 2) * (K * S * 4 + K) + N)
    roof["traffic_floor_bytes"] = floor
    roof["traffic_over_floor"] = (traffic / floor) if pmc else None
    if pmc:
        roof["traffic_source"] = pmc["source"]
    else:
        roof["traffic_model_bytes"] = traffic
    roof["hbm_GBs"] = traffic / t_all / 1e9
    roof["hbm_frac"] = t_hbm / t_all
    if issued > 0:
        roof["mfma_issued_tflops"] = issued / t_all / 1e12
        roof["mfma_issued_frac"] = t_mfma / t_all
        if S >= 40:
            # the matrix work of the contraction counted as what the fp32 matrix path issues for it (round 5's PMC count, a property
            # of the workload) against the fp32 matrix peak; the kernel itself issues it as bf16 pieces on the 16-bit pipe
            roof["mfma_issued_note"] = "fp32-equivalent matrix flops (v_mfma_f32_32x32x2_f32 count of the workload) / 157.3 TFLOP/s; issued as 6 x bf16 products, see arith"
            if pmc and pmc.get("mfma_bf16_busy_cycles"):
                roof["bf16_pipe_busy_frac"] = pmc["mfma_bf16_busy_cycles"] / 1024.0 / (t_all * 2.1e9)
    if pmc and pmc.get("partials_kernel_wave_cycles"):
        # PMC, same passes as the traffic: the share of the partials kernel's wave-cycles spent issuing or stalled at issue (the rest: s_waitcnt)
        wcy = pmc["partials_kernel_wave_cycles"]
        roof["partials_kernel_wave_cycles"] = wcy
        roof["issue_bound_frac"] = wcy["issuing"] + wcy["stalled_at_issue"]
    roof["kernel"] = impl
    roof["all_kernels_ms_per_step"] = all_ms
    roof["partials_kernel_ms_per_step"] = k_ms
    roof["partials_kernel_frac"] = max(t_hbm, t_mfma) / (k_ms * 1e-3) if k_ms > 0 else None
    roof["launches_per_step"] = klaunches / max(steps, 1)
    roof["algorithmic_bytes_per_step"] = abytes
    roof["flops_per_step"] = aflops
    out = {
        "metric": "site-pattern lnL (node-pattern conditional-likelihood) updates/sec",
        "value": value, "unit": "M updates/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": ("INVALID: ablation experiment (MBAMD_BENCH_NO_ASSERT)" if ablation else "synthetic") if not emulate else "INVALID: emulated control-flow test",
        "config": {"workload": desc, "golden_case": case, "states": S, "categories": K, "patterns": P, "taxa": N,
                   "chains": world, "parallelism": "chain-parallel (1 chain per GPU)" if world > 1 else "1 chain",
                   "units_per_step": units_per_step, "lnL": lnl,
                   "lnL_reference_fp64": ref_lnl if pinned else None,
                   "lnL_pinned": bool(pinned)},
        "instance_devices": inst_devices,
        "pattern_lnl_per_s": P * world * steps / dt,
        "full_tree_evals_per_s": world * steps / dt,
        "roofline": roof,
    }
    if world > 1:
        out["config"]["note"] = ("scaling harness: every rank replays full evaluations of its own (perturbed) chain state; the "
                                 "per-generation swap attempt is exchanged over RCCL but does not change what is replayed")
    if want_cpu_baseline and world == 1 and not emulate:
        try:
            sample = {"gtr": args.cpu_sample_patterns, "wag": args.cpu_sample_patterns // 4, "m3": args.cpu_sample_patterns // 16}[kind]
            gold["case"] = case
            out["cpu_baseline"] = cpu_baseline(kind, gold, sample)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            if verbose and not args.no_whole_cpu_baseline:
                # the headline workload also against the reference on the WHOLE alignment (not cache-resident on the host)
                wb = cpu_baseline(kind, gold, sample, whole=True)
                if wb:
                    out["cpu_baseline_whole_alignment"] = wb
                    out["speedup_vs_cpu_baseline_whole_alignment"] = value / wb["value"]
                    out["cpu_baseline_note"] = ("two timings of the same reference kernels on one core of a shared host: `cpu_baseline` on the first %d "
                                                "patterns (60 evaluations), `cpu_baseline_whole_alignment` on all of them (6 evaluations, the reference's "
                                                "O(P^2) pattern compression subtracted by the two-point window).  Neither working set fits a cache (64 bytes "
                                                "x patterns x nodes); they differ by the run-to-run spread of the host (+-10 %% between repeats of ONE "
                                                "sample: 131 / 164 in round 5's driver run, 144 / 148 in this round's).  Believed: the whole alignment -- it "
                                                "is the workload the GPU line is quoted on; the prefix is the bounded sample the contract asks for." % sample)
        except Exception as exc:          # the baseline is a report, never a reason to lose the bench line
            out["cpu_baseline"] = {"value": None, "unit": "M updates/s", "cores": 1, "kind": "port",
                                   "sample": "failed: %r" % (exc,)}
    return out


ARITH_NOTE = ("fp32 conditional likelihoods, fp32 accumulation.  From 40 states on the matrix-vector products run on v_mfma_f32_32x32x16_bf16 with every "
              "fp32 operand as the exact sum of three bf16 pieces (six products of order <= 4, K-block by K-block, smallest first: "
              "profiles/r06_bf16x3.txt); fewer states: v_mfma_f32_32x32x2_f32.  site_error_vs_fp64_engine: per-site log-likelihood "
              "against the double-precision engine (itself within 1e-11 of the reference's fp64 build)")


def site_error(cfg, lib):
    """Per-site log-likelihood of the single-precision engine of `lib` against its double-precision engine, on the workload."""
    import numpy as np
    from mrbayes_amd import likelihood as lk
    from mrbayes_amd.division import division_from_golden
    case = CONFIGS[cfg][0]
    div = division_from_golden(GOLD, case)
    vals = {}
    for dp in (True, False):
        bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_ALWAYS, double_precision=dp)
        lnl = bd.LogLike(0)
        site = np.array(bd.inst.get_site_log_likelihoods(), dtype=np.float64)
        bd.finalize()
        vals[dp] = (lnl, site)
    d = vals[False][1] - vals[True][1]
    return {"max": float(np.abs(d).max()), "rms": float(np.sqrt((d * d).mean())), "mean": float(d.mean()),
            "lnL_error": vals[False][0] - vals[True][0], "patterns": int(d.size)}


def arithmetic_report(args, cfg, o, device, lib):
    """The general-state workloads: how far the fp32 engine is from the fp64 engine per site -- for the product's kernels and, where the
    contraction runs on bf16 pieces (codon), for the same sources built with the fp32 MFMA chain (round 5's arithmetic, timed too)."""
    from mrbayes_amd import beagle as bg
    from mrbayes_amd import build as mbbuild
    o["arith"] = ARITH_NOTE
    o["site_error_vs_fp64_engine"] = site_error(cfg, lib)
    if o["config"]["states"] >= 40 and os.path.exists(mbbuild.FP32_CHAIN_LIB):
        alt = bg.BeagleLibrary(mbbuild.FP32_CHAIN_LIB)
        m = measure(args, cfg, max(args.steps, 200), args.warmup, 0, 0, 1, None, device, False, alt, False, verbose=False)
        o["fp32_chain_variant"] = {
            "what": "the same sources with the contraction on v_mfma_f32_32x32x2_f32 (mrbayes_amd/libhmsbeagle_fp32chain.so, -DMBAMD_WG_BF_MIN=999)",
            "ms_per_step": m["ms_per_step"], "all_kernels_ms_per_step": m["roofline"]["all_kernels_ms_per_step"],
            "partials_kernel_ms_per_step": m["roofline"]["partials_kernel_ms_per_step"], "value": m["value"],
            "site_error_vs_fp64_engine": site_error(cfg, alt)}


def summary_of(out):
    """The last object of the line, compact enough to survive a tail cut: per workload
    [value (M updates/s), ms_per_step, all_kernels_ms_per_step, partials_kernel_ms_per_step, roofline.frac, bound]."""
    def row(o):
        r = o.get("roofline", {})
        return [round(o["value"], 1), round(o["ms_per_step"], 4), round(r.get("all_kernels_ms_per_step") or 0.0, 4),
                round(r.get("partials_kernel_ms_per_step") or 0.0, 4), round(r.get("frac") or 0.0, 4), r.get("bound")]
    summ = {"columns": ["M_updates_per_s", "ms_per_step", "all_kernels_ms", "partials_kernel_ms", "roofline_frac", "bound"]}
    for o in [out] + [a for a in out.get("also", []) if "config" in a]:
        summ[o["config"]["golden_case"].replace("bench_", "")] = row(o)
    for d in out.get("double_precision", []):
        if "golden_case" in d:
            summ[d["golden_case"].replace("bench_", "") + "_f64"] = [round(d["value"], 1), round(d["ms_per_step"], 4)]
    if out.get("cpu_baseline", {}).get("value"):
        summ["cpu_M_updates_per_s"] = {"prefix_sample": round(out["cpu_baseline"]["value"], 1)}
        if out.get("cpu_baseline_whole_alignment", {}).get("value"):
            summ["cpu_M_updates_per_s"]["whole_alignment"] = round(out["cpu_baseline_whole_alignment"]["value"], 1)
    m = out.get("mcmc_gen_per_s") or {}
    gen = {}
    for mix in ("default_moves", "fixed_topology", "codon_m3_fixed_topology"):
        if isinstance(m.get(mix), dict):
            gen[mix] = {k: round(v, 1) for k, v in m[mix].items() if isinstance(v, float)}
    if gen:
        summ["mcmc_gen_per_s"] = gen
    return summ


def box_write_rate(device):
    """What THIS box sustains for a plain write stream (torch fill_ of 2 GiB, HIP events): the boxes of the pool differ by tens of
    per cent for HBM-bound kernels, and the tree walks are write streams.  GB/s, or None."""
    import torch
    try:
        a = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=device)
        for _ in range(3):
            a.fill_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            a.fill_(1.0)
        e1.record()
        torch.cuda.synchronize()
        rate = 4.0 * a.numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a
        torch.cuda.empty_cache()
        return rate
    except Exception:
        return None


def double_precision_line(cfg, steps, lib):
    """The same workload through the double-precision engine (BEAGLE_FLAG_PRECISION_DOUBLE, `set beagleprecision=double`): lnL
    against the reference's double build, wall time of replayed full-tree evaluations.  A report beside the fp32 line."""
    from mrbayes_amd import likelihood as lk
    from mrbayes_amd.division import division_from_golden
    case, kind, desc = CONFIGS[cfg]
    with open(os.path.join(GOLD, case + ".json")) as fh:
        gold = json.load(fh)
    div = division_from_golden(GOLD, case)
    bd = lk.BeagleDivision(div, lib, nchains=1, scaling=lk.MB_BEAGLE_SCALE_ALWAYS, double_precision=True)
    try:
        impl = bd.inst.details.implName.decode()
        lnl0 = bd.LogLike(0)
        ref = gold["lnL"]["fp64"]
        assert abs(lnl0 - ref) <= 2e-6 + 2e-9 * abs(ref), (cfg, lnl0, ref)
        bd.AcceptMove(0)
        evals = [lk.record_evaluation(bd), lk.record_evaluation(bd)]
        for i in range(5):
            evals[i & 1].run()
        t0 = time.perf_counter()
        for i in range(steps):
            rc, lnl = evals[i & 1].run()
            if rc != 0:
                raise RuntimeError("evaluation failed with code %d" % rc)
        dt = time.perf_counter() - t0
    finally:
        bd.finalize()
    return {"workload": desc, "golden_case": case, "dtype": "f64", "kernel": impl, "steps": steps, "ms_per_step": dt / steps * 1e3,
            "value": (div.ntaxa - 2) * div.npatterns * steps / dt / 1e6, "unit": "M updates/s", "lnL": lnl, "lnL_reference_fp64": ref,
            "abs_diff": abs(lnl - ref), "lnL_pinned": True}


def pattern_sharded(args, cfg, steps, warmup, rank, local_rank, world, dist, device, emulate, lib):
    """ONE chain whose site patterns are cut into `world` contiguous blocks, a block per rank / GPU (the
    north star's "site-pattern blocks sharded across the GPUs ... RCCL all-reduce of the per-generation lnL"): every rank
    evaluates the SAME tree state on its block, the block sums meet in one all-reduce(SUM) of a double per evaluation -- the
    only exchange.  Strong scaling: the units of a step are those of the whole alignment.  Collective: every rank calls it."""
    import torch
    from mrbayes_amd import likelihood as lk
    from mrbayes_amd.division import division_from_golden, synthetic_division

    case, kind, desc = CONFIGS[cfg]
    with open(os.path.join(GOLD, case + ".json")) as fh:
        gold = json.load(fh)
    div = synthetic_division(kind, 16, max(200, 96 * world), seed=3, tree_seed=4, golden_dir=GOLD) if emulate else division_from_golden(GOLD, case)
    P, N = div.npatterns, div.ntaxa
    blocks = (P + 63) // 64                              # whole 64-pattern blocks per rank, the remainder to the last ranks
    lo = min(P, (blocks * rank // world) * 64)
    hi = min(P, (blocks * (rank + 1) // world) * 64) if rank + 1 < world else P
    whole = None
    if emulate and rank == 0:                            # (no golden value for the emulated toy: evaluate it unsharded once)
        b0 = lk.BeagleDivision(div, lib, nchains=1, scaling=lk.MB_BEAGLE_SCALE_ALWAYS, resource=0)
        whole = b0.LogLike(0)
        b0.finalize()
    div.weights = div.weights[lo:hi]
    div.tip_states = [None if s is None else s[lo:hi].copy() for s in div.tip_states]
    div.tip_partials = [None if t is None else t[lo:hi].copy() for t in div.tip_partials]
    if div.inv_condlikes is not None:
        div.inv_condlikes = div.inv_condlikes[lo:hi].copy()
    # set-up is rank-local and may fail on one rank only (an empty block, no memory): the ranks agree on going ahead BEFORE the
    # first collective of the timed loop, or they all return the error -- never some of them inside an all-reduce the others skipped
    bd, evals, problem = None, None, None
    try:
        if hi <= lo:
            raise RuntimeError("rank %d of %d has no site patterns (%d patterns in 64-pattern blocks)" % (rank, world, P))
        bd = lk.BeagleDivision(div, lib, nchains=1, scaling=lk.MB_BEAGLE_SCALE_ALWAYS, resource=0 if emulate else local_rank)
        bd.LogLike(0)
        bd.AcceptMove(0)
        evals = [lk.record_evaluation(bd), lk.record_evaluation(bd)]
    except Exception as exc:
        problem = repr(exc)[:300]
    okflag = torch.tensor([0.0 if problem else 1.0], dtype=torch.float64, device=device)
    dist.all_reduce(okflag, op=dist.ReduceOp.MIN)
    if float(okflag.item()) == 0.0:
        if bd is not None:
            bd.finalize()
        return {"error": problem or "set-up failed on another rank"} if rank == 0 else None
    buf = torch.zeros(1, dtype=torch.float64, device=device)
    # The block's sum never visits the host on its own: the evaluation is left pending (mbamdSetDeferredResult), the engine adds its
    # block sums up on the device into `buf` and orders torch's stream behind that (mbamdReduceLogLikelihood), RCCL all-reduces `buf`,
    # and the ONE host read of a step is the chain's -- it needs the number before it can accept or reject.
    bd.inst.set_deferred_result(True)
    stream_handle = 0 if emulate else torch.cuda.current_stream(device).cuda_stream

    def step(i):
        rc, _ = evals[i & 1].run()
        if rc != 0:
            raise RuntimeError("evaluation failed with code %d" % rc)
        bd.inst.reduce_log_likelihood(buf.data_ptr(), stream_handle)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)       # the per-generation exchange: one double, device to device
        return float(buf.item())

    def fence():
        dist.barrier()
        if not emulate:
            torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    total = None
    for i in range(steps):
        total = step(i)
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    bd.inst.set_deferred_result(False)
    bd.finalize()
    if rank != 0:
        return None
    ref = whole if emulate else gold["lnL"]["fp64"]
    ok = abs(total - ref) <= REL_FP64 * abs(ref)
    if not ok:
        raise AssertionError("pattern-sharded lnL %r differs from the reference's %r" % (total, ref))
    return {"what": "ONE chain, site patterns in %d contiguous blocks (one per rank / GPU), all-reduce(SUM) of one double per evaluation over %s"
                    % (world, "gloo (EMULATED)" if emulate else "RCCL"),
            "scaling": "strong", "n_gpus": world, "steps": steps, "ms_per_step": dt / steps * 1e3,
            "host_synchronisations_per_step": 1,
            "value": (N - 2) * P * steps / dt / 1e6, "unit": "M updates/s",
            "lnL": total, "lnL_reference_fp64": ref, "lnL_pinned": True}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-patterns", type=int, default=8000)
    ap.add_argument("--no-whole-cpu-baseline", action="store_true",
                    help="skip the second CPU baseline of the headline workload: the reference on the whole alignment (about a minute at 1000 x 50000)")
    ap.add_argument("--no-also", action="store_true", help="skip the other three workloads (c2, c3, c5) at N=1")
    ap.add_argument("--no-mcmc", action="store_true", help="skip whole-MCMC generations/s of the unmodified MrBayes binary")
    ap.add_argument("--no-arith", action="store_true", help="skip the per-site error report of the general-state workloads (and the timing of the fp32-chain build beside the codon workload)")
    ap.add_argument("--mcmc", action="store_true", help="longer MCMC windows (adds minutes)")
    ap.add_argument("--shard", action="store_true",
                    help="ONE chain whose site patterns are sharded over the N GPUs inside one engine instance (strong scaling; "
                         "rank 0 drives all devices, the other ranks idle)")
    ap.add_argument("--no-mpi", action="store_true", help="skip the MPI-build MCMC measurement (mbamd_mpirun -n N mb_amd_mpi_pars)")
    ap.add_argument("--mpi-full", action="store_true", help="the MPI measurement's DNA analysis on the 1000 x 50000 alignment (minutes of start-up)")
    ap.add_argument("--secondary-timeout", type=float, default=600.0,
                    help="N > 1: seconds the secondary modes (pattern-sharded chain, MPI-build MCMC) may take together before every rank "
                         "leaves and rank 0 prints the line with the modes that finished (0: no limit)")
    ap.add_argument("--emulate", action="store_true",
                    help="TEST ONLY: run the control flow on the CPU (host-emulation engine, gloo, tiny workload); "
                         "the JSON line is marked invalid")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started plainly (`python bench.py --gpus N`) instead of under a launcher: without this the script would see a world of
        # one, measure ONE GPU and print n_gpus = 1 with exit code 0.  It launches itself -- one rank per GPU, same arguments --
        # and passes on the children's output and exit status.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # (either direction: a launcher with more ranks than --gpus, or -- the case nobody would notice -- fewer)
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: refusing to measure another job than the one asked for" % (world, args.gpus))

    import torch
    emulate = args.emulate
    if not emulate and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # local rank -> device: the launcher may show every rank all GPUs (the usual case: device = local rank) or one each
    # (HIP_VISIBLE_DEVICES per rank: device 0); either way the PCI bus ids gathered below must be distinct, or nothing is reported
    dev_index = 0 if emulate else local_rank % max(1, torch.cuda.device_count())
    device = torch.device("cpu") if emulate else torch.device("cuda", dev_index)
    if not emulate:
        torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        if emulate:
            dist_.init_process_group("gloo", timeout=datetime.timedelta(minutes=60))
        else:
            dist_.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=60))
        dist = dist_

    from mrbayes_amd import beagle as bg
    if emulate:
        from tests.hostemu import build_emu
        lib = bg.library(build_emu.build())
    else:
        lib = bg.library()

    if args.shard and world > 1:
        # one chain, site patterns over the N GPUs inside ONE instance (Instance::makeChildren, a child per device): the only
        # exchange is N doubles added on the host per evaluation.  Rank 0 drives every device; the other ranks wait.
        out = None
        if rank == 0:
            os.environ["MBAMD_SHARD"] = str(world)
            out = measure(args, args.config, args.steps, args.warmup, 0, 0, 1, None, device, emulate, lib, False)
            out["n_gpus"] = world
            out["scaling"] = "strong"
            out["config"]["chains"] = 1
            out["config"]["parallelism"] = "site-pattern shards of one chain over %d GPUs inside one instance (MBAMD_SHARD)" % world
            out["config"]["lnL_pinned"] = False
            if len(set(out["instance_devices"])) != world and not emulate:
                raise SystemExit("bench.py --shard: the %d shards of the instance sit on %s" % (world, out["instance_devices"]))
    else:
        out = measure(args, args.config, args.steps, args.warmup, rank, dev_index, world, dist, device, emulate, lib,
                      not args.no_cpu_baseline)
    if world > 1 and dist is not None:
        # which physical GPU every rank computes on (PCI bus id of its resource): a scaling line whose ranks share a device is
        # not a scaling line -- refuse it instead of reporting it
        mine = (("emu:rank%d" % rank) if emulate else lib.pci_bus_id(dev_index)).encode()[:63]
        ids = torch.zeros(world, 64, dtype=torch.uint8, device=device)
        ids[rank, :len(mine)] = torch.tensor(list(mine), dtype=torch.uint8, device=device)
        dist.all_reduce(ids, op=dist.ReduceOp.SUM)
        names = [bytes(int(x) for x in row if int(x) != 0).decode() for row in ids.cpu()]
        if len(set(names)) != world and not emulate:
            raise SystemExit("bench.py --gpus %d: ranks share a device (%s): refusing to report a scaling number" % (world, names))
        if rank == 0 and out is not None:
            out["ranks_devices"] = names
    # The primary measurement is done.  What follows at N > 1 (pattern-sharded chain, the MPI-build MCMC) are secondary modes nobody
    # can debug on an 8-GPU node after the fact: each is bounded, and if one hangs (a collective that never completes, a child
    # process that never returns) every rank leaves after --secondary-timeout seconds and rank 0 prints the line with what finished.
    watchdog = None
    if world > 1 and dist is not None and args.secondary_timeout > 0:
        import threading

        def bail():
            if rank == 0 and out is not None:
                out["timed_out"] = [m for m in ("pattern_sharded", "mpi_mcmc") if m not in out and not (m == "pattern_sharded" and args.shard) and not (m == "mpi_mcmc" and args.no_mpi)]
                out["summary"] = summary_of(out)
                print(json.dumps(out), flush=True)
            _rank_log(rank, "secondary modes exceeded %g s: leaving" % args.secondary_timeout)
            os._exit(0)
        watchdog = threading.Timer(args.secondary_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
    if world > 1 and not args.shard and dist is not None:
        try:                                     # (collective: every rank takes part; failures are reported, not fatal)
            ps = pattern_sharded(args, args.config, args.steps, args.warmup, rank, dev_index, world, dist, device, emulate, lib)
        except AssertionError:
            raise
        except Exception as exc:
            ps = {"error": repr(exc)[:600]}
        if rank == 0:
            out["pattern_sharded"] = ps
    if rank == 0 and world == 1 and not emulate:
        fill = box_write_rate(device)
        if fill:                                 # a box-normalised reading beside the fraction of the nominal 8 TB/s
            for o in [out]:
                o["roofline"]["box_write_stream_GBs"] = fill
                o["roofline"]["hbm_frac_of_box_write_stream"] = o["roofline"]["hbm_GBs"] / fill
        if not args.no_also:
            out["also"] = []
            for other in ("c2", "c3", "c5", "c4"):
                if other == args.config:
                    continue
                try:
                    out["also"].append(measure(args, other, max(args.steps, 200), args.warmup, 0, dev_index, 1, None, device,
                                               False, lib, not args.no_cpu_baseline, verbose=False))
                    if other in ("c3", "c5") and not args.no_arith:
                        try:
                            arithmetic_report(args, other, out["also"][-1], device, lib)
                        except Exception as exc:
                            out["also"][-1]["arith_error"] = repr(exc)[:300]
                    if fill:
                        out["also"][-1]["roofline"]["box_write_stream_GBs"] = fill
                        out["also"][-1]["roofline"]["hbm_frac_of_box_write_stream"] = out["also"][-1]["roofline"]["hbm_GBs"] / fill
                except Exception as exc:
                    out["also"].append({"workload": other, "error": repr(exc)})
        if args.config in ("c3", "c5") and not args.no_arith:
            try:
                arithmetic_report(args, args.config, out, device, lib)
            except Exception as exc:
                out["arith_error"] = repr(exc)[:300]
        if not args.no_also:
            out["double_precision"] = []
            for other in (args.config, "c5"):
                try:
                    out["double_precision"].append(double_precision_line(other, 50, lib))
                except Exception as exc:
                    out["double_precision"].append({"workload": other, "error": repr(exc)[:300]})
        if not args.no_mcmc:
            try:
                with open(os.path.join(GOLD, "bench_c2.json")) as fh:
                    out["mcmc_gen_per_s"] = mcmc_gen_per_s(json.load(fh), quick=not args.mcmc)
            except Exception as exc:
                out["mcmc_gen_per_s"] = {"error": repr(exc)}
    if rank == 0 and not args.no_mpi and not (world == 1 and args.no_mcmc):
        try:
            out["mpi_mcmc"] = mpi_mcmc(world, emulate=emulate, full=args.mpi_full)
        except Exception as exc:
            out["mpi_mcmc"] = {"error": repr(exc)[:600]}
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        out["summary"] = summary_of(out)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _rank_log(rank, text):
    """A line for this rank under gpurun_out/ (scratch; merged back by gpurun): what a failed multi-GPU run leaves behind."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_rank%d.log" % rank), "a") as fh:
            fh.write(text.rstrip() + "\n")
    except OSError:
        pass


if __name__ == "__main__":
    try:
        main()
    except BaseException as exc:                     # (SystemExit included: a refusal to report is worth a line per rank too)
        if not (isinstance(exc, SystemExit) and exc.code in (0, None)):
            import traceback
            _rank_log(int(os.environ.get("RANK", "0")), "rank %s failed:\n%s" % (os.environ.get("RANK", "0"), traceback.format_exc()))
        raise
