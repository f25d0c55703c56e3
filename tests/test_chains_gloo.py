"""The N>1 path on CPU: world_size-2 `gloo` process group, each rank driving its own engine instance (the
TEST-ONLY host-emulation build stands in for the GPU here) for its own heated chains; the per-generation
exchange and the swap decisions must be identical on every rank and equal to a single-process run."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mrbayes_amd import beagle as bg
    from mrbayes_amd import chains as ch
    from mrbayes_amd import likelihood as lk
    from mrbayes_amd.division import synthetic_division
    from tests.hostemu import build_emu
    lib = bg.library(build_emu.build())
    nchains = 4
    ex = ch.ChainExchange(nchains, dist=dist, chain_temp=0.1, swap_seed=777)
    trace = []
    engines = {}
    for c in ex.local:                      # every chain has its own state (branch lengths differ)
        div = synthetic_division("gtr", 24, 150, seed=5, tree_seed=6)
        div.tree.length = [l * (1.0 + 0.1 * c) for l in div.tree.length]
        engines[c] = lk.BeagleDivision(div, lib)
    ex.warm_up()
    for gen in range(12):
        lnl = {}
        for c, bd in engines.items():
            bd.TouchAllTreeNodes(0)
            lnl[c] = bd.LogLike(0)
            bd.AcceptMove(0)
        all_lnl, _ = ex.all_states(lnl)               # (reporting collective; the swap itself is pairwise)
        a, b, ok = ex.swap_generation(lnl)
        trace.append((all_lnl.tolist(), a, b, ok, {c: ex.chain_id[c] for c in ex.local}))
    for bd in engines.values():
        bd.finalize()
    q.put((rank, trace))
    dist.barrier()
    dist.destroy_process_group()


def test_chain_assignment():
    sys.path.insert(0, ROOT)
    from mrbayes_amd import chains as ch
    assert ch.chains_of_rank(8, 8, 3) == [3]
    assert ch.chains_of_rank(8, 2, 1) == [4, 5, 6, 7]
    assert ch.chains_of_rank(5, 2, 0) == [0, 1, 2] and ch.chains_of_rank(5, 2, 1) == [3, 4]
    assert sorted(sum((ch.chains_of_rank(7, 4, r) for r in range(4)), [])) == list(range(7))
    assert ch.temperature(0, 4) == 1.0 and abs(ch.temperature(2, 4, 0.1) - 1 / 1.2) < 1e-15
    # exchanging identical states is always accepted; the rule is antisymmetric in (A, B)
    assert ch.swap_log_ratio(-10.0, -1.0, 1.0, -10.0, -1.0, 0.5) == 0.0
    assert ch.swap_log_ratio(-10.0, 0.0, 1.0, -12.0, 0.0, 0.5) < 0.0 < ch.swap_log_ratio(-12.0, 0.0, 1.0, -10.0, 0.0, 0.5)


def test_collective_fallback_makes_the_same_decisions():
    """ChainExchange.pairwise = False (set by warm_up when the backend cannot do subset send/recv) routes swap
    attempts through the all-reduce of every chain's state: same draws, same outcomes."""
    sys.path.insert(0, ROOT)
    from mrbayes_amd import chains as ch
    lnl = {0: -1000.0, 1: -1003.5, 2: -998.2, 3: -1010.0}
    p2p, coll = ch.ChainExchange(4, swap_seed=99), ch.ChainExchange(4, swap_seed=99)
    coll.pairwise = False
    for gen in range(50):
        cur = {c: v - 0.1 * gen * (c + 1) for c, v in lnl.items()}
        assert p2p.swap_generation(cur) == coll.swap_generation(cur)
        assert p2p.chain_id == coll.chain_id
    assert p2p.swaps_done == coll.swaps_done > 0


def test_two_ranks_agree_with_one():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = 29650 + world
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        results[world] = dict(got)
    single = results[1][0]
    involved = 0
    for r in (0, 1):
        for gen, (lnl, a, b, ok, ids) in enumerate(results[2][r]):
            s_lnl, s_a, s_b, s_ok, s_ids = single[gen]
            assert lnl == s_lnl and (a, b) == (s_a, s_b), "rank %d drew a different pair in generation %d" % (r, gen)
            if ok is not None:                    # this rank owns one of the two chains: it took part and knows the outcome
                assert ok == s_ok
                involved += 1
            for c, heat in ids.items():           # a rank's own chains always carry the heats of the single-process run
                assert heat == s_ids[c], (r, gen, c)
    assert involved >= 6                          # every generation involves at least one of the two ranks
    lnl0 = np.array(single[0][0])
    assert np.all(np.isfinite(lnl0)) and len(set(lnl0.tolist())) == 4
