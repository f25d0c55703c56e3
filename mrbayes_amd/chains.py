"""Chain-parallel Metropolis coupling across GPUs: the N>1 path.

The reference's only parallelism is "chains -> MPI ranks" (src/mcmc.c:18331-18384): every rank evaluates
its own chains with its own likelihood calculator, and the ranks exchange a handful of doubles when a swap
between two heated chains is attempted (AttemptSwap, src/mcmc.c:591-1140: myStateInfo = lnL, lnPrior, ...).
Here one process drives one MI355X (one engine instance = the chains that live on that GPU) and the
exchange runs over torch.distributed -- backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
As in the reference, a swap attempt is a PAIRWISE exchange: every rank draws the same two chains and the same
acceptance number from the shared swap RNG (`swapseed`); only the (at most two) ranks that own those chains
talk -- one send/receive of (lnL, lnPrior, heat id) each way -- and everybody else goes straight on to the
next generation ("chains that do not swap, continue", src/mcmc.c:653-668).  `all_states` (one all-reduce of
the 2*nchains vector) is the collective used for reporting only.

Nothing here touches conditional likelihoods: chains never exchange partials.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def temperature(chain_id: int, nchains: int, chain_temp: float = 0.1, user_temps: Optional[Sequence[float]] = None) -> float:
    """Temperature (heat) of a chain, src/mcmc.c:18963-18977."""
    i = chain_id % nchains
    if user_temps is not None:
        return float(user_temps[i])
    return 1.0 / (1.0 + chain_temp * i)


def chains_of_rank(nchains_total: int, world: int, rank: int) -> List[int]:
    """Block assignment of chains to ranks, the reference's rule (src/mcmc.c:611-640): the first
    (total % world) ranks get one chain more."""
    base, extra = divmod(nchains_total, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


class SwapRng:
    """The reference's linear congruential generator (RandomNumber, src/utils.c:13802-13814); every rank
    seeds it with the same `swapseed`, so all ranks draw the same swap proposals."""

    def __init__(self, seed: int):
        self.seed = int(seed)

    def random(self) -> float:
        hi, lo = divmod(self.seed, 127773)
        test = 16807 * lo - 2836 * hi
        self.seed = test if test > 0 else test + 2147483647
        return self.seed / 2147483647.0


def swap_log_ratio(lnl_a, lnpr_a, temp_a, lnl_b, lnpr_b, temp_b) -> float:
    """lnR of exchanging the heats of chains A and B (no character reweighting), src/mcmc.c:719."""
    return (temp_b * (lnl_a + lnpr_a) + temp_a * (lnl_b + lnpr_b)) - (temp_a * (lnl_a + lnpr_a) + temp_b * (lnl_b + lnpr_b))


class ChainExchange:
    """Per-generation state exchange + swap decision for `nchains` chains spread over the process group."""

    def __init__(self, nchains: int, dist=None, device=None, chain_temp: float = 0.1, swap_seed: int = 12345):
        self.nchains = nchains
        self.dist = dist
        self.device = device
        self.chain_temp = chain_temp
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.local = chains_of_rank(nchains, self.world, self.rank)
        self.chain_id = list(range(nchains))         # chainId[]: which heat each chain currently runs at
        self.rng = SwapRng(swap_seed)
        self.swaps_tried = 0
        self.swaps_done = 0
        self._swaps_led = 0                          # accepted swaps in which this rank owned chain a (counted once globally)
        self._buf = None
        self.pairwise = True                         # False: swap attempts use the all-reduce of all chains' states

    def all_states(self, lnl: Dict[int, float], lnprior: Optional[Dict[int, float]] = None) -> Tuple[np.ndarray, np.ndarray]:
        """Every rank contributes the (lnL, lnPrior) of its own chains; everybody gets all of them."""
        import torch
        if self._buf is None:
            self._buf = torch.zeros(2 * self.nchains, dtype=torch.float64, device=self.device or "cpu")
        v = np.zeros(2 * self.nchains)
        for c in self.local:
            v[c] = lnl[c]
            v[self.nchains + c] = (lnprior or {}).get(c, 0.0)
        self._buf.copy_(torch.from_numpy(v))
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(self._buf)            # sum of disjoint contributions
        out = self._buf.cpu().numpy()
        return out[: self.nchains].copy(), out[self.nchains:].copy()

    def attempt_swap(self, lnl: np.ndarray, lnprior: np.ndarray) -> Tuple[int, int, bool]:
        """One swap attempt between two chains picked with the shared RNG (RunChain, src/mcmc.c:16941-16957,
        then AttemptSwap's acceptance rule).  Identical on every rank."""
        a = int(self.rng.random() * self.nchains)
        b = int(self.rng.random() * (self.nchains - 1))
        if b >= a:
            b += 1
        ta = temperature(self.chain_id[a], self.nchains, self.chain_temp)
        tb = temperature(self.chain_id[b], self.nchains, self.chain_temp)
        lnr = swap_log_ratio(lnl[a], lnprior[a], ta, lnl[b], lnprior[b], tb)
        r = 0.0 if lnr < -100.0 else 1.0 if lnr > 0.0 else math.exp(lnr)
        ok = self.rng.random() < r
        self.swaps_tried += 1
        if ok:
            self.chain_id[a], self.chain_id[b] = self.chain_id[b], self.chain_id[a]
            self.swaps_done += 1
        return a, b, ok

    # ---- the reference's pattern: only the ranks that own the two chains communicate ------------------
    def rank_of_chain(self, chain: int) -> int:
        base, extra = divmod(self.nchains, self.world)
        cut = extra * (base + 1)
        return chain // (base + 1) if chain < cut else extra + (chain - cut) // max(base, 1)

    def _pair_exchange(self, peer: int, mine: Sequence[float]) -> List[float]:
        import torch
        send = torch.tensor(list(mine), dtype=torch.float64, device=self.device or "cpu")
        recv = torch.empty_like(send)
        ops = [self.dist.P2POp(self.dist.isend, send, peer), self.dist.P2POp(self.dist.irecv, recv, peer)]
        if self.rank > peer:
            ops.reverse()
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()
        return recv.cpu().tolist()

    def warm_up(self) -> None:
        """Open every pairwise connection once (outside any timed region)."""
        if self.dist is None or self.world < 2:
            return
        # (with the NCCL backend the first operation on a group has to involve all of its ranks; batched
        #  send/receive between two of them is allowed from then on)
        import torch
        first = torch.zeros(1, dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(first)
        # Capability is decided BEFORE anybody blocks in a pairwise exchange (a rank that raised while its peer waits in
        # req.wait() would leave the agreement all-reduce hanging): both backends this engine runs on have batched
        # point-to-point operations; anything else takes the collective path on every rank.
        ok = 1.0 if (hasattr(self.dist, "batch_isend_irecv") and self.dist.get_backend() in ("nccl", "gloo")) else 0.0
        agree = torch.tensor([ok], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(agree, op=self.dist.ReduceOp.MIN)
        self.pairwise = bool(agree.item() > 0.5)
        if not self.pairwise:
            return
        for i in range(self.world):
            for j in range(i + 1, self.world):
                if self.rank in (i, j):
                    self._pair_exchange(j if self.rank == i else i, [0.0, 0.0, 0.0])

    def swap_generation(self, lnl: Dict[int, float], lnprior: Optional[Dict[int, float]] = None):
        """One swap attempt of this generation (RunChain picks the pair, src/mcmc.c:16941-16957; AttemptSwap,
        src/mcmc.c:591-1140).  `lnl` / `lnprior`: this rank's own chains.  Returns (a, b, accepted) with
        accepted = None on a rank that owns neither chain (it does not communicate and does not learn the outcome,
        exactly like a reference rank; `chain_id` is authoritative for a rank's own chains only)."""
        if not self.pairwise:                               # collective fallback: every rank learns every state
            all_lnl, all_pr = self.all_states(lnl, lnprior)
            return self.attempt_swap(all_lnl, all_pr)
        a = int(self.rng.random() * self.nchains)
        b = int(self.rng.random() * (self.nchains - 1))
        if b >= a:
            b += 1
        u = self.rng.random()                               # drawn by everybody: the streams stay in step
        self.swaps_tried += 1
        ra, rb = self.rank_of_chain(a), self.rank_of_chain(b)
        if self.rank not in (ra, rb):
            return a, b, None
        pr = lnprior or {}
        state = {}
        for c in (a, b):
            if self.rank_of_chain(c) == self.rank:
                state[c] = (float(lnl[c]), float(pr.get(c, 0.0)), float(self.chain_id[c]))
        if ra != rb:
            mine, other = (a, b) if self.rank == ra else (b, a)
            state[other] = tuple(self._pair_exchange(rb if self.rank == ra else ra, state[mine]))
        ta = temperature(int(state[a][2]), self.nchains, self.chain_temp)
        tb = temperature(int(state[b][2]), self.nchains, self.chain_temp)
        lnr = swap_log_ratio(state[a][0], state[a][1], ta, state[b][0], state[b][1], tb)
        r = 0.0 if lnr < -100.0 else 1.0 if lnr > 0.0 else math.exp(lnr)
        ok = u < r
        if ok:
            ida, idb = int(state[a][2]), int(state[b][2])
            self.chain_id[a], self.chain_id[b] = idb, ida   # (entries of chains this rank does not own are not used)
            self.swaps_done += 1
            if self.rank == ra:
                self._swaps_led += 1
        return a, b, ok

    def cold_chain(self) -> int:
        """The chain that currently runs at heat 0.  In pairwise mode a rank only learns the outcome of swaps it takes part
        in (like a reference rank), so `chain_id` is authoritative for the rank's OWN chains only: the global answer needs
        everybody -- one all-reduce of "the cold chain among mine" (reporting path, not the per-generation path)."""
        if self.dist is None or self.world < 2 or not self.pairwise:
            return self.chain_id.index(0)
        import torch
        mine = [c for c in self.local if self.chain_id[c] == 0]
        v = torch.tensor([float(mine[0]) if mine else -1.0], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(v, op=self.dist.ReduceOp.MAX)
        cold = int(v.item())
        if cold < 0:
            raise RuntimeError("no rank owns the cold chain: chain_id tables are inconsistent")
        return cold

    def swap_counts(self) -> Tuple[int, int]:
        """(attempted, accepted) over the whole run: `swaps_tried` counts on every rank, `swaps_done` only on the ranks that
        took part -- the global number of accepted swaps is the all-reduced count of the lower-ranked participant."""
        if self.dist is None or self.world < 2 or not self.pairwise:
            return self.swaps_tried, self.swaps_done
        import torch
        v = torch.tensor([float(self._swaps_led)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(v)
        return self.swaps_tried, int(v.item())
