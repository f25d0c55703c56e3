"""Unrooted binary trees in the shape MrBayes hands to its likelihood calculators.

Mirrors the parts of the reference data contract the hot path consumes
(`TreeNode`/`Tree`, reference src/bayes.h:572-621): tips have index < ntaxa, interior nodes
ntaxa..2*ntaxa-3, `int_down_pass` is the post-order list of interior nodes
(`Tree.intDownPass`, filled by GetDownPass, src/utils.c:3909), and an *unrooted* tree is stored
with a tip as the calculation root whose single child `root_left` is the top interior node
(src/likelihood.c:7910-7918, src/mbbeagle.c:1231-1241).

Only what the likelihood path needs is here: no priors, no moves.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple


@dataclass
class Tree:
    ntaxa: int
    left: List[int]            # child index or -1 (tips)
    right: List[int]
    anc: List[int]             # parent index (root tip: -1)
    length: List[float]        # branch to anc; for root_left this is the root-tip branch
    root: int                  # the tip used as calculation root (MrBayes: t->root)
    root_left: int             # top interior node (t->root->left)
    int_down_pass: List[int] = field(default_factory=list)   # post-order, interior only
    all_down_pass: List[int] = field(default_factory=list)   # post-order, every node except root tip

    @property
    def n_nodes(self) -> int:           # 2N-2 for an unrooted tree (src/utils.c:2115)
        return 2 * self.ntaxa - 2

    @property
    def n_int_nodes(self) -> int:       # N-2
        return self.ntaxa - 2

    def is_tip(self, i: int) -> bool:
        return i < self.ntaxa

    def _fill_passes(self) -> None:
        order: List[int] = []
        stack: List[Tuple[int, bool]] = [(self.root_left, False)]
        while stack:
            node, done = stack.pop()
            if done or self.left[node] < 0:
                order.append(node)
                continue
            stack.append((node, True))
            stack.append((self.right[node], False))
            stack.append((self.left[node], False))
        self.all_down_pass = order
        self.int_down_pass = [n for n in order if self.left[n] >= 0]

    def levels(self) -> List[List[int]]:
        """Interior nodes grouped by dependency level (level 0 = both children are tips)."""
        lvl: Dict[int, int] = {}
        out: List[List[int]] = []
        for p in self.int_down_pass:
            l = 0
            for c in (self.left[p], self.right[p]):
                if c >= self.ntaxa:
                    l = max(l, lvl[c] + 1)
            lvl[p] = l
            while len(out) <= l:
                out.append([])
            out[l].append(p)
        return out

    def to_newick(self, names: Optional[Sequence[str]] = None, fmt: str = "%.15e") -> str:
        def lab(i: int) -> str:
            return names[i] if names is not None else str(i + 1)

        def rec(i: int) -> str:
            if self.left[i] < 0:
                return lab(i)
            a, b = self.left[i], self.right[i]
            return "(%s:%s,%s:%s)" % (rec(a), fmt % self.length[a], rec(b), fmt % self.length[b])

        p = self.root_left
        a, b = self.left[p], self.right[p]
        return "(%s:%s,%s:%s,%s:%s);" % (
            lab(self.root), fmt % self.length[p],
            rec(a), fmt % self.length[a], rec(b), fmt % self.length[b])


def _from_adjacency(ntaxa: int, adj: Dict[int, List[Tuple[int, float]]], root_tip: int) -> Tree:
    """Build the MrBayes-shaped tree from an undirected adjacency map (tips 0..ntaxa-1 have degree
    1, every other vertex degree 3).  Interior vertices are renumbered ntaxa.. in post-order."""
    n_nodes = 2 * ntaxa - 2
    left = [-1] * n_nodes
    right = [-1] * n_nodes
    anc = [-1] * n_nodes
    length = [0.0] * n_nodes
    new_id: Dict[int, int] = {}
    next_int = [ntaxa]

    (top, top_len), = adj[root_tip]
    # iterative DFS: (vertex, parent vertex, branch length to parent)
    result: Dict[int, int] = {}
    stack: List[Tuple[int, int, float, int]] = [(top, root_tip, top_len, 0)]
    kids: Dict[int, List[Tuple[int, float]]] = {}
    while stack:
        v, par, blen, state = stack.pop()
        if v < ntaxa:
            result[v] = v
            length[v] = blen
            continue
        if state == 0:
            ch = [(u, l) for (u, l) in adj[v] if u != par]
            if len(ch) != 2:
                raise ValueError("interior vertex %d has degree %d (need 3)" % (v, len(ch) + 1))
            kids[v] = ch
            stack.append((v, par, blen, 1))
            stack.append((ch[1][0], v, ch[1][1], 0))
            stack.append((ch[0][0], v, ch[0][1], 0))
        else:
            me = next_int[0]
            next_int[0] += 1
            result[v] = me
            a, b = result[kids[v][0][0]], result[kids[v][1][0]]
            left[me], right[me] = a, b
            anc[a] = me
            anc[b] = me
            length[me] = blen
    top_id = result[top]
    anc[top_id] = root_tip
    t = Tree(ntaxa, left, right, anc, length, root_tip, top_id)
    t._fill_passes()
    if len(t.int_down_pass) != ntaxa - 2:
        raise ValueError("tree is not a binary unrooted tree on %d taxa" % ntaxa)
    return t


def random_tree(ntaxa: int, seed: int, brlen: Optional[float] = 0.05,
                brlen_mean: float = 0.05, root_tip: int = 0) -> Tree:
    """Random pairwise joins of shuffled tips (the synthetic-input recipe of SURVEY §8(d)).
    brlen=None draws exponential(brlen_mean) lengths instead of a constant."""
    rng = random.Random(seed)
    adj: Dict[int, List[Tuple[int, float]]] = {i: [] for i in range(ntaxa)}
    live = list(range(ntaxa))
    nxt = ntaxa

    def bl() -> float:
        return brlen if brlen is not None else max(1e-6, rng.expovariate(1.0 / brlen_mean))

    while len(live) > 3:
        i = rng.randrange(len(live)); a = live.pop(i)
        j = rng.randrange(len(live)); b = live.pop(j)
        v = nxt; nxt += 1
        adj[v] = []
        for u in (a, b):
            l = bl()
            adj[v].append((u, l)); adj[u].append((v, l))
        live.append(v)
    v = nxt
    adj[v] = []
    for u in live:
        l = bl()
        adj[v].append((u, l)); adj[u].append((v, l))
    return _from_adjacency(ntaxa, adj, root_tip)


def caterpillar_tree(ntaxa: int, brlen: float = 0.05) -> Tree:
    """Maximally unbalanced tree (deepest dependency chain) -- an edge case for level scheduling."""
    adj: Dict[int, List[Tuple[int, float]]] = {i: [] for i in range(ntaxa)}
    nxt = ntaxa
    prev = None
    # chain of interior vertices v0..v(N-3); v0 holds tips 0,1; last holds tips N-2,N-1
    ints = list(range(ntaxa, 2 * ntaxa - 2))
    for v in ints:
        adj[v] = []
    def link(a, b):
        adj[a].append((b, brlen)); adj[b].append((a, brlen))
    link(ints[0], 0); link(ints[0], 1)
    for k in range(1, len(ints)):
        link(ints[k], ints[k - 1]); link(ints[k], k + 1)
    link(ints[-1], ntaxa - 1)
    return _from_adjacency(ntaxa, adj, 0)


def parse_newick(s: str, names: Optional[Sequence[str]] = None, root_tip: Optional[int] = None) -> Tree:
    """Parse an unrooted (top-level trifurcation) or rooted-binary Newick string.
    Labels are taxon names (looked up in `names`) or 1-based taxon numbers."""
    s = s.strip()
    if s.endswith(";"):
        s = s[:-1]
    name_to_idx = {n: i for i, n in enumerate(names)} if names is not None else None
    pos = [0]
    adj: Dict[int, List[Tuple[int, float]]] = {}
    nint = [0]
    tips: List[int] = []

    def parse_len() -> float:
        if pos[0] < len(s) and s[pos[0]] == ":":
            pos[0] += 1
            st = pos[0]
            while pos[0] < len(s) and s[pos[0]] not in ",()":
                pos[0] += 1
            return float(s[st:pos[0]])
        return 0.0

    def parse_node() -> Tuple[int, float]:
        if s[pos[0]] == "(":
            pos[0] += 1
            ch = [parse_node()]
            while s[pos[0]] == ",":
                pos[0] += 1
                ch.append(parse_node())
            assert s[pos[0]] == ")", "unbalanced newick"
            pos[0] += 1
            # optional interior label
            while pos[0] < len(s) and s[pos[0]] not in ":,()":
                pos[0] += 1
            vid = -1 - nint[0]          # temporary negative ids for interior vertices
            nint[0] += 1
            adj[vid] = []
            for (c, l) in ch:
                adj[vid].append((c, l)); adj[c].append((vid, l))
            return vid, parse_len()
        st = pos[0]
        while s[pos[0]] not in ":,()":
            pos[0] += 1
        label = s[st:pos[0]].strip()
        if name_to_idx is not None and label in name_to_idx:
            idx = name_to_idx[label]
        else:
            idx = int(label) - 1
        adj.setdefault(idx, [])
        tips.append(idx)
        return idx, parse_len()

    top, _ = parse_node()
    ntaxa = len(tips)
    # a rooted binary top (degree 2): splice it out
    if len(adj[top]) == 2:
        (a, la), (b, lb) = adj[top]
        adj[a] = [(u, l) for (u, l) in adj[a] if u != top] + [(b, la + lb)]
        adj[b] = [(u, l) for (u, l) in adj[b] if u != top] + [(a, la + lb)]
        del adj[top]
    # renumber interior temp ids to ntaxa..
    remap = {}
    k = ntaxa
    for v in list(adj.keys()):
        if v < 0:
            remap[v] = k; k += 1
    adj2: Dict[int, List[Tuple[int, float]]] = {}
    for v, lst in adj.items():
        adj2[remap.get(v, v)] = [(remap.get(u, u), l) for (u, l) in lst]
    if root_tip is None:
        root_tip = 0
    return _from_adjacency(ntaxa, adj2, root_tip)
