// mbamd_walk4_host.h -- host side of the 4-state tree-walk kernel: compiles one hazard-free segment of a
// beagleUpdatePartials operation list into per-wave PROGRAMS (Walk4Entry, mbamd_walk4.h).
//
// What the reference does node by node on the host (the post-order loop of LaunchLogLikeForDivision,
// src/likelihood.c:7851-7972, with FlipCondLikeSpace / CondLikeDown / CondLikeScaler per node) becomes here:
//   1. cut the operation forest into subtrees and pack them into W bins (one per wave); the ancestors of the cut
//      ("cap") form the next, much smaller phase -- repeated until nothing is left.  Phases are separated by a
//      workgroup barrier; inside a phase the waves never talk to each other;
//   2. order each subtree by Sethi-Ullman numbers (the child that needs more live values first), so that a wave
//      keeps about log2(subtree) results alive instead of the ~23 MrBayes' left-first post-order needs on 500 taxa;
//   3. linear-scan allocation of each wave's private LDS slots: a result consumed later by the same wave stays in a
//      slot; children that live in HBM (buffers of earlier launches, results of other waves, evicted values) are
//      prefetched into a slot by LDS-DMA a few entries ahead of their consumer; Belady eviction when slots run out;
//   4. exact s_waitcnt counts: the host replays the vector-memory instruction sequence of the kernel loop and
//      stores in every entry how many younger instructions may still be in flight when its inputs must have landed.
// The result depends only on the STRUCTURE of the list (who produces whose child, which children are tips), not on
// the buffer / matrix / scale indices, which change with every accept / reject flip: the structural part is cached
// (Walk4Template) and a list seen before is only re-filled.
//
// Pure host code without HIP dependencies (unit-tested on the CPU through the host-emulation build).
#ifndef MBAMD_WALK4_HOST_H_
#define MBAMD_WALK4_HOST_H_

#include <algorithm>
#include <functional>
#include <cstdint>
#include <cstring>
#include <queue>
#include <utility>
#include <vector>

namespace mbamd {

// one operation of the list, buffer indices only
struct Walk4Op {
    int dst, c1, c2;           // partials buffers (c1/c2: tip-state buffer when the tip flag is set)
    int m1, m2;                // transition matrices
    int scaleWrite, scaleRead; // exponent buffers or -1
    uint8_t tip1, tip2;
};

struct Walk4Template {
    struct Entry {
        int op = -1;                               // -1: NOP or PF entry
        uint8_t flags = 0;                         // NOP / BARRIER
        uint8_t c1slot = 0xFF, c2slot = 0xFF, dslot = 0xFF;   // child in an LDS slot (not a tip) / result kept in a slot
        uint8_t vmwait = 0xFF;                     // 0xFF: no wait
        bool reread = false;                       // (memSlots = false) the result leaves this wave's slots and comes back from HBM later in the same phase
        int pfOp[2] = {-1, -1};                    // PF entry: prefetch child pfChild of operation pfOp ...
        uint8_t pfChild[2] = {0, 0}, pfSlot[2] = {0, 0};      // ... into this slot
    };
    std::vector<int> key;
    int W = 1, entries = 0, nslots = 1;            // entries per wave (incl. the trailing NOPs)
    int tail = 2;                                  // trailing NOPs
    std::vector<Entry> prog;                       // [W][entries]
    int phases = 1, reloads = 0, externals = 0;
    int evictions = 0;                             // results evicted from a wave's slots and re-read by the same wave
};

struct Walk4Scratch {
    std::vector<int> prod1, prod2, parent, ncons, need, size, phaseOf, waveOf, posOf, order, stack, heapTmp;
    std::vector<char> assigned, cap;
    // kept between builds so that compiling a short list (a root-ward path: every MCMC generation) allocates nothing
    std::vector<int> binLoad, roots, frontier, slotHolder, lastUse, slotOfVal, freeFrom, memAt, nextOp, pieces, load, dryNodes;
    std::vector<char> fwd;
    std::vector<std::vector<int>> phaseStart;
    std::vector<Walk4Template::Entry> scan;
    std::vector<std::vector<Walk4Template::Entry>> fin;
};

class Walk4Builder {
public:
    int maxW = 1;              // waves per workgroup the launch geometry allows
    int maxSlots = 8;          // LDS slots per wave
    int maxSlots1 = 8;         // ... of a single-wave program (root-ward paths: every spare slot is a prefetch in flight)
    int prefetchDistance = 6;  // entries a child prefetch is issued ahead of its consumer (if a slot is free)
    bool safeWaits = false;    // debug: every wait is vmcnt(0)
    int smallPhase = 16;       // a phase with at most this many operations runs on one wave
    // general-state tree walk (k_walkg, mbamd_walkg.h): memSlots = false -- children that live in HBM are loaded straight
    // into registers by the kernel's operand pipeline (no slot, no prefetch entry; c?slot = 0xFF marks them); the loads of
    // entry j are issued during entry j-1, so a value this wave re-reads from HBM must have been stored by entry j-2 or earlier
    bool memSlots = true;
    bool alwaysKeep = false;
    bool phasesAreLaunches = false;  // every phase is its own kernel launch: nothing stays in LDS across a phase boundary
    // program frame: leading NOP entries, loop unroll factor of the kernel, trailing (read-ahead) NOP entries
    int leadNops = 0, unroll = 2, tailNops = 2;
    // 4-state walk: a result whose only consumer is the NEXT operation of the same wave -- in a post-order walk every parent
    // follows its last interior child directly -- stays in registers (c?slot = 0xFE): no LDS slot, no write / read-back round trip
    bool forward = false;

    // ops: one hazard-free segment (no buffer is written twice, none is written after it was read, a buffer read
    // after it was written is a dependency).  Fills `t` (structure) -- the caller turns it into Walk4Entry words.
    bool build(const std::vector<Walk4Op>& ops, Walk4Template& t);

private:
    Walk4Scratch s;
    void suOrder(int root, std::vector<int>& out);
};

// Sethi-Ullman post-order of the subtree below `root` restricted to operations of the current phase
// (s.phaseOf[o] == s.phaseOf[root] and same wave marker not yet needed): children first, the one that needs more
// live values first.
inline void Walk4Builder::suOrder(int root, std::vector<int>& out)
{
    std::vector<int>& st = s.stack;
    st.clear();
    st.push_back(root << 1);
    const int ph = s.phaseOf[root];
    while (!st.empty()) {
        const int top = st.back();
        st.pop_back();
        const int o = top >> 1;
        if (top & 1) { out.push_back(o); continue; }
        st.push_back((o << 1) | 1);
        int a = s.prod1[o], b = (s.prod2[o] != s.prod1[o]) ? s.prod2[o] : -1;
        if (a >= 0 && (s.phaseOf[a] != ph || s.parent[a] != o)) a = -1;
        if (b >= 0 && (s.phaseOf[b] != ph || s.parent[b] != o)) b = -1;
        if (a >= 0 && b >= 0 && s.need[b] > s.need[a]) std::swap(a, b);
        if (b >= 0) st.push_back(b << 1);          // pushed first = visited second
        if (a >= 0) st.push_back(a << 1);
    }
}

inline bool Walk4Builder::build(const std::vector<Walk4Op>& ops, Walk4Template& t)
{
    const int n = (int) ops.size();
    if (n == 0) return false;
    // ---- dependencies inside the segment -----------------------------------------------------------------
    std::vector<int>&prod1 = s.prod1, &prod2 = s.prod2, &parent = s.parent, &ncons = s.ncons;
    prod1.assign(n, -1); prod2.assign(n, -1); parent.assign(n, -1); ncons.assign(n, 0);
    {
        int maxBuf = 0;
        for (const Walk4Op& o : ops) maxBuf = std::max(maxBuf, std::max(o.dst, std::max(o.tip1 ? 0 : o.c1, o.tip2 ? 0 : o.c2)));
        std::vector<int>& writer = s.heapTmp;
        writer.assign((size_t) maxBuf + 1, -1);
        for (int o = 0; o < n; ++o) {
            if (!ops[o].tip1) prod1[o] = writer[ops[o].c1];
            if (!ops[o].tip2) prod2[o] = writer[ops[o].c2];
            writer[ops[o].dst] = o;
        }
    }
    bool dag = false;                              // a value with two consumers: no tree parallelism (single wave)
    for (int o = 0; o < n; ++o) {
        const int a = prod1[o], b = (prod2[o] != prod1[o]) ? prod2[o] : -1;
        if (a >= 0) { if (ncons[a]++ > 0) dag = true; else parent[a] = o; }
        if (b >= 0) { if (ncons[b]++ > 0) dag = true; else parent[b] = o; }
    }
    int W = dag ? 1 : std::max(1, std::min(maxW, 8));
    if (n < 2 * smallPhase) W = 1;

    // ---- phases and bins -----------------------------------------------------------------------------------
    std::vector<int>&phaseOf = s.phaseOf, &waveOf = s.waveOf, &need = s.need, &size = s.size;
    phaseOf.assign(n, -1); waveOf.assign(n, 0); need.assign(n, 1); size.assign(n, 1);
    std::vector<char>& cap = s.cap;
    int nphases = 0;
    std::vector<std::vector<std::vector<int>>> seq;            // [phase][wave] -> operations in execution order
    int remaining = n;
    std::vector<int>&binLoad = s.binLoad, &roots = s.roots, &frontier = s.frontier;
    while (remaining > 0) {
        const int ph = nphases++;
        seq.emplace_back(std::vector<std::vector<int>>(W));
        // subtree sizes and Sethi-Ullman numbers over the unassigned operations (children precede parents in the list)
        for (int o = 0; o < n; ++o) {
            if (phaseOf[o] >= 0) continue;
            int a = prod1[o], b = (prod2[o] != prod1[o]) ? prod2[o] : -1;
            if (a >= 0 && (phaseOf[a] >= 0 || parent[a] != o)) a = -1;
            if (b >= 0 && (phaseOf[b] >= 0 || parent[b] != o)) b = -1;
            const int na = a >= 0 ? need[a] : 0, nb = b >= 0 ? need[b] : 0;
            need[o] = std::max(1, (na == nb) ? na + (na > 0 ? 1 : 0) : std::max(na, nb));
            size[o] = 1 + (a >= 0 ? size[a] : 0) + (b >= 0 ? size[b] : 0);
        }
        roots.clear();
        for (int o = 0; o < n; ++o)
            if (phaseOf[o] < 0 && (parent[o] < 0 || dag)) roots.push_back(o);
        if (dag) {                                 // list order is a valid order
            for (int o = 0; o < n; ++o) { phaseOf[o] = ph; seq[ph][0].push_back(o); }
            remaining = 0;
            break;
        }
        if (W == 1 || remaining <= smallPhase) {
            // everything that is left, on one wave: the one that produced most of the values these operations read
            int w = 0;
            if (W > 1) {
                binLoad.assign(W, 0);
                for (int o = 0; o < n; ++o) {
                    if (phaseOf[o] >= 0) continue;
                    if (prod1[o] >= 0 && phaseOf[prod1[o]] >= 0) binLoad[waveOf[prod1[o]]]++;
                    if (prod2[o] >= 0 && phaseOf[prod2[o]] >= 0) binLoad[waveOf[prod2[o]]]++;
                }
                w = (int) (std::max_element(binLoad.begin(), binLoad.end()) - binLoad.begin());
            }
            for (int r : roots) phaseOf[r] = ph;   // (suOrder follows phaseOf: mark the whole rest first)
            for (int o = 0; o < n; ++o) if (phaseOf[o] < 0) phaseOf[o] = ph;
            std::sort(roots.begin(), roots.end(), [&](int x, int y) { return size[x] > size[y]; });
            for (int r : roots) suOrder(r, seq[ph][w]);
            for (int o : seq[ph][w]) waveOf[o] = w;
            remaining = 0;
            break;
        }
        // split the largest subtrees until the pieces pack well into W bins; the split nodes form the cap
        cap.assign(n, 0);
        auto cmp = [&](int x, int y) { return size[x] < size[y]; };
        std::priority_queue<int, std::vector<int>, decltype(cmp)> heap(cmp);
        for (int r : roots) heap.push(r);
        const int target = (remaining + W - 1) / W;
        int splits = 0, capSize = 0;
        // Every split moves one node into the cap -- the next, partly serial phase -- and makes the pieces pack better.
        // Dry run: the cost (fullest bin + cap) after every split; then split exactly as often as the cheapest point says.
        auto fullestBin = [&](const std::vector<int>& nodes) {      // (scratch vectors: no allocation per call)
            std::vector<int>&pieces = s.pieces, &load = s.load;
            pieces.clear();
            for (int v : nodes) pieces.push_back(size[v]);
            std::sort(pieces.begin(), pieces.end(), std::greater<int>());
            load.assign(W, 0);
            for (int sz : pieces) *std::min_element(load.begin(), load.end()) += sz;
            return *std::max_element(load.begin(), load.end());
        };
        int bestSplits = 0;
        {
            auto dry = heap;
            std::vector<int>& nodes = s.dryNodes;
            nodes.assign(roots.begin(), roots.end());                 // (the heap holds exactly the roots here)
            long bestCost = (long) fullestBin(nodes) + 0;
            if ((int) nodes.size() < W) bestCost = 1L << 40;          // (fewer pieces than bins: keep splitting)
            int k = 0, capNow = 0;
            while (!dry.empty() && k < 16 * W) {
                const int v = dry.top();
                if (size[v] <= std::max(target / 3, 4)) break;
                dry.pop();
                ++k; ++capNow;
                nodes.erase(std::find(nodes.begin(), nodes.end(), v));
                int a = prod1[v], b = (prod2[v] != prod1[v]) ? prod2[v] : -1;
                if (a >= 0 && phaseOf[a] < 0 && parent[a] == v) { dry.push(a); nodes.push_back(a); }
                if (b >= 0 && phaseOf[b] < 0 && parent[b] == v) { dry.push(b); nodes.push_back(b); }
                if ((int) nodes.size() < W) continue;
                // the cap runs on one wave (the others wait): its operations count fully; + the barrier entry
                const long cost = (long) fullestBin(nodes) + capNow + 1;
                if (cost < bestCost) { bestCost = cost; bestSplits = k; }
            }
        }
        while (!heap.empty() && splits < bestSplits) {
            const int v = heap.top();
            heap.pop();
            cap[v] = 1; ++capSize; ++splits;
            int a = prod1[v], b = (prod2[v] != prod1[v]) ? prod2[v] : -1;
            if (a >= 0 && phaseOf[a] < 0 && parent[a] == v) heap.push(a);
            if (b >= 0 && phaseOf[b] < 0 && parent[b] == v) heap.push(b);
        }
        frontier.clear();
        while (!heap.empty()) { frontier.push_back(heap.top()); heap.pop(); }    // largest first
        if (frontier.empty()) {                    // (cannot happen: a split node always leaves a child or ends the loop)
            for (int o = 0; o < n; ++o) if (phaseOf[o] < 0) { phaseOf[o] = ph; seq[ph][0].push_back(o); }
            remaining = 0;
            break;
        }
        binLoad.assign(W, 0);
        // mark the phase of every operation below the frontier first (suOrder needs it), then order bin by bin
        std::vector<int>& st = s.stack;
        for (int r : frontier) {
            const int w = (int) (std::min_element(binLoad.begin(), binLoad.end()) - binLoad.begin());
            binLoad[w] += size[r];
            st.clear();
            st.push_back(r);
            while (!st.empty()) {
                const int o = st.back();
                st.pop_back();
                phaseOf[o] = ph;
                waveOf[o] = w;
                int a = prod1[o], b = (prod2[o] != prod1[o]) ? prod2[o] : -1;
                if (a >= 0 && phaseOf[a] < 0 && parent[a] == o) st.push_back(a);
                if (b >= 0 && phaseOf[b] < 0 && parent[b] == o) st.push_back(b);
            }
        }
        for (int r : frontier) suOrder(r, seq[ph][waveOf[r]]);
        for (int w = 0; w < W; ++w) remaining -= (int) seq[ph][w].size();
        (void) capSize;
    }

    // ---- per-wave item sequences: phase p >= 1 opens with a NOP|BARRIER entry in every wave ------------------------
    struct Item { int op; uint8_t flags; };
    std::vector<std::vector<Item>> items(W);
    std::vector<int>& posOf = s.posOf;
    posOf.assign(n, -1);
    std::vector<std::vector<int>>& phaseStart = s.phaseStart;  // first position after the barrier entry of each phase
    if ((int) phaseStart.size() < W) phaseStart.resize(W);
    for (int w = 0; w < W; ++w) phaseStart[w].clear();
    for (int ph = 0; ph < nphases; ++ph)
        for (int w = 0; w < W; ++w) {
            if (ph > 0) items[w].push_back(Item{-1, (uint8_t) (MBAMD_W4_NOP | MBAMD_W4_BARRIER)});
            phaseStart[w].push_back((int) items[w].size());
            for (int o : seq[ph][w]) { posOf[o] = (int) items[w].size(); items[w].push_back(Item{o, 0}); }
        }
    size_t longest = 0;
    for (int w = 0; w < W; ++w) longest = std::max(longest, items[w].size());
    t.W = W;
    t.phases = nphases;
    t.reloads = t.externals = t.evictions = 0;
    std::vector<std::vector<Walk4Template::Entry>>& fin = s.fin;   // final per-wave programs (PF entries inserted)
    if ((int) fin.size() < W) fin.resize(W);
    for (int w = 0; w < W; ++w) fin[w].clear();
    std::vector<Walk4Template::Entry>& scan = s.scan;

    // ---- slots, prefetches, wait counts: one linear scan per wave -------------------------------------------------
    int slotsUsed = 1;
    // (the operand pipeline of the general-state kernels reads evicted values from HBM itself: one slot -- the latest result -- is a valid budget)
    const int S = std::max(memSlots ? 2 : 1, std::min(W == 1 ? std::max(maxSlots, maxSlots1) : maxSlots, 250));
    // a short single-wave list is a root-ward path: fetch its siblings as early as slots allow -- loads issued before the
    // first store do not wait for any store (in-order vmcnt, see mbamd_walk4.h)
    const int distance = (W == 1 && n <= 128) ? (1 << 20) : prefetchDistance;
    std::vector<int>&slotHolder = s.slotHolder, &lastUse = s.lastUse, &slotOfVal = s.slotOfVal;   // slotHolder: value (op) or -2 - (prefetch id), -1 free
    slotHolder.assign(S, 0); lastUse.assign(n, 0); slotOfVal.assign(n, 0);
    struct Mem { int op, child, lo, use; int slot; bool issued; };
    std::vector<Mem> mems;
    std::vector<int>& memAt = s.memAt;                         // per position: first mem whose `use` is this position (sorted)
    memAt.clear();
    std::vector<int>& freeFrom = s.freeFrom;                   // slot is free for a DMA issued at positions >= freeFrom
    freeFrom.assign(S, 0);
    for (int w = 0; w < W; ++w) {
        const int L = (int) items[w].size();
        scan.assign((size_t) L, Walk4Template::Entry());
        Walk4Template::Entry* prog = scan.data();
        std::fill(slotHolder.begin(), slotHolder.end(), -1);
        std::fill(freeFrom.begin(), freeFrom.end(), 0);
        // last same-wave use of every result of this wave
        for (int j = 0; j < L; ++j) if (items[w][j].op >= 0) { lastUse[items[w][j].op] = -1; slotOfVal[items[w][j].op] = -1; }
        for (int j = 0; j < L; ++j) {
            const int o = items[w][j].op;
            if (o < 0) continue;
            const int pr[2] = {prod1[o], prod2[o]};
            for (int c = 0; c < 2; ++c)
                if (pr[c] >= 0 && waveOf[pr[c]] == w && posOf[pr[c]] < j && (!phasesAreLaunches || phaseOf[pr[c]] == phaseOf[o]))
                    lastUse[pr[c]] = j;
        }
        // forwarding: the position of the next operation after each position, and which results travel in registers
        std::vector<int>& nextOp = s.nextOp;
        std::vector<char>& fwd = s.fwd;
        nextOp.assign((size_t) L + 1, 1 << 30);
        for (int j = L - 1; j >= 0; --j) nextOp[j] = (j + 1 < L && items[w][j + 1].op >= 0) ? j + 1 : nextOp[j + 1];
        if ((int) fwd.size() < n) fwd.resize(n);
        for (int j = 0; j < L; ++j) {
            const int o = items[w][j].op;
            if (o < 0) continue;
            fwd[o] = forward && memSlots && !phasesAreLaunches && ncons[o] == 1 && lastUse[o] == nextOp[j];
        }
        // children that must come from memory (known up front): external buffers and results of other waves
        mems.clear();
        auto phaseLo = [&](int j) {                // first position of the phase that contains position j
            int lo = 0;
            for (int p : phaseStart[w]) if (p <= j) lo = p;
            return lo;
        };
        for (int j = 0; j < L; ++j) {
            const int o = items[w][j].op;
            if (o < 0) continue;
            const int pr[2] = {prod1[o], prod2[o]};
            const bool tip[2] = {ops[o].tip1 != 0, ops[o].tip2 != 0};
            for (int c = 0; c < 2; ++c) {
                if (tip[c]) continue;
                if (c == 1 && !tip[0] && ops[o].c2 == ops[o].c1) continue;        // same buffer twice: one copy serves both
                if (pr[c] < 0) { if (memSlots) mems.push_back(Mem{o, c, 0, j, -1, false}); t.externals++; }
                else if (waveOf[pr[c]] != w) { if (memSlots) mems.push_back(Mem{o, c, phaseLo(j), j, -1, false}); t.reloads++; }
            }
        }
        auto nextUseOfValue = [&](int v, int from) {           // next same-wave consumer position of value v at or after `from`
            if (!dag) {                                        // a forest: the one consumer of a value is its parent
                const int p = parent[v];
                if (p < 0 || waveOf[p] != w) return 1 << 30;
                return posOf[p] >= std::max(from, posOf[v] + 1) ? posOf[p] : (1 << 30);
            }
            for (int j = std::max(from, posOf[v] + 1); j < L; ++j) {
                const int o = items[w][j].op;
                if (o >= 0 && (prod1[o] == v || prod2[o] == v)) return j;
            }
            return 1 << 30;
        };
        auto evictOne = [&](int j, int keepA, int keepB) {      // free a slot holding a RESULT not needed at position j
            int best = -1, bestUse = -1;
            for (int sl = 0; sl < S; ++sl) {
                const int v = slotHolder[sl];
                if (v < 0 || v == keepA || v == keepB) continue;
                int u = nextUseOfValue(v, j);
                if (u == j) continue;
                if (posOf[v] == j - 1) u = j + 1;              // written a moment ago: the last candidate (its ds_write may still be in flight)
                if (u > bestUse) { bestUse = u; best = sl; }
            }
            if (best < 0) return -1;
            const int v = slotHolder[best];
            slotOfVal[v] = -1;
            slotHolder[best] = -1;
            t.evictions++;
            prog[posOf[v]].reread = true;
            // its remaining consumers read it from memory: no earlier than now, no earlier than its store was issued
            int uLo = std::max(j, posOf[v] + 1), uHi = L;
            if (!dag) { const int p = parent[v]; if (p >= 0 && waveOf[p] == w && posOf[p] >= uLo) { uLo = posOf[p]; uHi = uLo + 1; } else uHi = uLo; }
            for (int u = uLo; u < uHi; ++u) {
                const int o = items[w][u].op;
                if (o < 0) continue;
                if (prod1[o] == v) { if (memSlots) mems.push_back(Mem{o, 0, std::max(j, posOf[v] + 1), u, -1, false}); t.reloads++; }
                if (prod2[o] == v && !(prod1[o] == v)) { if (memSlots) mems.push_back(Mem{o, 1, std::max(j, posOf[v] + 1), u, -1, false}); t.reloads++; }
            }
            return best;
        };
        auto findFree = [&](int j) {
            for (int sl = 0; sl < S; ++sl) if (slotHolder[sl] == -1 && freeFrom[sl] <= j) return sl;
            return -1;
        };
        int npfAt = 0;
        for (int j = 0; j < L; ++j) {
            Walk4Template::Entry& e = prog[j];
            const int o = items[w][j].op;
            e.op = o;
            e.flags = items[w][j].flags;
            npfAt = 0;
            if (phasesAreLaunches && (e.flags & MBAMD_W4_BARRIER)) {      // a new launch starts with empty slots
                for (int q = 0; q < S; ++q) {
                    if (slotHolder[q] >= 0) slotOfVal[slotHolder[q]] = -1;
                    slotHolder[q] = -1;
                    freeFrom[q] = j;
                }
            }
            const int pr[2] = {o >= 0 ? prod1[o] : -1, o >= 0 ? prod2[o] : -1};
            // 1. prefetches issued with this entry: mandatory ones (consumer == this entry) first, then look-ahead
            for (int pass = 0; pass < 2; ++pass) {
                for (size_t mi = 0; mi < mems.size(); ++mi) {
                    Mem& m = mems[mi];
                    if (m.issued || m.lo > j) continue;
                    const bool mandatory = m.use == j;
                    if (pass == 0 ? !mandatory : (mandatory || m.use > j + distance || m.use < j)) continue;
                    if (m.use < j) continue;
                    if (npfAt >= 2) {
                        if (mandatory) return false;            // (three memory children of one entry cannot happen)
                        continue;
                    }
                    int sl = findFree(j);
                    if (sl >= 0 && !mandatory) {
                        // keep one slot in reserve for the results of the entries in between
                        int nfree = 0;
                        for (int q = 0; q < S; ++q) nfree += slotHolder[q] == -1 && freeFrom[q] <= j;
                        if (nfree < 2) continue;
                    }
                    if (sl < 0 && mandatory) sl = evictOne(j, pr[0], pr[1]);
                    if (sl < 0) { if (mandatory) return false; continue; }
                    m.slot = sl;
                    m.issued = true;
                    slotHolder[sl] = -2 - (int) mi;
                    e.pfOp[npfAt] = m.op; e.pfChild[npfAt] = (uint8_t) m.child; e.pfSlot[npfAt] = (uint8_t) sl;
                    ++npfAt;
                    slotsUsed = std::max(slotsUsed, sl + 1);
                }
            }
            if (o < 0) continue;
            // 2. the operation: where its children are
            uint8_t* cslot[2] = {&e.c1slot, &e.c2slot};
            const bool tip[2] = {ops[o].tip1 != 0, ops[o].tip2 != 0};
            for (int c = 0; c < 2; ++c) {
                if (tip[c]) continue;
                if (c == 1 && !tip[0] && ops[o].c2 == ops[o].c1) { e.c2slot = e.c1slot; continue; }
                if (pr[c] >= 0 && waveOf[pr[c]] == w && fwd[pr[c]] && lastUse[pr[c]] == j) { *cslot[c] = 0xFE; continue; }   // still in registers
                if (pr[c] >= 0 && waveOf[pr[c]] == w && slotOfVal[pr[c]] >= 0) { *cslot[c] = (uint8_t) slotOfVal[pr[c]]; continue; }
                int found = -1;
                for (size_t mi = 0; mi < mems.size(); ++mi)
                    if (mems[mi].op == o && mems[mi].child == c && mems[mi].issued) { found = (int) mi; break; }
                if (found < 0) {
                    if (!memSlots) {                                     // read from HBM by the kernel itself
                        // (its loads are issued one entry ahead: a value of this wave must have been stored before that)
                        if (pr[c] >= 0 && waveOf[pr[c]] == w && posOf[pr[c]] >= j - 1) return false;
                        *cslot[c] = 0xFF;
                        continue;
                    }
                    return false;
                }
                *cslot[c] = (uint8_t) mems[found].slot;
            }
            // children read for the last time release their slots: usable by this entry's result, by DMAs from j+1 on
            for (int c = 0; c < 2; ++c) {
                if (tip[c] || (c == 1 && !tip[0] && ops[o].c2 == ops[o].c1)) continue;
                const int sl = *cslot[c];
                if (sl == 0xFF || sl == 0xFE) continue;
                const int v = slotHolder[sl];
                if (v <= -2) { slotHolder[sl] = -1; freeFrom[sl] = j + 1; }
                else if (v >= 0 && lastUse[v] <= j) { slotHolder[sl] = -1; freeFrom[sl] = j + 1; slotOfVal[v] = -1; }
            }
            // (alwaysKeep) results nobody in this wave reads again were copied out by the writer during the previous entry
            if (alwaysKeep)
                for (int q = 0; q < S; ++q) {
                    const int v = slotHolder[q];
                    if (v >= 0 && lastUse[v] < 0 && posOf[v] < j) { slotHolder[q] = -1; freeFrom[q] = j + 1; slotOfVal[v] = -1; }
                }
            // 3. the result: kept in a slot if this wave reads it again
            if ((lastUse[o] > j && !fwd[o]) || alwaysKeep) {
                int sl = -1;
                for (int q = 0; q < S; ++q) if (slotHolder[q] == -1) { sl = q; break; }   // (freed children included: reads precede the write)
                if (sl < 0) {
                    // evict the value needed farthest in the future -- possibly this result itself
                    const int mine = alwaysKeep ? -1 : nextUseOfValue(o, j + 1);     // (alwaysKeep: this result must get a slot)
                    int best = -1, bestUse = mine;
                    for (int q = 0; q < S; ++q) {
                        const int v = slotHolder[q];
                        if (v < 0) continue;
                        const int u = nextUseOfValue(v, j + 1);
                        // (a result the NEXT entry reads must be kept -- it cannot come back from HBM that soon -- even against a
                        //  value needed just as soon: that one was stored long enough ago)
                        if (u > bestUse || (!memSlots && mine == j + 1 && best < 0 && u >= bestUse)) { bestUse = u; best = q; }
                    }
                    if (best >= 0) {
                        const int v = slotHolder[best];
                        slotOfVal[v] = -1;
                        slotHolder[best] = -1;
                        t.evictions++;
                        prog[posOf[v]].reread = true;
                        int uLo = j + 1, uHi = L;
                        if (!dag) { const int p = parent[v]; if (p >= 0 && waveOf[p] == w && posOf[p] >= uLo) { uLo = posOf[p]; uHi = uLo + 1; } else uHi = uLo; }
                        for (int u = uLo; u < uHi; ++u) {
                            const int q = items[w][u].op;
                            if (q < 0) continue;
                            if (prod1[q] == v) { if (memSlots) mems.push_back(Mem{q, 0, j + 1, u, -1, false}); t.reloads++; }
                            if (prod2[q] == v && prod1[q] != v) { if (memSlots) mems.push_back(Mem{q, 1, j + 1, u, -1, false}); t.reloads++; }
                        }
                        sl = best;
                    }
                }
                if (sl >= 0) {
                    slotHolder[sl] = o;
                    slotOfVal[o] = sl;
                    e.dslot = (uint8_t) sl;
                    slotsUsed = std::max(slotsUsed, sl + 1);
                } else {
                    t.evictions++;
                    e.reread = true;
                    int uLo = j + 1, uHi = L;
                    if (!dag) { const int p = parent[o]; if (p >= 0 && waveOf[p] == w && posOf[p] >= uLo) { uLo = posOf[p]; uHi = uLo + 1; } else uHi = uLo; }
                    for (int u = uLo; u < uHi; ++u) {           // not kept: its consumers prefetch it (after this entry's store)
                        const int q = items[w][u].op;
                        if (q < 0) continue;
                        if (prod1[q] == o) { if (memSlots) mems.push_back(Mem{q, 0, j + 1, u, -1, false}); t.reloads++; }
                        if (prod2[q] == o && prod1[q] != o) { if (memSlots) mems.push_back(Mem{q, 1, j + 1, u, -1, false}); t.reloads++; }
                    }
                }
            }
        }
        // 4. final program: a PF entry in front of every entry that carries prefetches
        std::vector<Walk4Template::Entry>& out = fin[w];
        out.clear();
        for (int j = 0; j < L; ++j) {
            Walk4Template::Entry e = prog[j];
            if (e.pfOp[0] >= 0) {
                Walk4Template::Entry pf;
                pf.flags = MBAMD_W4_NOP;
                for (int q = 0; q < 2; ++q) { pf.pfOp[q] = e.pfOp[q]; pf.pfChild[q] = e.pfChild[q]; pf.pfSlot[q] = e.pfSlot[q]; }
                out.push_back(pf);
                e.pfOp[0] = e.pfOp[1] = -1;
            }
            out.push_back(e);
        }
        // 5. wait counts: replay the vector-memory instruction sequence of the kernel loop (mbamd_walk4.h)
        //    prologue: [exponent DMA for entry 0 if SCALE_READ]
        //    iteration: [PF DMAs] WAIT [exponent DMA for the next entry if SCALE_READ] [2 stores if an operation]
        auto reads = [&](const Walk4Template::Entry& e) { return e.op >= 0 && ops[e.op].scaleRead >= 0 && ops[e.op].scaleWrite < 0; };
        long issued = 0;
        long expSeq = -1;                                       // sequence number of the exponent DMA of the entry about to run
        if (!out.empty() && reads(out[0])) expSeq = issued++;
        std::vector<long> pfSeq(mems.size(), -1);
        for (size_t j = 0; j < out.size(); ++j) {
            Walk4Template::Entry& e = out[j];
            for (int q = 0; q < 2; ++q)
                if (e.pfOp[q] >= 0) {
                    for (size_t mi = 0; mi < mems.size(); ++mi)
                        if (mems[mi].op == e.pfOp[q] && mems[mi].child == e.pfChild[q] && mems[mi].issued && pfSeq[mi] < 0 &&
                            mems[mi].slot == e.pfSlot[q]) { pfSeq[mi] = issued; break; }
                    ++issued;
                }
            long needed = -1;
            if (e.op >= 0) {
                for (size_t mi = 0; mi < mems.size(); ++mi)
                    if (mems[mi].op == e.op && pfSeq[mi] >= 0) needed = std::max(needed, pfSeq[mi]);
                if (reads(e)) needed = std::max(needed, expSeq);
            }
            if (needed < 0) e.vmwait = 0xFF;                        // (no wait)
            else e.vmwait = safeWaits ? 0 : (uint8_t) walk4_round_wait(issued - (needed + 1));
            expSeq = -1;
            if (j + 1 < out.size() && reads(out[j + 1])) expSeq = issued++;
            if (e.op >= 0) issued += 2;
        }
    }
    size_t longestFinal = 0;
    for (int w = 0; w < W; ++w) longestFinal = std::max(longestFinal, fin[w].size());
    // frame: leading NOPs, the programs padded to a multiple of the kernel's loop unroll factor, read-ahead NOPs
    const int body = (leadNops + (int) longestFinal + unroll - 1) / unroll * unroll;
    const int tail = tailNops;
    const int entries = body + tail;
    t.entries = entries;
    t.tail = tail;
    t.prog.assign((size_t) W * entries, Walk4Template::Entry());
    for (Walk4Template::Entry& e : t.prog) e.flags = MBAMD_W4_NOP;
    for (int w = 0; w < W; ++w) std::copy(fin[w].begin(), fin[w].end(), t.prog.begin() + (size_t) w * entries + leadNops);
    t.nslots = slotsUsed;
    return true;
}

}  // namespace mbamd
#endif
