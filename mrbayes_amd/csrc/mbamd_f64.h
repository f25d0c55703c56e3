// mbamd_f64.h -- the double-precision engine behind BEAGLE_FLAG_PRECISION_DOUBLE (`set beagleprecision=double`, reference
// src/command.c:6760-6766; the build it corresponds to is the reference with CLFlt = double, src/bayes.h:110-112).
// Included by mbamd_engine.cpp.  A compact, level-synchronous engine for any state count <= 64: conditional likelihoods,
// transition matrices, sums and logarithms all in fp64; rescaling by exact powers of two with integer exponents, like the
// fp32 engine.  First correct path: one launch per dependency level and kind, no tree walk, no matrix cores -- the fp32
// engines are the optimised ones, this one exists so that the precision flag of the seam means what it says.
//
// HBM layout: partials double [buffer][K][S][P_pad] (a thread owns one pattern; every access is coalesced across
// patterns), compact tips uint8 [P_pad], matrices double [buffer][K][S][S] (row = from-state, for the edge integration)
// followed by the transposed copy [K][S][SPAD] with zero-padded rows (the operand of the partials kernel: wave-uniform,
// read through the scalar cache), scale buffers int32 [P_pad] (binary exponents; cumulative buffers are their sums).
#ifndef MBAMD_F64_H_
#define MBAMD_F64_H_

namespace mbamd {

struct Op64 {
    double* dst;
    const void* c1;              // partials (double) or compact states (uint8)
    const void* c2;
    const double* m1T;           // transposed matrices of child 1: [K][S][SPAD]
    const double* m2T;
    int32_t* scale;              // exponents written (mode 1) or read (mode 2)
    int32_t* cum;                // cumulative exponents the written ones are added to, or null
    int c1_tip, c2_tip, mode;
    int first, last;             // the operation covers patterns [first, last): everything, or one partition (v3 *ByPartition)
    int pad_;
};

template <int IB>
__device__ __forceinline__ void f64_child_factor(const void* ptr, int tip, const double* __restrict__ mT, int S, int SPAD, int k,
                                                 size_t Ppad, size_t c, int i0, double (&f)[IB])
{
    if (tip) {
        const unsigned s = reinterpret_cast<const uint8_t*>(ptr)[c];
        if (s >= (unsigned) S) {
#pragma unroll
            for (int i = 0; i < IB; ++i) f[i] = 1.0;
        } else {
            const double* col = mT + (size_t) s * SPAD + i0;           // P(i -> s), all i: contiguous
#pragma unroll
            for (int i = 0; i < IB; ++i) f[i] = col[i];
        }
    } else {
        const double* cl = reinterpret_cast<const double*>(ptr) + (size_t) k * S * Ppad + c;
#pragma unroll
        for (int i = 0; i < IB; ++i) f[i] = 0.0;
        for (int j = 0; j < S; ++j) {
            const double vj = cl[(size_t) j * Ppad];
            const double* __restrict__ col = mT + (size_t) j * SPAD + i0;
#pragma unroll
            for (int i = 0; i < IB; ++i) f[i] = fma(col[i], vj, f[i]);
        }
    }
}

// CondLikeDown_* in fp64 (reference src/likelihood.c:204-375 with CLFlt = double): grid (P_pad/64, operations of a level)
template <int IB>
__global__ void __launch_bounds__(64)
k64_partials(const Op64* __restrict__ ops, int S, int SPAD, int K, int Ppad_)
{
    const Op64& op = ops[blockIdx.y];
    const size_t Ppad = (size_t) Ppad_, c = (size_t) blockIdx.x * 64 + threadIdx.x;
    if (c < (size_t) op.first || c >= (size_t) op.last) return;
    // blockIdx.z = (category, state block): a few hundred waves of patterns alone leave the chip empty
    const int nib = SPAD / IB, k = (int) blockIdx.z / nib, i0 = ((int) blockIdx.z % nib) * IB;
    (void) K;
    double f1[IB], f2[IB];
    f64_child_factor<IB>(op.c1, op.c1_tip, op.m1T + (size_t) k * S * SPAD, S, SPAD, k, Ppad, c, i0, f1);
    f64_child_factor<IB>(op.c2, op.c2_tip, op.m2T + (size_t) k * S * SPAD, S, SPAD, k, Ppad, c, i0, f2);
#pragma unroll
    for (int i = 0; i < IB; ++i)
        if (i0 + i < S) op.dst[((size_t) k * S + i0 + i) * Ppad + c] = f1[i] * f2[i];
}


// CondLikeDown_Gen / _NY98 in fp64 on the fp64 MATRIX cores (v_mfma_f64_16x16x4_f64) for 16 <= S <= 64: a wave owns 16 patterns of
// one category and all states of the destination -- NT = ceil(S / 16) output tiles of 16 states x 16 patterns, each the sum over
// ceil(S / 4) steps of A (16 out-states x 4 in-states, from the TRANSPOSED matrix copy: 16 consecutive doubles per lane row) times
// B (4 in-states x 16 patterns of the child: 16 consecutive doubles).  Accumulator register r of lane (n = lane & 15, g = lane >> 4)
// is state 16 it + g + 4 r of pattern n.  A compact tip's factor is a gather from the transposed matrix laid out for the same
// registers.  grid (P_pad / 16, operations of a level, K).  Round 2's k64_partials<IB> ran this contraction on the vector ALU with
// the matrix column through the scalar cache: 154 us per level at protein 200 x 10 000, 8.7 ms per codon M3 evaluation.
// the 16-pattern product tiles of category k: p[it][r] = state 16 it + g + 4 r of pattern n
template <int NT>
__device__ __forceinline__ void f64_mfma_tiles(const MBAMD_AS_CONST Op64* op, int S, int SPAD, size_t Ppad, int k, size_t c, int n, int g,
                                               double __attribute__((ext_vector_type(4))) (&p)[NT])
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int steps = (S + 3) / 4;
    d4 f[2][NT];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const void* ptr = ch ? op->c2 : op->c1;
        const bool tip = ch ? op->c2_tip : op->c1_tip;
        const MBAMD_AS_GLOBAL double* mT = as_global(ch ? op->m2T : op->m1T) + (size_t) k * S * SPAD;
        if (tip) {
            const unsigned st = as_global(reinterpret_cast<const uint8_t*>(ptr))[c];
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + g + 4 * r;
                    f[ch][it][r] = st >= (unsigned) S ? 1.0 : (i < S ? mT[(size_t) st * SPAD + i] : 0.0);
                }
            continue;
        }
        const MBAMD_AS_GLOBAL double* cl = as_global(reinterpret_cast<const double*>(ptr)) + (size_t) k * S * Ppad + c;
#pragma unroll
        for (int it = 0; it < NT; ++it) f[ch][it] = (d4) (0.0);
        for (int t0 = 0; t0 < steps; t0 += 4) {              // four steps' operands in flight
            double b[4], a[4][NT];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * (t0 + u) + g;              // in-state of this lane's operand rows
                const int jc = j < S ? j : S - 1;
                b[u] = j < S ? cl[(size_t) jc * Ppad] : 0.0;
#pragma unroll
                for (int it = 0; it < NT; ++it) {
                    const int i = 16 * it + n;               // out-state of this lane's A row
                    a[u][it] = (j < S && i < S) ? mT[(size_t) jc * SPAD + (i < SPAD ? i : 0)] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int it = 0; it < NT; ++it) f[ch][it] = mbd_mfma_f64_16x16x4(a[u][it], b[u], f[ch][it]);
        }
    }
#pragma unroll
    for (int it = 0; it < NT; ++it) p[it] = f[0][it] * f[1][it];
}

// KF = 0: one category per wave (blockIdx.z), the rescale in its own pass (k64_rescale); KF = K > 0: a wave computes all K
// categories of its 16 patterns and rescales in registers (CondLikeScaler_*: per-pattern maximum over categories and states --
// the four lane groups of a pattern meet through two lane exchanges), one pass over HBM instead of three.
template <int NT, int KF>
__global__ void __launch_bounds__(64)
k64_partials_mfma(const Op64* __restrict__ ops, int S, int SPAD, int Ppad_)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    const MBAMD_AS_CONST Op64* op = as_const(ops) + blockIdx.y;
    const size_t Ppad = (size_t) Ppad_;
    const int lane = (int) threadIdx.x, n = lane & 15, g = lane >> 4;
    const size_t c = (size_t) blockIdx.x * 16 + n;
    if ((size_t) blockIdx.x * 16 + 16 <= (size_t) op->first || (size_t) blockIdx.x * 16 >= (size_t) op->last) return;   // (wave-uniform)
    const bool mine = c >= (size_t) op->first && c < (size_t) op->last;
    if constexpr (KF == 0) {
        const int k = (int) blockIdx.z;
        d4 p[NT];
        f64_mfma_tiles<NT>(op, S, SPAD, Ppad, k, c, n, g, p);
        MBAMD_AS_GLOBAL double* dst = as_global(op->dst) + (size_t) k * S * Ppad + c;
        if (mine) {
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + g + 4 * r;
                    if (i < S) dst[(size_t) i * Ppad] = p[it][r];
                }
        }
    } else {
        d4 p[KF][NT];
        double mx = 0.0;
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            f64_mfma_tiles<NT>(op, S, SPAD, Ppad, k, c, n, g, p[k]);
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * it + g + 4 * r < S) mx = fmax(mx, p[k][it][r]);
        }
        mx = fmax(mx, mbd_shfl_xor(mx, 16));
        mx = fmax(mx, mbd_shfl_xor(mx, 32));
        int e = 0;
        if (op->mode == 1) {
            if (mx > 0.0 && mx < 1.0e300) (void) frexp(mx, &e);
            e = e < -1000 ? -1000 : e;
            if (mine && g == 0) {
                as_global(op->scale)[c] = e;
                if (op->cum != nullptr && e != 0) atomicAdd(op->cum + c, e);
            }
        } else if (op->mode == 2) {
            e = as_global(op->scale)[c];
        }
        if (mine) {
#pragma unroll
            for (int k = 0; k < KF; ++k) {
                MBAMD_AS_GLOBAL double* dst = as_global(op->dst) + (size_t) k * S * Ppad + c;
#pragma unroll
                for (int it = 0; it < NT; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * it + g + 4 * r;
                        if (i < S) dst[(size_t) i * Ppad] = e != 0 ? ldexp(p[k][it][r], -e) : p[k][it][r];
                    }
            }
        }
    }
}

// The same contraction with the matrices through LDS.  On the kernel above every wave fetches both transposed matrices itself -- at 61
// states 62 KiB through the CU's L1 for 16 patterns, 20 eight-byte loads per 16 matrix instructions, and the L1's 64 B/clk are spent
// at a quarter of the matrix cores' rate.  Here a workgroup of four waves (64 patterns of one operation) parks the matrices of its
// non-tip children in LDS once, in FRAGMENT order -- the 64 lanes' A operands of (step t, tile it) are 64 consecutive doubles, so a
// wave's read is one conflict-free ds_read_b64 -- and only the child's partials (one load per four matrix instructions, the next
// group's in flight behind the current group's arithmetic) still come through the L1.  Same instructions on the same operands in
// the same order as above: the results are bit-identical.  grid (P_pad / 64, operations, K if unfused), 256 threads,
// dynamic LDS 2 * max(KF, 1) * stepsP * NT * 64 doubles, stepsP = ceil(S / 4) rounded up to a multiple of four (64 KiB at 61 states).
// (PRE: the first group's partials of both children were loaded before the matrices were parked -- `pre`.  Requesting ALL of a child's
//  partials ahead was measured on the four-wave workgroups of the narrow levels, where nothing else hides a cold load: at the start of
//  its contraction 2.8 -> 3.6 us per child, before the matrices are parked 2.8 -> 2.3 us but the load phase 3.3 -> 5.0 us -- that phase is
//  the level's read burst at HBM bandwidth (every workgroup of the level loads at the same time), not latency.  profiles/r04_f64.txt)
template <int NT, bool PRE>
__device__ __forceinline__ void f64_mfma_tiles_lds(const MBAMD_AS_CONST Op64* op, int S, int SPAD, size_t Ppad, int k, size_t c, int n, int g, int lane,
                                                   const double* lds1, const double* lds2, const double (&pre)[2][4],
                                                   double __attribute__((ext_vector_type(4))) (&p)[NT]
                                                   )
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int stepsP = (((S + 3) / 4) + 3) & ~3;           // steps of four in-states, padded to the groups of four the loop runs
    d4 f[2][NT];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const void* ptr = ch ? op->c2 : op->c1;
        const bool tip = ch ? op->c2_tip : op->c1_tip;
        if (tip) {
            // the column of the tip's state from the PARKED matrix (element (row st, column i) of fragment order; zero beyond S): gathered
            // from global memory this was 256 L2 requests per wave, as many as everything else the wave reads
            const unsigned st = as_global(reinterpret_cast<const uint8_t*>(ptr))[c];
            const unsigned sc = st >= (unsigned) S ? 0u : st;
            const double* row = (ch ? lds2 : lds1) + (size_t) ((sc >> 2) * NT) * 64 + (sc & 3u) * 16 + g;
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = row[it * 64 + 4 * r];
                    f[ch][it][r] = st >= (unsigned) S ? 1.0 : v;
                }
            continue;
        }
        const double* la = (ch ? lds2 : lds1) + lane;
        const MBAMD_AS_GLOBAL double* cl = as_global(reinterpret_cast<const double*>(ptr)) + (size_t) k * S * Ppad + c;
#pragma unroll
        for (int it = 0; it < NT; ++it) f[ch][it] = (d4) (0.0);
        // (every load below is unconditional -- clamped row, the value MULTIPLIED by one or zero: a select would be turned into a branch
        //  around the load -- and the loop body straight-line code: a load inside a branch makes the compiler wait for ALL loads at the
        //  join, the prefetched ones included)
        double b[4], bn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 4 * u + g;
            if constexpr (PRE) b[u] = pre[ch][u] * (j < S ? 1.0 : 0.0);       // (multiplied here, not where it was loaded: that would wait for it there)
            else b[u] = cl[(size_t) (j < S ? j : S - 1) * Ppad] * (j < S ? 1.0 : 0.0);
        }
        for (int t0 = 0; t0 < stepsP; t0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {                    // the next group's partials, behind this group's arithmetic
                const int j = 4 * (t0 + 4 + u) + g;
                bn[u] = cl[(size_t) (j < S ? j : S - 1) * Ppad] * (j < S ? 1.0 : 0.0);
            }
            double a[4][NT];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int it = 0; it < NT; ++it) a[u][it] = la[(size_t) ((t0 + u) * NT + it) * 64];      // (zero rows beyond S)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int it = 0; it < NT; ++it) f[ch][it] = mbd_mfma_f64_16x16x4(a[u][it], b[u], f[ch][it]);
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = bn[u];
        }
    }
#pragma unroll
    for (int it = 0; it < NT; ++it) p[it] = f[0][it] * f[1][it];
}

// NW waves per workgroup (16 NW patterns): 8 on the large levels -- the LDS the matrices take allows two workgroups per CU, and two
// waves per SIMD leave the matrix cores idle 60 % of the time (a wave also gathers tips, stores, and waits for its stores to drain);
// 4 on the small ones, where 8 would leave CUs without work.
// (one operation on the workgroup's 16 NW patterns; every thread of the workgroup passes the one barrier inside.  Walking the narrow
//  levels at the top of the tree as chains inside ONE launch of this function -- a workgroup's patterns only depend on the same patterns
//  of the children -- was measured and dropped: 152 us against 158 us for the eleven launches it replaced, an operation is a 11 us latency
//  chain in either form, profiles/r04_f64.txt.)
template <int NT, int KF, int NW>
__device__ __forceinline__ void f64_lds_operation(const MBAMD_AS_CONST Op64* op, double* lds, int S, int SPAD, size_t Ppad)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    constexpr int KL = KF > 0 ? KF : 1;                      // (a matrix is at most 4 NT steps x NT tiles blocks of 64 lanes)
    const int tid = (int) threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int stepsP = (((S + 3) / 4) + 3) & ~3, nb = stepsP * NT, frag = nb * 64;         // doubles of one matrix in fragment order (zero rows beyond S)
    const bool inRange = !((size_t) blockIdx.x * (16 * NW) + 16 * NW <= (size_t) op->first || (size_t) blockIdx.x * (16 * NW) >= (size_t) op->last);   // (workgroup-uniform)
    const size_t tile0 = (size_t) blockIdx.x * (16 * NW) + (size_t) wave * 16;
    const bool waveIn = inRange && !(tile0 + 16 <= (size_t) op->first || tile0 >= (size_t) op->last);      // (wave-uniform)
    const size_t c = tile0 + n;
    // ---- the first group of the children's partials (first category): in flight while the matrices are parked ---------------------------------
    double pre[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    if (waveIn) {
        const int k0 = KF > 0 ? 0 : (int) blockIdx.z;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (ch ? op->c2_tip : op->c1_tip) continue;
            const MBAMD_AS_GLOBAL double* cl = as_global(reinterpret_cast<const double*>(ch ? op->c2 : op->c1)) + (size_t) k0 * S * Ppad + c;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * u + g;
                pre[ch][u] = cl[(size_t) (j < S ? j : S - 1) * Ppad];
            }
        }
    }
    // ---- the matrices into LDS: element (t, it, lane = (n', g')) = mT[(4 t + g') * SPAD + 16 it + n'] -----------------------------------------
    // (branch-free -- a tip child's matrix is parked too, unused -- so that all loads, up to 32 per thread, are in flight together:
    //  clamped addresses, values multiplied by one or zero, then the LDS stores)
    if (inRange) {
        // two adjacent columns per lane and load (16 bytes): unit u = doubles 2u, 2u + 1 of fragment order = lanes (g, n = 2 n2), (g, 2 n2 + 1)
        // of block u / 32 -- half the load instructions of a double per lane
        typedef double d2 __attribute__((ext_vector_type(2)));
        constexpr int R2 = (2 * NT * NT + NW - 1) / NW;       // units per thread and matrix: 32 per block over 64 NW threads
        d2 tmp[KL][2][R2];
#pragma unroll
        for (int kk = 0; kk < KL; ++kk) {
            const int k = KF > 0 ? kk : (int) blockIdx.z;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const MBAMD_AS_GLOBAL double* mT = as_global(ch ? op->m2T : op->m1T) + (size_t) k * S * SPAD;
#pragma unroll
                for (int r = 0; r < R2; ++r) {
                    const int u = r * (64 * NW) + tid, uc = u < nb * 32 ? u : nb * 32 - 1;
                    const int bc = uc >> 5, w = uc & 31, gg = w >> 3, n2 = w & 7;
                    const int it = bc % NT, t = bc / NT;
                    const int j = 4 * t + gg, i = 16 * it + 2 * n2;
                    const MBAMD_AS_GLOBAL double* src = mT + (size_t) (j < S ? j : S - 1) * SPAD + (i < S ? i : 0);
                    d2 v;
                    __builtin_memcpy(&v, (const void*) src, sizeof v);            // (one 16-byte load; the transposed copies start at an odd multiple of 8 bytes at 61 states)
                    v.x *= (j < S && i < S) ? 1.0 : 0.0;
                    v.y *= (j < S && i + 1 < S) ? 1.0 : 0.0;
                    tmp[kk][ch][r] = v;
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < KL; ++kk)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                d2* dstl = reinterpret_cast<d2*>(lds + (size_t) (ch * KL + kk) * frag);
#pragma unroll
                for (int r = 0; r < R2; ++r)
                    if (r * (64 * NW) + tid < nb * 32) dstl[r * (64 * NW) + tid] = tmp[kk][ch][r];
            }
    }
    MBAMD_SYNC();
    if (!waveIn) return;                                     // (no barrier below)
    const bool mine = c >= (size_t) op->first && c < (size_t) op->last;
    if constexpr (KF == 0) {
        const int k = (int) blockIdx.z;
        d4 p[NT];
        f64_mfma_tiles_lds<NT, true>(op, S, SPAD, Ppad, k, c, n, g, lane, lds, lds + frag, pre, p);
        MBAMD_AS_GLOBAL double* dst = as_global(op->dst) + (size_t) k * S * Ppad + c;
        if (mine) {
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + g + 4 * r;
                    if (i < S) dst[(size_t) i * Ppad] = p[it][r];
                }
        }
    } else {
        d4 p[KF][NT];
        double mx = 0.0;
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            if (k == 0) f64_mfma_tiles_lds<NT, true>(op, S, SPAD, Ppad, k, c, n, g, lane, lds + (size_t) k * frag, lds + (size_t) (KF + k) * frag, pre, p[k]);
            else f64_mfma_tiles_lds<NT, false>(op, S, SPAD, Ppad, k, c, n, g, lane, lds + (size_t) k * frag, lds + (size_t) (KF + k) * frag, pre, p[k]);
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * it + g + 4 * r < S) mx = fmax(mx, p[k][it][r]);
        }
        mx = fmax(mx, mbd_shfl_xor(mx, 16));
        mx = fmax(mx, mbd_shfl_xor(mx, 32));
        int e = 0;
        if (op->mode == 1) {
            if (mx > 0.0 && mx < 1.0e300) (void) frexp(mx, &e);
            e = e < -1000 ? -1000 : e;
            if (mine && g == 0) {
                as_global(op->scale)[c] = e;
                if (op->cum != nullptr && e != 0) atomicAdd(op->cum + c, e);
            }
        } else if (op->mode == 2) {
            e = as_global(op->scale)[c];
        }
        if (mine) {
#pragma unroll
            for (int k = 0; k < KF; ++k) {
                MBAMD_AS_GLOBAL double* dst = as_global(op->dst) + (size_t) k * S * Ppad + c;
#pragma unroll
                for (int it = 0; it < NT; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * it + g + 4 * r;
                        if (i < S) dst[(size_t) i * Ppad] = e != 0 ? ldexp(p[k][it][r], -e) : p[k][it][r];
                    }
            }
        }
    }
}

// (second launch bound: four waves per SIMD leave 128 registers -- the wide instantiations, NT x KF >= 6, spilled 5 ... 53 of theirs to
//  scratch at eight waves per workgroup; they get two waves per SIMD, i.e. one such workgroup per CU, and no scratch)
template <int NT, int KF, int NW>
__global__ void __launch_bounds__(64 * NW, (NW == 8 && NT * KF >= 6) ? 2 : NW / 2)
k64_partials_mfma_lds(const Op64* __restrict__ ops, int S, int SPAD, int Ppad_)
{
    f64_lds_operation<NT, KF, NW>(as_const(ops) + blockIdx.y, mbd_dyn_lds<double>(), S, SPAD, (size_t) Ppad_);
}

// (a descriptor by value, member by member: the scalar loads are issued where this is called, the wait is where a member is first used)
__device__ __forceinline__ Op64 f64_load_op(const MBAMD_AS_CONST Op64* p)
{
    Op64 d;
    d.dst = p->dst; d.c1 = p->c1; d.c2 = p->c2; d.m1T = p->m1T; d.m2T = p->m2T; d.scale = p->scale; d.cum = p->cum;
    d.c1_tip = p->c1_tip; d.c2_tip = p->c2_tip; d.mode = p->mode; d.first = p->first; d.last = p->last; d.pad_ = p->pad_;
    return d;
}

// A CHAIN of operations -- each one's result a child of the next: the root-ward path of an MCMC move -- in one launch, the running result
// kept in registers.  The accumulator layout of v_mfma_f64_16x16x4_f64 IS its B layout: register r of output tile it at lane (n, g) holds
// state 16 it + g + 4 r of pattern n, which is what step t = 4 it + r of the next contraction wants from that lane.  So the result of an
// operation feeds the next one's matrix instructions as it stands (the rescaled values -- the same doubles that go to HBM for later
// lists), and a level costs the matrix instructions of its two contractions instead of a launch, a read burst from HBM and a drain
// (11 - 15 us per level on the level kernels).  Per operation: the matrices (fetched into registers during the previous operation) are
// parked in LDS in fragment order, the SIBLING's partials (fetched then too) and the running result are contracted, product, rescale,
// store; a tip sibling is a column of the parked matrix.  A workgroup = four waves = 64 patterns of one chain (a codon model's eigen
// parts are chains of their own); no pattern partitions.  Op64::pad_: 0 first operation of a chain (child 1 from memory or a tip,
// child 2 the "sibling"), 1 / 2 = child 1 / 2 is the previous result.  The same instructions on the same operands: bit-identical to
// the level kernels.  grid (P_pad / 64, chains), 256 threads, dynamic LDS as k64_partials_mfma_lds.
template <int NT, int KF>
__global__ void __launch_bounds__(256)
k64_partials_chain(const Op64* __restrict__ ops, const int* __restrict__ chainStart, int S, int SPAD, int Ppad_)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int NW = 4, R2 = (2 * NT * NT + NW - 1) / NW;
    double* lds = mbd_dyn_lds<double>();
    const size_t Ppad = (size_t) Ppad_;
    const int tid = (int) threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int stepsP = (((S + 3) / 4) + 3) & ~3, nb = stepsP * NT, frag = nb * 64;
    const int ob = chainStart[blockIdx.y], oe = chainStart[blockIdx.y + 1];
    const size_t c = (size_t) blockIdx.x * 64 + (size_t) wave * 16 + n;
    d4 prev[KF][NT];                                         // the previous operation's (rescaled) result
    d2 mt[KF][2][R2];                                        // the next operation's matrices on their way to LDS
    double sib[KF][NT][4];                                   // the next operation's sibling partials: [category][group of four steps][step]
#pragma unroll
    for (int k = 0; k < KF; ++k)
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            prev[k][it] = (d4) (0.0);
#pragma unroll
            for (int u = 0; u < 4; ++u) sib[k][it][u] = 0.0;
        }
    auto fetchMatrices = [&](const Op64& d) {
#pragma unroll
        for (int k = 0; k < KF; ++k)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const MBAMD_AS_GLOBAL double* mT = as_global(ch ? d.m2T : d.m1T) + (size_t) k * S * SPAD;
#pragma unroll
                for (int r = 0; r < R2; ++r) {
                    const int u = r * (64 * NW) + tid, uc = u < nb * 32 ? u : nb * 32 - 1;
                    const int bc = uc >> 5, w = uc & 31, gg = w >> 3, n2 = w & 7;
                    const int it = bc % NT, t = bc / NT;
                    const int j = 4 * t + gg, i = 16 * it + 2 * n2;
                    const MBAMD_AS_GLOBAL double* src = mT + (size_t) (j < S ? j : S - 1) * SPAD + (i < S ? i : 0);
                    d2 v;
                    __builtin_memcpy(&v, (const void*) src, sizeof v);
                    mt[k][ch][r] = v;                        // (as loaded: masked where it is written to LDS -- a multiplication here would wait for the load)
                }
            }
    };
    unsigned sibState = 0;                                   // the next operation's sibling, if it is a tip: its state
    int storedExp = 0;                                       // the next operation's exponent, if it re-uses a stored one (mode 2)
    auto fetchSibling = [&](const Op64& d) {
        // (the same loads whatever the sibling is -- a tip's "partials" are read from the operation's own destination, a node's "state"
        //  from the chain table, both unused: a load inside a branch would make every later wait a wait for everything in flight)
        const int si = d.pad_ == 2 ? 0 : 1;                // the sibling is child 2 unless child 2 is the chain
        const bool tip = si ? d.c2_tip : d.c1_tip;
        const void* sp = si ? d.c2 : d.c1;
        sibState = as_global(reinterpret_cast<const uint8_t*>(tip ? sp : (const void*) chainStart))[tip ? c : 0];
        storedExp = as_global(d.mode == 2 ? (const int32_t*) d.scale : (const int32_t*) chainStart)[d.mode == 2 ? c : 0];   // (a stored exponent: ahead as well)
        const MBAMD_AS_GLOBAL double* cl = as_global(tip ? (const double*) d.dst : reinterpret_cast<const double*>(sp)) + c;
#pragma unroll
        for (int k = 0; k < KF; ++k)
#pragma unroll
            for (int gq = 0; gq < NT; ++gq)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = 4 * (4 * gq + u) + g;
                    sib[k][gq][u] = cl[(size_t) k * S * Ppad + (size_t) (j < S ? j : S - 1) * Ppad];
                }
    };
    // the descriptors through the scalar cache TWO operations ahead (a first touch is ~1 us: read where they are used, that is two or
    // three serial misses per operation)
    if (ob >= oe) return;
    Op64 dcur = f64_load_op(as_const(ops) + ob), dnxt = f64_load_op(as_const(ops) + (ob + 1 < oe ? ob + 1 : ob));
    fetchMatrices(dcur);
    fetchSibling(dcur);
    for (int o = ob; o < oe; ++o) {
        const Op64 dnn = f64_load_op(as_const(ops) + (o + 2 < oe ? o + 2 : oe - 1));
        const Op64* op = &dcur;
#pragma unroll
        for (int k = 0; k < KF; ++k)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                d2* dstl = reinterpret_cast<d2*>(lds + (size_t) (ch * KF + k) * frag);
#pragma unroll
                for (int r = 0; r < R2; ++r) {
                    const int u = r * (64 * NW) + tid;
                    if (u < nb * 32) {
                        const int bc = u >> 5, w = u & 31, it = bc % NT, t = bc / NT;
                        const int j = 4 * t + (w >> 3), i = 16 * it + 2 * (w & 7);
                        d2 v = mt[k][ch][r];
                        v.x *= (j < S && i < S) ? 1.0 : 0.0;
                        v.y *= (j < S && i + 1 < S) ? 1.0 : 0.0;
                        dstl[u] = v;
                    }
                }
            }
        if (o + 1 < oe) fetchMatrices(dnxt);               // (in flight during this operation's arithmetic)
        MBAMD_SYNC();
        const int cc = op->pad_, si = cc == 2 ? 0 : 1;       // chain child code, sibling's child index
        d4 p[KF][NT];
        double mx = 0.0;
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            d4 f[2][NT];
            // one child's factor tiles: a tip's column from the parked matrix, or NT groups of four steps with b(gq, u) as operand
            auto column = [&](int ch, unsigned st) {
                const unsigned sc = st >= (unsigned) S ? 0u : st;
                const double* row = lds + (size_t) (ch * KF + k) * frag + (size_t) ((sc >> 2) * NT) * 64 + (sc & 3u) * 16 + g;
#pragma unroll
                for (int it = 0; it < NT; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double v = row[it * 64 + 4 * r];
                        f[ch][it][r] = st >= (unsigned) S ? 1.0 : v;
                    }
            };
            auto contract = [&](int ch, auto bval) {
                const double* la = lds + (size_t) (ch * KF + k) * frag + lane;
#pragma unroll
                for (int it = 0; it < NT; ++it) f[ch][it] = (d4) (0.0);
#pragma unroll
                for (int gq = 0; gq < NT; ++gq) {
                    if (4 * gq >= stepsP) break;             // (wave-uniform)
                    double a[4][NT];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int it = 0; it < NT; ++it) a[u][it] = la[(size_t) ((4 * gq + u) * NT + it) * 64];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double bu = bval(gq, u) * ((4 * (4 * gq + u) + g) < S ? 1.0 : 0.0);
#pragma unroll
                        for (int it = 0; it < NT; ++it) f[ch][it] = mbd_mfma_f64_16x16x4(a[u][it], bu, f[ch][it]);
                    }
                }
            };
            // the running result first -- it needs nothing from memory, and the sibling's partials get that much longer to arrive --
            // or (first operation of a chain) child 1: a tip or partials in memory
            if (cc != 0) contract(1 - si, [&](int gq, int u) { return prev[k][gq][u]; });
            else if (op->c1_tip) column(0, as_global(reinterpret_cast<const uint8_t*>(op->c1))[c]);
            else {
                const MBAMD_AS_GLOBAL double* cl = as_global(reinterpret_cast<const double*>(op->c1)) + (size_t) k * S * Ppad + c;
                double first[NT][4];
#pragma unroll
                for (int gq = 0; gq < NT; ++gq)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = 4 * (4 * gq + u) + g;
                        first[gq][u] = cl[(size_t) (j < S ? j : S - 1) * Ppad];
                    }
                contract(0, [&](int gq, int u) { return first[gq][u]; });
            }
            if (si ? op->c2_tip : op->c1_tip) column(si, sibState);
            else contract(si, [&](int gq, int u) { return sib[k][gq][u]; });
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                p[k][it] = f[0][it] * f[1][it];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * it + g + 4 * r < S) mx = fmax(mx, p[k][it][r]);
            }
        }
        const int eStored = storedExp;
        if (o + 1 < oe) fetchSibling(dnxt);                // (every category's sibling values have been used)
        mx = fmax(mx, mbd_shfl_xor(mx, 16));
        mx = fmax(mx, mbd_shfl_xor(mx, 32));
        int e = 0;
        if (op->mode == 1) {
            if (mx > 0.0 && mx < 1.0e300) (void) frexp(mx, &e);
            e = e < -1000 ? -1000 : e;
            if (g == 0) {
                as_global(op->scale)[c] = e;
                if (op->cum != nullptr && e != 0) atomicAdd(op->cum + c, e);
            }
        } else if (op->mode == 2) {
            e = eStored;
        }
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            MBAMD_AS_GLOBAL double* dst = as_global(op->dst) + (size_t) k * S * Ppad + c;
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + g + 4 * r;
                    const double v = e != 0 ? ldexp(p[k][it][r], -e) : p[k][it][r];
                    prev[k][it][r] = v;
                    if (i < S) dst[(size_t) i * Ppad] = v;
                }
        }
        MBAMD_SYNC();                                        // every wave is done with the parked matrices
        dcur = dnxt;
        dnxt = dnn;
    }
}

// Both children compact tips: no contraction, the product of two matrix columns -- a gather.  On the kernel above that is 32 scattered
// loads and 16 stores per wave at two waves per SIMD (its accumulators), 1.2 us per codon operation against 0.5 us of stores; here a
// lane owns states g, g + 4, ... of pattern n (NSL of them per category, KF categories: KF x NSL <= 32 products in registers), the
// same product and the same rescale, eight waves per SIMD.  grid (P_pad / 16, operations).
template <int NSL, int KF>
__global__ void __launch_bounds__(64)
k64_partials_tips(const Op64* __restrict__ ops, int S, int SPAD, int Ppad_)
{
    const MBAMD_AS_CONST Op64* op = as_const(ops) + blockIdx.y;
    const size_t Ppad = (size_t) Ppad_;
    const int lane = (int) threadIdx.x, n = lane & 15, g = lane >> 4;
    const size_t c = (size_t) blockIdx.x * 16 + n;
    if ((size_t) blockIdx.x * 16 + 16 <= (size_t) op->first || (size_t) blockIdx.x * 16 >= (size_t) op->last) return;   // (wave-uniform)
    const bool mine = c >= (size_t) op->first && c < (size_t) op->last;
    const unsigned s1 = as_global(reinterpret_cast<const uint8_t*>(op->c1))[c], s2 = as_global(reinterpret_cast<const uint8_t*>(op->c2))[c];
    const bool gap1 = s1 >= (unsigned) S, gap2 = s2 >= (unsigned) S;
    const MBAMD_AS_GLOBAL double* r1 = as_global(op->m1T) + (size_t) (gap1 ? 0u : s1) * SPAD;
    const MBAMD_AS_GLOBAL double* r2 = as_global(op->m2T) + (size_t) (gap2 ? 0u : s2) * SPAD;
    double p[KF][NSL];
    double mx = 0.0;
#pragma unroll
    for (int k = 0; k < KF; ++k)
#pragma unroll
        for (int q = 0; q < NSL; ++q) {
            const int i = g + 4 * q, ic = i < S ? i : S - 1;
            const double a = r1[(size_t) k * S * SPAD + ic], b = r2[(size_t) k * S * SPAD + ic];
            p[k][q] = i < S ? (gap1 ? 1.0 : a) * (gap2 ? 1.0 : b) : 0.0;
            mx = fmax(mx, p[k][q]);
        }
    mx = fmax(mx, mbd_shfl_xor(mx, 16));
    mx = fmax(mx, mbd_shfl_xor(mx, 32));
    int e = 0;
    if (op->mode == 1) {
        if (mx > 0.0 && mx < 1.0e300) (void) frexp(mx, &e);
        e = e < -1000 ? -1000 : e;
        if (mine && g == 0) {
            as_global(op->scale)[c] = e;
            if (op->cum != nullptr && e != 0) atomicAdd(op->cum + c, e);
        }
    } else if (op->mode == 2) {
        e = as_global(op->scale)[c];
    }
    if (mine) {
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            MBAMD_AS_GLOBAL double* dst = as_global(op->dst) + (size_t) k * S * Ppad + c;
#pragma unroll
            for (int q = 0; q < NSL; ++q) {
                const int i = g + 4 * q;
                if (i < S) dst[(size_t) i * Ppad] = e != 0 ? ldexp(p[k][q], -e) : p[k][q];
            }
        }
    }
}

// The same for 256 patterns per workgroup with both matrices parked in LDS (rows SPAD | 1 doubles apart, so that the lanes' rows fall on
// different banks): the gather above makes 512 L2 requests per wave of 16 patterns -- 1 GB through the L1 miss queues for 244 MB of
// results at codon size, 121 us where the stores alone take 36 (tools/microbench/store_patterns.hip) -- here a workgroup reads its two
// matrices once, coalesced.  lane = pattern: every store is 512 contiguous bytes.  Two passes over LDS (maximum, then products) instead
// of K x S products in registers.  Same products, same rescale.  grid (ceil(P_pad / 256), operations), dynamic LDS 2 K S (SPAD | 1) doubles.
__global__ void __launch_bounds__(256)
k64_partials_tips_lds(const Op64* __restrict__ ops, int S, int SPAD, int K, int Ppad_)
{
    double* lds = mbd_dyn_lds<double>();
    const MBAMD_AS_CONST Op64* op = as_const(ops) + blockIdx.y;
    const size_t Ppad = (size_t) Ppad_;
    const int tid = (int) threadIdx.x;
    const int SL = SPAD | 1, rows = K * S;
    if ((size_t) blockIdx.x * 256 + 256 <= (size_t) op->first || (size_t) blockIdx.x * 256 >= (size_t) op->last) return;   // (workgroup-uniform)
    {   // rows [k][state] of both transposed matrices, 16 loads per thread in flight
        const MBAMD_AS_GLOBAL double* m1 = as_global(op->m1T);
        const MBAMD_AS_GLOBAL double* m2 = as_global(op->m2T);
        const int total = rows * SPAD;                       // elements of one matrix (all categories)
        for (int base = 0; base < total; base += 256 * 8) {
            double t1[8], t2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid, ic = idx < total ? idx : total - 1;
                t1[u] = m1[ic];
                t2[u] = m2[ic];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                if (idx < total) {
                    const int r = idx / SPAD, i = idx - r * SPAD;
                    lds[(size_t) r * SL + i] = t1[u];
                    lds[(size_t) (rows + r) * SL + i] = t2[u];
                }
            }
        }
    }
    MBAMD_SYNC();
    const size_t c = (size_t) blockIdx.x * 256 + tid;
    if (c >= Ppad) return;                                   // (whole waves: P_pad is a multiple of 64; no barrier below)
    const bool mine = c >= (size_t) op->first && c < (size_t) op->last;
    const unsigned s1 = as_global(reinterpret_cast<const uint8_t*>(op->c1))[c], s2 = as_global(reinterpret_cast<const uint8_t*>(op->c2))[c];
    const bool gap1 = s1 >= (unsigned) S, gap2 = s2 >= (unsigned) S;
    const double* r1 = lds + (size_t) (gap1 ? 0u : s1) * SL;
    const double* r2 = lds + (size_t) rows * SL + (size_t) (gap2 ? 0u : s2) * SL;
    const size_t kstep = (size_t) S * SL;                    // a category further
    double mx = 0.0;
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < S; ++i) {
            const double a = r1[k * kstep + i], b = r2[k * kstep + i];
            mx = fmax(mx, (gap1 ? 1.0 : a) * (gap2 ? 1.0 : b));
        }
    int e = 0;
    if (op->mode == 1) {
        if (mx > 0.0 && mx < 1.0e300) (void) frexp(mx, &e);
        e = e < -1000 ? -1000 : e;
        if (mine) {
            as_global(op->scale)[c] = e;
            if (op->cum != nullptr && e != 0) atomicAdd(op->cum + c, e);
        }
    } else if (op->mode == 2) {
        e = as_global(op->scale)[c];
    }
    if (!mine) return;
    MBAMD_AS_GLOBAL double* dst = as_global(op->dst) + c;
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < S; ++i) {
            const double a = r1[k * kstep + i], b = r2[k * kstep + i];
            const double p = (gap1 ? 1.0 : a) * (gap2 ? 1.0 : b);
            dst[((size_t) k * S + i) * Ppad] = e != 0 ? ldexp(p, -e) : p;
        }
}


// The same with the rescale fused (K == KF categories, S <= IB: all K x S results of a pattern stay in registers): one pass
// over HBM instead of three.  Instantiated for four states and the default four gamma categories (at 20 states the K x S
// results need all 256 VGPRs and the fused kernel is no faster: measured, dropped).
template <int IB, int KF>
__global__ void __launch_bounds__(64)
k64_partials_fused(const Op64* __restrict__ ops, int S, int SPAD, int Ppad_)
{
    const Op64& op = ops[blockIdx.y];
    const size_t Ppad = (size_t) Ppad_, c = (size_t) blockIdx.x * 64 + threadIdx.x;
    if (c < (size_t) op.first || c >= (size_t) op.last) return;
    int32_t* cumulative = op.cum;
    double out[KF][IB];
    double mx = 0.0;
#pragma unroll
    for (int k = 0; k < KF; ++k) {
        double f2[IB];
        f64_child_factor<IB>(op.c1, op.c1_tip, op.m1T + (size_t) k * S * SPAD, S, SPAD, k, Ppad, c, 0, out[k]);
        f64_child_factor<IB>(op.c2, op.c2_tip, op.m2T + (size_t) k * S * SPAD, S, SPAD, k, Ppad, c, 0, f2);
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            out[k][i] *= f2[i];
            if (i < S) mx = fmax(mx, out[k][i]);
        }
    }
    int e = 0;
    if (op.mode == 1) {
        if (mx > 0.0 && mx < 1.0e300) (void) frexp(mx, &e);
        e = e < -1000 ? -1000 : e;
        op.scale[c] = e;
        if (cumulative != nullptr && e != 0) atomicAdd(cumulative + c, e);
    } else if (op.mode == 2) {
        e = op.scale[c];
    }
#pragma unroll
    for (int k = 0; k < KF; ++k)
#pragma unroll
        for (int i = 0; i < IB; ++i)
            if (i < S) op.dst[((size_t) k * S + i) * Ppad + c] = e != 0 ? ldexp(out[k][i], -e) : out[k][i];
}

// CondLikeScaler_* (reference src/likelihood.c:4939-4988): per-pattern maximum over categories and states, exact
// power-of-two rescale, exponent kept (and added to the cumulative buffer of the call)
__global__ void __launch_bounds__(64)
k64_rescale(const Op64* __restrict__ ops, int S, int K, int Ppad_)
{
    const Op64& op = ops[blockIdx.y];
    if (op.mode == 0) return;
    const size_t Ppad = (size_t) Ppad_, c = (size_t) blockIdx.x * 64 + threadIdx.x;
    if (c < (size_t) op.first || c >= (size_t) op.last) return;
    int32_t* cumulative = op.cum;
    double* dst = op.dst + c;
    const int n = K * S;
    int e = 0;
    if (op.mode == 1) {
        double mx = 0.0;
        for (int r = 0; r < n; ++r) mx = fmax(mx, dst[(size_t) r * Ppad]);
        if (mx > 0.0 && mx < 1.0e300) (void) frexp(mx, &e);
        e = e < -1000 ? -1000 : e;
        op.scale[c] = e;
        if (cumulative != nullptr && e != 0) atomicAdd(cumulative + c, e);
    } else {
        e = op.scale[c];
    }
    if (e != 0)
        for (int r = 0; r < n; ++r) dst[(size_t) r * Ppad] = ldexp(dst[(size_t) r * Ppad], -e);
}


// ---------------------------------------------------------------------------------------------------------------------------
// Four states: the tree walk in fp64 -- ONE launch per operation list instead of one per dependency level.  A WAVE owns PW = 64 / KP
// patterns with all their categories: lane = category * PW + pattern (KP = the category count rounded up to a power of two; the
// lanes of a category beyond the last repeat the last one, bit for bit, and store the same values to the same addresses).  The
// rescaling maximum is per PATTERN (CondLikeScaler_*): the categories of a pattern meet through log2(KP) lane exchanges -- no
// barrier, no LDS exchange, a workgroup is one wave.  All waves interpret one program compiled by the same Walk4Builder as the
// fp32 walks (mbamd_walk4_host.h, register-fed mode: a child is a compact tip, a slot of the wave's LDS, or read from HBM in
// place).  Results are stored once and children the wave produced itself are read back from LDS: HBM sees (almost) only the
// write stream -- the level kernels read every child back.  Arithmetic and its order are those of k64_partials_fused: the two
// paths give the same bits (MBAMD_F64_NO_WALK=1 selects the levels).
// (Round 3's version -- a workgroup of K waves, one per category, the maxima exchanged through LDS behind a barrier per
//  operation, matrices as scalar operands, every load issued where it was used: 2.50 ms per evaluation at 1000 x 50 000, the
//  barrier version with this file's fetch-ahead 1.64 ms; 782 four-wave workgroups also spread unevenly over 256 CUs.)
struct Walk64Entry {             // 32 bytes, one scalar load: INDICES (buffer, matrix, exponent row), the bases are kernel arguments
    uint32_t dst;                // partials buffer written
    uint32_t c1, c2;             // memory child: partials buffer; compact tip: row of the state array; LDS child: unused
    uint32_t m1, m2;             // matrix buffers
    uint32_t scaleR, scaleW;     // exponent rows read (mode 2) / written (mode 1); the instance's scratch row when unused
    uint32_t ctl;                // kind1 | kind2 << 2 | mode << 4 | nop << 6 | slot1 << 8 | slot2 << 16 | keep << 24
};                               //   kind: 0 LDS slot, 1 memory, 2 compact tip; keep: slot the result is also written to, 0xFF none
static_assert(sizeof(Walk64Entry) == 32, "Walk64Entry is one 32-byte scalar load");
struct Walk64Args {
    const Walk64Entry* prog;
    int entries, nslots;
    double* partials;            // [buffer][K][4][Ppad]
    unsigned bufDoubles;         // doubles per buffer (K x 4 x Ppad)
    const uint8_t* states;       // [row][Ppad]
    const double* matricesT;     // transposed copy [K][4][4] of matrix 0; matrix m is matDoubles further
    unsigned matDoubles;
    int32_t* scale;              // [row][Ppad]
    int32_t* cum;                // cumulative row of the list, or nullptr
    int Ppad;
    int scratchRow;              // exponent row nobody reads
    int K;                       // categories (<= KP of the instantiation)
};
__device__ __forceinline__ Walk64Entry w64_load(const MBAMD_AS_CONST Walk64Entry* p)
{
    Walk64Entry e;
    e.dst = p->dst; e.c1 = p->c1; e.c2 = p->c2; e.m1 = p->m1; e.m2 = p->m2; e.scaleR = p->scaleR; e.scaleW = p->scaleW; e.ctl = p->ctl;
    return e;
}
// Vector-memory results return in order behind everything issued before them: a load issued after an entry's stores waits for
// those stores to reach HBM (microseconds under a write stream).  What every entry loads -- the states of its compact tips (a
// byte per lane) and its two 4 x 4 matrices per category -- is therefore fetched ONE ENTRY AHEAD, before the previous entry's
// stores, by loads that every entry issues whatever its children are (an entry without tips reads one harmless, cached byte):
// straight-line code, so the compiler's wait counts are exact and leave the stores in flight.  Children that live in HBM (evicted
// from the LDS slots: 3 of 996 at 500 taxa with four slots, none with five) and stored exponents (SCALE_READ) are loaded where
// they are used and CONSUMED there (a load still pending where branches meet makes the compiler wait for every vector-memory
// instruction, the previous entry's stores included, on all paths).  An entry without a scale buffer writes its zero exponents
// to the instance's scratch row (the same five stores for every entry); the host leaves no no-operation entries in the program;
// the loop is entered after a whole first entry, so that both ways into the loop head end with the same instruction sequence.
__device__ __forceinline__ unsigned w64_fetch_state(const Walk64Args& a, unsigned kind, unsigned row, unsigned c)
{
    const unsigned long idle = (unsigned long) a.prog;
    const unsigned long tip = kind == 2u ? ~0ul : 0ul;
    const unsigned long base = idle + ((((unsigned long) a.states + (unsigned long) row * (unsigned) a.Ppad) - idle) & tip);
    return *reinterpret_cast<const MBAMD_AS_GLOBAL uint8_t*>(base + (c & (unsigned) tip));
}
// The matrices of an entry: 2 children x KP categories x 16 doubles = KP / 2 doubles per lane (element g = t * 64 + lane:
// child g / (16 KP), category (g / 16) % KP, entry g % 16), parked in LDS as [child][category][18] (the two pad doubles keep
// the KP lane groups, which read at the same offset of different categories, on different banks) and read back by every lane
// from ITS category's rows.
template <int KP> struct Walk64Fetched {
    static constexpr int NL = KP >= 2 ? KP / 2 : 1;
    double m[NL];
    unsigned st1, st2;
};
template <int KP>
__device__ __forceinline__ void w64_fetch(const Walk64Args& a, const Walk64Entry& e, unsigned c, int lane, Walk64Fetched<KP>& f)
{
    const unsigned o1 = e.m1 * a.matDoubles * 8u, o2 = e.m2 * a.matDoubles * 8u;           // (below 4 GiB: the host checks)
#pragma unroll
    for (int t = 0; t < Walk64Fetched<KP>::NL; ++t) {
        const int g = (t * 64 + lane) & (32 * KP - 1);                                     // (KP = 1: lanes 32 .. 63 repeat)
        const int child = g / (16 * KP), cat = (g / 16) % KP, el = g % 16;
        const int kc = cat < a.K ? cat : a.K - 1;
        const unsigned off = (child ? o2 : o1) + (unsigned) (kc * 16 + el) * 8u;
        f.m[t] = *reinterpret_cast<const MBAMD_AS_GLOBAL double*>((unsigned long) a.matricesT + off);
    }
    f.st1 = w64_fetch_state(a, e.ctl & 3u, e.c1, c);
    f.st2 = w64_fetch_state(a, (e.ctl >> 2) & 3u, e.c2, c);
}

template <int KP>
__global__ void __launch_bounds__(64, 4)
k64_walk4(Walk64Args a)
{
    constexpr int PW = 64 / KP;                        // patterns per wave
    constexpr int NL = Walk64Fetched<KP>::NL;
    double* const slots = mbd_dyn_lds<double>();      // [slot][4][64] | matrices [2][KP][18]
    double* const mats = slots + (size_t) a.nslots * 4 * 64;
    const unsigned Ppad = (unsigned) a.Ppad;
    const int lane = (int) threadIdx.x & 63, kk = lane / PW;
    const int kc = kk < a.K ? kk : a.K - 1;            // (lanes beyond the last category repeat it)
    const unsigned c = blockIdx.x * (unsigned) PW + (unsigned) (lane % PW);
    const unsigned laneOff = ((unsigned) kc * 4u * Ppad + c) * 8u;        // this lane's (category, pattern) inside a buffer, bytes
    const unsigned scratchOff = (unsigned) a.scratchRow * Ppad * 4u + c * 4u;
    double* const mySlots = slots + lane;              // + slot * 256 + q * 64
    const double* const myMats = mats + kk * 18;       // + child * KP * 18
    int sum = 0;
    const MBAMD_AS_CONST Walk64Entry* cprog = as_const(a.prog);
    const int last = a.entries - 1;
    Walk64Entry cur = w64_load(cprog), nxt = w64_load(cprog + (last > 0 ? 1 : 0));
    // one entry: what it needs from HBM in `in` (fetched during the previous entry), the next entry's fetched into `next`
    auto step = [&](int j, const Walk64Fetched<KP>& in, Walk64Fetched<KP>& next) {
        w64_fetch<KP>(a, nxt, c, lane, next);          // (a harmless repeat of the last entry at the end)
        const unsigned kind1 = cur.ctl & 3u, kind2 = (cur.ctl >> 2) & 3u, mode = (cur.ctl >> 4) & 3u;
        const unsigned slot1 = (cur.ctl >> 8) & 0xFFu, slot2 = (cur.ctl >> 16) & 0xFFu, keep = cur.ctl >> 24;
#pragma unroll
        for (int t = 0; t < NL; ++t) {
            const int g = (t * 64 + lane) & (32 * KP - 1);
            mats[(g / 16) * 18 + g % 16] = in.m[t];
        }
        MBAMD_WAVE_SYNC();
        double out[4], f2[4], mx = 0.0;
        // a compact tip's factor is the row of its state (the gather the level kernels do; the product with an indicator vector
        // would add exact zeros to the same bits), missing data = 1
        auto factor = [&](unsigned kind, unsigned slot, unsigned buf, unsigned st, const double* m, double (&f)[4]) {
            if (kind == 2u) {
                const double* row = m + (st < 4u ? st : 0u) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = st < 4u ? row[i] : 1.0;
                return;
            }
            double v[4];
            if (kind == 0u) {
                const double* sl = mySlots + (size_t) slot * 256;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = sl[q * 64];
            } else {
                const unsigned long base = (unsigned long) a.partials + (unsigned long) buf * a.bufDoubles * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const MBAMD_AS_GLOBAL double*>(base + (unsigned long) q * Ppad * 8 + laneOff);
                MBAMD_CONSUME4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = fma(m[q * 4 + i], v[q], f[i]);
        };
        factor(kind1, slot1, cur.c1, in.st1, myMats, out);
        factor(kind2, slot2, cur.c2, in.st2, myMats + KP * 18, f2);
        MBAMD_WAVE_SYNC();                               // (the next entry's matrices overwrite these)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[i] *= f2[i];
            mx = fmax(mx, out[i]);
        }
        int ex = 0;
        if (mode == 1u) {
            mx = mbd_max_across_groups<PW>(mx);
            if (mx > 0.0 && mx < 1.0e300) (void) frexp(mx, &ex);
            ex = ex < -1000 ? -1000 : ex;
            sum += ex;
        } else if (mode == 2u) {
            ex = *reinterpret_cast<const MBAMD_AS_GLOBAL int32_t*>((unsigned long) a.scale + (unsigned long) cur.scaleR * Ppad * 4 + c * 4u);
            MBAMD_CONSUME1(ex);
        }
        const unsigned long dst = (unsigned long) a.partials + (unsigned long) cur.dst * a.bufDoubles * 8;
        // (category 0's lanes store the pattern's exponent; an entry without a scale buffer: scaleW is the scratch row)
        const unsigned scaleOff = kk == 0 ? cur.scaleW * Ppad * 4u + c * 4u : scratchOff;
        // the descriptor after next: a scalar load that the stores below and the next entry's fetch hide
        cur = nxt;
        nxt = w64_load(cprog + (j + 2 < last ? j + 2 : last));
        double v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = ex != 0 ? ldexp(out[i], -ex) : out[i];
            // (non-temporal: with write-allocate the result stream evicts matrices and programs from L2)
            MBAMD_STORE_NT(v[i], reinterpret_cast<MBAMD_AS_GLOBAL double*>(dst + (unsigned long) i * Ppad * 8 + laneOff));
        }
        *reinterpret_cast<MBAMD_AS_GLOBAL int32_t*>((unsigned long) a.scale + scaleOff) = ex;
        if (keep != 0xFFu) {
            double* sl = mySlots + (size_t) keep * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) sl[i * 64] = v[i];
        }
    };
    Walk64Fetched<KP> A, B;
    w64_fetch<KP>(a, cur, c, lane, A);
    step(0, A, B);
    for (int j = 1; j <= last; j += 2) {
        step(j, B, A);
        if (j + 1 > last) break;
        step(j + 1, A, B);
    }
    if (kk == 0 && a.cum != nullptr && sum != 0) as_global(a.cum)[c] += sum;
}

struct MatrixJob64 {
    double* out;                 // [K][S][S] then transposed [K][S][SPAD]
    double length;
    const double* eig;           // [U | U^-1 | lambda]
    double pad_;
};
__global__ void __launch_bounds__(256)
k64_exponentials(const MatrixJob64* __restrict__ jobs, RatesArg rates, int S, int K, int total, double* __restrict__ ev)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int s = g % S, bk = g / S;
    const int b = bk / K, k = bk % K;
    ev[g] = exp(jobs[b].eig[(size_t) 2 * S * S + s] * jobs[b].length * rates.r[k]);
}
// TiProbs_Gen (reference src/likelihood.c:9498-9545): P_k = U diag(exp(lambda t r_k)) U^-1, negatives clamped to zero
__global__ void __launch_bounds__(256)
k64_matrices(const MatrixJob64* __restrict__ jobs, const double* __restrict__ ev, int S, int SPAD, int K)
{
    const int b = blockIdx.x / K, k = blockIdx.x % K;
    const double* __restrict__ U = jobs[b].eig;
    const double* __restrict__ Ui = jobs[b].eig + (size_t) S * S;
    const double* __restrict__ e = ev + (size_t) blockIdx.x * S;
    double* __restrict__ M = jobs[b].out + (size_t) k * S * S;
    double* __restrict__ MT = jobs[b].out + (size_t) K * S * S + (size_t) k * S * SPAD;
    for (int idx = threadIdx.x; idx < S * S; idx += blockDim.x) {
        const int i = idx / S, j = idx % S;
        double sum = 0.0;
        for (int s = 0; s < S; ++s) sum += U[i * S + s] * e[s] * Ui[s * S + j];
        const double v = sum < 0.0 ? 0.0 : sum;
        M[(size_t) i * S + j] = v;
        MT[(size_t) j * SPAD + i] = v;
    }
}

// the same product on the fp64 matrix cores for 16 <= S <= 64 (one wave per 16 rows, as k_transition_matrices_mfma of the fp32 engine)
template <int NJ>
__global__ void __launch_bounds__(64 * NJ)
k64_matrices_mfma(const MatrixJob64* __restrict__ jobs, const double* __restrict__ ev, int S, int SPAD, int K)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int b = blockIdx.x / K, k = blockIdx.x % K;
    const MBAMD_AS_GLOBAL double* __restrict__ U = as_global(jobs[b].eig);
    const MBAMD_AS_GLOBAL double* __restrict__ Ui = U + (size_t) S * S;
    const MBAMD_AS_GLOBAL double* __restrict__ e = as_global(ev) + (size_t) blockIdx.x * S;
    const int wave = (int) threadIdx.x >> 6, lane = (int) threadIdx.x & 63, li = lane & 15, ls = lane >> 4;
    const int i = 16 * wave + li, ic = i < S ? i : S - 1;
    d4 acc[NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) acc[jt] = (d4) (0.0);
    int jc[NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) jc[jt] = 16 * jt + li < S ? 16 * jt + li : S - 1;
    const int steps = (S + 3) / 4;
    for (int st0 = 0; st0 < steps; st0 += 4) {
        double a[4], bb[4][NJ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = 4 * (st0 + u) + ls, sc = s < S ? s : S - 1;
            a[u] = (s < S && i < S) ? U[(size_t) ic * S + sc] * e[sc] : 0.0;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) bb[u][jt] = Ui[(size_t) sc * S + jc[jt]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) acc[jt] = mbd_mfma_f64_16x16x4(a[u], bb[u][jt], acc[jt]);
    }
    MBAMD_AS_GLOBAL double* __restrict__ M = as_global(jobs[b].out) + (size_t) k * S * S;
    MBAMD_AS_GLOBAL double* __restrict__ MT = as_global(jobs[b].out) + (size_t) K * S * S + (size_t) k * S * SPAD;
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + ls + 4 * r, j = 16 * jt + li;
            if (row < S && j < S) {
                const double v = acc[jt][r] < 0.0 ? 0.0 : acc[jt][r];
                M[(size_t) row * S + j] = v;
                MT[(size_t) j * SPAD + row] = v;
            }
        }
}

// Likelihood_* (reference src/likelihood.c:5764-5917, 6975-7040) with BEAGLE's root / edge semantics; one thread per pattern
struct IntegrateArgs64 {
    const double*  parent[MBAMD_MAX_SUBSETS];
    const void*    child[MBAMD_MAX_SUBSETS];      // nullptr: root integration
    const double*  matrix[MBAMD_MAX_SUBSETS];     // [K][S][S]
    const double*  weights[MBAMD_MAX_SUBSETS];
    const double*  freqs[MBAMD_MAX_SUBSETS];
    const int32_t* cum[MBAMD_MAX_SUBSETS];
    uint8_t        child_tip[MBAMD_MAX_SUBSETS];
    int            count;
};
__global__ void __launch_bounds__(64)
k64_integrate(IntegrateArgs64 a, int S, int K, int first, int last, int Ppad_, const double* __restrict__ pattern_weights,
              double* __restrict__ site, double* __restrict__ wsite)
{
    // patterns [first, last): everything, or one partition; blocks are counted from the 64-pattern block that holds `first`
    const size_t Ppad = (size_t) Ppad_, c = (size_t) (first / 64 + (int) blockIdx.x) * 64 + threadIdx.x;
    double wl = 0.0;
    if (c >= (size_t) first && c < (size_t) last) {
        int emax = -2147483647;
        for (int n = 0; n < a.count; ++n) {
            const int e = a.cum[n] ? a.cum[n][c] : 0;
            emax = e > emax ? e : emax;
        }
        double total = 0.0;
        for (int n = 0; n < a.count; ++n) {
            double like = 0.0;
            for (int k = 0; k < K; ++k) {
                const double* par = a.parent[n] + (size_t) k * S * Ppad + c;
                double cat = 0.0;
                if (a.child[n] == nullptr) {
                    for (int i = 0; i < S; ++i) cat += par[(size_t) i * Ppad] * a.freqs[n][i];
                } else if (a.child_tip[n]) {
                    const unsigned s = reinterpret_cast<const uint8_t*>(a.child[n])[c];
                    for (int i = 0; i < S; ++i) {
                        const double pc = s >= (unsigned) S ? 1.0 : a.matrix[n][((size_t) k * S + i) * S + s];
                        cat += par[(size_t) i * Ppad] * pc * a.freqs[n][i];
                    }
                } else {
                    const double* ch = reinterpret_cast<const double*>(a.child[n]) + (size_t) k * S * Ppad + c;
                    for (int i = 0; i < S; ++i) {
                        const double* row = a.matrix[n] + ((size_t) k * S + i) * S;
                        double acc = 0.0;
                        for (int j = 0; j < S; ++j) acc = fma(row[j], ch[(size_t) j * Ppad], acc);
                        cat += par[(size_t) i * Ppad] * acc * a.freqs[n][i];
                    }
                }
                like += cat * a.weights[n][k];
            }
            const int e = a.cum[n] ? a.cum[n][c] : 0;
            total += ldexp(like, e - emax);
        }
        const double lnl = log(total) + (double) emax * 0.69314718055994530942;
        site[c] = lnl;
        wl = lnl * pattern_weights[c];
    }
    mbd_wave_sum_store(wl, wsite + blockIdx.x);
}

// The same for larger state counts: eight threads per pattern (thread group g takes the from-states i = g, g + 8, ...), their
// partial sums added in a fixed order by the pattern's first thread -- an eighth of the serial chain (61 states, three omega
// classes, one thread per pattern: 78 us for 5 000 patterns).  block = 512: thread = g * 64 + pattern.
__global__ void __launch_bounds__(512)
k64_integrate_wide(IntegrateArgs64 a, int S, int K, int first, int last, int Ppad_, const double* __restrict__ pattern_weights,
                   double* __restrict__ site, double* __restrict__ wsite)
{
    double (*part)[8][64] = reinterpret_cast<double (*)[8][64]>(mbd_dyn_lds<double>());       // [subset][group][pattern]
    const int p = (int) threadIdx.x & 63, g = (int) threadIdx.x >> 6;
    const size_t Ppad = (size_t) Ppad_, c = (size_t) (first / 64 + (int) blockIdx.x) * 64 + p;
    const bool live = c >= (size_t) first && c < (size_t) last;
    for (int n = 0; n < a.count; ++n) {
        double like = 0.0;
        if (live) {
            for (int k = 0; k < K; ++k) {
                const double* par = a.parent[n] + (size_t) k * S * Ppad + c;
                double cat = 0.0;
                if (a.child[n] == nullptr) {
                    for (int i = g; i < S; i += 8) cat += par[(size_t) i * Ppad] * a.freqs[n][i];
                } else if (a.child_tip[n]) {
                    const unsigned s = reinterpret_cast<const uint8_t*>(a.child[n])[c];
                    for (int i = g; i < S; i += 8) {
                        const double pc = s >= (unsigned) S ? 1.0 : a.matrix[n][((size_t) k * S + i) * S + s];
                        cat += par[(size_t) i * Ppad] * pc * a.freqs[n][i];
                    }
                } else {
                    const double* ch = reinterpret_cast<const double*>(a.child[n]) + (size_t) k * S * Ppad + c;
                    for (int i = g; i < S; i += 8) {
                        const double* row = a.matrix[n] + ((size_t) k * S + i) * S;
                        double acc = 0.0;
                        for (int j = 0; j < S; ++j) acc = fma(row[j], ch[(size_t) j * Ppad], acc);
                        cat += par[(size_t) i * Ppad] * acc * a.freqs[n][i];
                    }
                }
                like += cat * a.weights[n][k];
            }
        }
        part[n][g][p] = like;
    }
    MBAMD_SYNC();
    if (g != 0) return;
    double wl = 0.0;
    if (live) {
        int emax = -2147483647;
        for (int n = 0; n < a.count; ++n) {
            const int e = a.cum[n] ? a.cum[n][c] : 0;
            emax = e > emax ? e : emax;
        }
        double total = 0.0;
        for (int n = 0; n < a.count; ++n) {
            const double like = ((part[n][0][p] + part[n][1][p]) + (part[n][2][p] + part[n][3][p])) +
                                ((part[n][4][p] + part[n][5][p]) + (part[n][6][p] + part[n][7][p]));
            const int e = a.cum[n] ? a.cum[n][c] : 0;
            total += ldexp(like, e - emax);
        }
        const double lnl = log(total) + (double) emax * 0.69314718055994530942;
        site[c] = lnl;
        wl = lnl * pattern_weights[c];
    }
    mbd_wave_sum_store(wl, wsite + blockIdx.x);
}

__global__ void __launch_bounds__(256)
k64_scale_accumulate(const int32_t* const* __restrict__ src, int count, int sign, int first, int last, int32_t* __restrict__ cum)
{
    const int c = first + (int) (blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= last) return;
    int s = 0;
    for (int i = 0; i < count; ++i) s += src[i][c];
    cum[c] += sign * s;
}

// ------------------------------------------------------------------------------------------------------------------
class Engine64 {
public:
    int device = 0, tipCount = 0, nBuffers = 0, S = 0, SPAD = 0, IB = 4, P = 0, Ppad = 0, K = 1, nEigen = 0, nMatrices = 0, nScale = 0;
    hipStream_t stream{};
    bool live = false;
    double* d_partials = nullptr;          // [nBuffers][K*S*Ppad]
    uint8_t* d_states = nullptr;           // [tipCount][Ppad]
    std::vector<char> isTip, valid;
    std::vector<int> stateSlot;            // buffer -> row of d_states (MrBayes numbers its tips i * nCijkParts, reference src/mbbeagle.c:148)
    int slotsUsed = 0;
    double* d_matrices = nullptr;          // [nMatrices][K*S*S + K*S*SPAD]
    double* d_eigen = nullptr;             // [nEigen][2*S*S + S]
    double* d_freqs = nullptr;             // [nEigen][S]
    double* d_weights = nullptr;           // [nEigen][K]
    double* d_pweights = nullptr;          // [Ppad]
    int32_t* d_scale = nullptr;            // [nScale][Ppad]
    double* d_site = nullptr;              // [Ppad]
    double* d_sums = nullptr;              // [partitions of a call][Ppad/64]
    size_t sumsCap = 0;
    double* d_ev = nullptr;
    size_t evCap = 0;
    void* d_stage = nullptr;
    size_t stageCap = 0;
    // small host -> device transfers (operation lists, matrix jobs, weights, frequencies) go through a ring: the bytes are copied into
    // pinned host memory, from there asynchronously into the device ring's slot of the same offset, and a slot is written again only
    // after the ring wrapped -- one stream synchronisation per RING_BYTES instead of one per call (a codon M3 evaluation made ten).
    static constexpr size_t RING_BYTES = 4u << 20, RING_MAX_ITEM = 256u << 10;
    uint8_t* h_ring = nullptr;
    uint8_t* d_ring = nullptr;
    size_t ringPos = 0;
    double* h_sums = nullptr;              // pinned: the block sums of a log-likelihood call
    size_t hSumsCap = 0;
    std::vector<RatesArg> rateSets;
    bool haveSite = false;
    std::vector<std::pair<int, int>> parts;       // v3: [first, last) of every pattern partition (empty: none were set)
    // four-state tree walk (k64_walk4): the program compiler, its latest program (re-used when the same list comes again)
    Walk4Builder walkBuilder;
    Walk4Template walkTemplate;
    std::vector<int> walkKey;
    std::vector<Walk4Op> walkOps;
    std::vector<Walk64Entry> walkProg;
    // operation lists waiting to run (see updatePartialsEx)
    struct QueuedOp { BeagleOperation op; int partition, cum; char tip1, tip2; };
    std::vector<QueuedOp> queue;
    std::vector<char> queuedScale;          // [nScale]: an exponent buffer some queued operation reads, writes or accumulates into
    uint64_t walkLaunches = 0, levelLaunches = 0;
    bool walkAlways = false, walkVerbose = false;   // MBAMD_F64_WALK_ALWAYS, MBAMD_VERBOSE
    bool noMfma = false;                           // MBAMD_F64_NO_MFMA: the vector-ALU level kernels for 16..64 states too
    bool walkOff = false;                          // MBAMD_F64_NO_WALK (read when the instance is created): level kernels only
    size_t bufDoubles = 0, matDoubles = 0, eigDoubles = 0;

    ~Engine64() { destroy(); }

    static int blockOf(int S) { return S <= 4 ? 4 : S <= 8 ? 8 : S <= 16 ? 16 : S <= 20 ? 20 : 32; }

    int create(int tips, int partialsBuffers, int compactBuffers, int states, int patterns, int eigens, int matrices, int cats, int scales, int dev)
    {
        device = dev; tipCount = tips; nBuffers = partialsBuffers + compactBuffers; S = states; P = patterns; Ppad = round_up(patterns, 64);
        K = cats; nEigen = eigens; nMatrices = matrices; nScale = scales;
        IB = blockOf(S);
        walkOff = std::getenv("MBAMD_F64_NO_WALK") != nullptr;
        walkAlways = std::getenv("MBAMD_F64_WALK_ALWAYS") != nullptr;
        walkVerbose = std::getenv("MBAMD_VERBOSE") != nullptr;
        noMfma = std::getenv("MBAMD_F64_NO_MFMA") != nullptr;
        SPAD = (S + IB - 1) / IB * IB;
        bufDoubles = (size_t) K * S * Ppad;
        matDoubles = (size_t) K * S * S + (size_t) K * S * SPAD;
        eigDoubles = (size_t) 2 * S * S + S;
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        live = true;
        HIP_TRY(hipMalloc(&d_partials, std::max<size_t>(1, (size_t) nBuffers * bufDoubles) * sizeof(double)));
        HIP_TRY(hipMalloc(&d_states, std::max<size_t>(1, (size_t) tipCount * Ppad)));
        HIP_TRY(hipMalloc(&d_matrices, std::max<size_t>(1, (size_t) nMatrices * matDoubles) * sizeof(double)));
        HIP_TRY(hipMemsetAsync(d_matrices, 0, std::max<size_t>(1, (size_t) nMatrices * matDoubles) * sizeof(double), stream));
        HIP_TRY(hipMalloc(&d_eigen, std::max<size_t>(1, (size_t) nEigen * eigDoubles) * sizeof(double)));
        HIP_TRY(hipMalloc(&d_freqs, std::max<size_t>(1, (size_t) nEigen * S) * sizeof(double)));
        HIP_TRY(hipMalloc(&d_weights, std::max<size_t>(1, (size_t) nEigen * K) * sizeof(double)));
        HIP_TRY(hipMalloc(&d_pweights, (size_t) Ppad * sizeof(double)));
        HIP_TRY(hipMemsetAsync(d_pweights, 0, (size_t) Ppad * sizeof(double), stream));
        // (one row more than the caller's: the scratch row the walk's entries without a scale buffer write their zero exponents to)
        HIP_TRY(hipMalloc(&d_scale, (size_t) (nScale + 1) * Ppad * sizeof(int32_t)));
        HIP_TRY(hipMemsetAsync(d_scale, 0, (size_t) (nScale + 1) * Ppad * sizeof(int32_t), stream));
        HIP_TRY(hipMalloc(&d_site, (size_t) Ppad * sizeof(double)));
        sumsCap = (size_t) (Ppad / 64);
        HIP_TRY(hipMalloc(&d_sums, sumsCap * sizeof(double)));
        isTip.assign((size_t) nBuffers, 0);
        stateSlot.assign((size_t) nBuffers, -1);
        valid.assign((size_t) nBuffers, 0);
        rateSets.assign(1, RatesArg());
        for (int k = 0; k < MBAMD_MAX_RATES; ++k) rateSets[0].r[k] = 1.0;
        std::vector<double> ones((size_t) Ppad, 0.0);
        for (int c = 0; c < P; ++c) ones[c] = 1.0;
        HIP_TRY(hipMemcpyAsync(d_pweights, ones.data(), (size_t) Ppad * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return BEAGLE_SUCCESS;
    }
    void destroy()
    {
        if (!live) return;
        (void) hipSetDevice(device);
        (void) hipStreamSynchronize(stream);
        void* all[] = {d_partials, d_states, d_matrices, d_eigen, d_freqs, d_weights, d_pweights, d_scale, d_site, d_sums, d_ev, d_stage};
        for (void* p : all)
            if (p) (void) hipFree(p);
        if (d_ring) (void) hipFree(d_ring);
        if (h_ring) (void) hipHostFree(h_ring);
        if (h_sums) (void) hipHostFree(h_sums);
        (void) hipStreamDestroy(stream);
        live = false;
    }
    // a ring slot holding `bytes` from src (host and device side), or nullptr when the item is too large for the ring
    int ringPut(const void* src, size_t bytes, uint8_t** hostSlot, uint8_t** devSlot)
    {
        *hostSlot = *devSlot = nullptr;
        if (bytes > RING_MAX_ITEM || std::getenv("MBAMD_F64_NO_RING") != nullptr) return BEAGLE_SUCCESS;
        if (h_ring == nullptr) {
            HIP_TRY(hipHostMalloc((void**) &h_ring, RING_BYTES, hipHostMallocDefault));
            HIP_TRY(hipMalloc((void**) &d_ring, RING_BYTES));
        }
        const size_t need = (bytes + 255) & ~(size_t) 255;
        if (ringPos + need > RING_BYTES) {
            HIP_TRY(hipStreamSynchronize(stream));              // every slot's copy and its readers are behind us
            ringPos = 0;
        }
        *hostSlot = h_ring + ringPos;
        *devSlot = d_ring + ringPos;
        ringPos += need;
        std::memcpy(*hostSlot, src, bytes);
        return BEAGLE_SUCCESS;
    }
    int stage(const void* src, size_t bytes, void** out)
    {
        {
            uint8_t *hs, *ds;
            int rc = ringPut(src, bytes, &hs, &ds);
            if (rc) return rc;
            if (ds != nullptr) {
                HIP_TRY(hipMemcpyAsync(ds, hs, bytes, hipMemcpyHostToDevice, stream));
                *out = ds;
                return BEAGLE_SUCCESS;
            }
        }
        HIP_TRY(hipStreamSynchronize(stream));                  // (the staging buffer is re-used: wait for its last reader)
        if (bytes > stageCap) {
            if (d_stage) (void) hipFree(d_stage);
            d_stage = nullptr;
            stageCap = std::max(bytes * 2, (size_t) 65536);
            HIP_TRY(hipMalloc(&d_stage, stageCap));
        }
        HIP_TRY(hipMemcpyAsync(d_stage, src, bytes, hipMemcpyHostToDevice, stream));
        *out = d_stage;
        return BEAGLE_SUCCESS;
    }
    int upload(void* dst, const void* src, size_t bytes)
    {
        {
            uint8_t *hs, *ds;
            int rc = ringPut(src, bytes, &hs, &ds);
            if (rc) return rc;
            if (hs != nullptr) {                                 // (stream order keeps it behind the earlier readers of dst)
                HIP_TRY(hipMemcpyAsync(dst, hs, bytes, hipMemcpyHostToDevice, stream));
                return BEAGLE_SUCCESS;
            }
        }
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return BEAGLE_SUCCESS;
    }
    double* partialsPtr(int b) const { return d_partials + (size_t) b * bufDoubles; }
    uint8_t* statesPtr(int b) const { return d_states + (size_t) stateSlot[b] * Ppad; }
    double* matrixPtr(int m) const { return d_matrices + (size_t) m * matDoubles; }

    int setTipStates(int tip, const int* states)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (tip < 0 || tip >= nBuffers) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetTipStates: tip index");
        if (stateSlot[tip] < 0) {
            if (slotsUsed >= tipCount) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetTipStates: more compact buffers than tips");
            stateSlot[tip] = slotsUsed++;
        }
        std::vector<uint8_t> h((size_t) Ppad, (uint8_t) S);
        for (int c = 0; c < P; ++c) h[c] = (uint8_t) ((states[c] < 0 || states[c] >= S) ? S : states[c]);
        isTip[tip] = 1;
        valid[tip] = 1;
        return upload(statesPtr(tip), h.data(), (size_t) Ppad);
    }
    // in: [K][P][S] (withCategories) or [P][S] replicated over the categories
    int setPartials(int idx, const double* in, bool withCategories)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (idx < 0 || idx >= nBuffers) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetPartials: buffer index");
        std::vector<double> h(bufDoubles, 0.0);
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < P; ++c)
                for (int i = 0; i < S; ++i)
                    h[((size_t) k * S + i) * Ppad + c] = in[((size_t) (withCategories ? k : 0) * P + c) * S + i];
        isTip[idx] = 0;
        valid[idx] = 1;
        return upload(partialsPtr(idx), h.data(), bufDoubles * sizeof(double));
    }
    int getPartials(int idx, double* out)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (idx < 0 || idx >= nBuffers || !valid[idx]) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleGetPartials: buffer index");
        if (isTip[idx]) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleGetPartials: a compact (tip state) buffer");
        std::vector<double> h(bufDoubles);
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(h.data(), partialsPtr(idx), bufDoubles * sizeof(double), hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < P; ++c)
                for (int i = 0; i < S; ++i) out[((size_t) k * P + c) * S + i] = h[((size_t) k * S + i) * Ppad + c];
        return BEAGLE_SUCCESS;
    }
    int setEigen(int idx, const double* U, const double* Ui, const double* lam)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (idx < 0 || idx >= nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetEigenDecomposition: eigen index");
        std::vector<double> h(eigDoubles);
        std::memcpy(h.data(), U, sizeof(double) * S * S);
        std::memcpy(h.data() + (size_t) S * S, Ui, sizeof(double) * S * S);
        std::memcpy(h.data() + (size_t) 2 * S * S, lam, sizeof(double) * S);
        return upload(d_eigen + (size_t) idx * eigDoubles, h.data(), eigDoubles * sizeof(double));
    }
    // host mirrors of the frequencies / weights on the device (NaN = nothing sent yet): true when `v` is what the device already holds
    std::vector<double> hostFreqs, hostWeights;
    static void forget(std::vector<double>& mirror, size_t at, size_t n)
    {
        for (size_t i = 0; i < n && at + i < mirror.size(); ++i) mirror[at + i] = std::numeric_limits<double>::quiet_NaN();
    }
    static bool sameAsLast(std::vector<double>& mirror, size_t total, size_t at, const double* v, size_t n)
    {
        if (mirror.size() != total) mirror.assign(total, std::numeric_limits<double>::quiet_NaN());
        if (std::memcmp(mirror.data() + at, v, n * sizeof(double)) == 0) return true;      // (bitwise: a NaN pattern never equals user data by accident of -0.0 / 0.0)
        std::memcpy(mirror.data() + at, v, n * sizeof(double));
        return false;
    }
    int setFreqs(int idx, const double* f)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (idx < 0 || idx >= nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetStateFrequencies: index");
        // (MrBayes sets the frequencies and category weights of every eigen part before every evaluation: unchanged values are not sent again)
        if (sameAsLast(hostFreqs, (size_t) nEigen * S, (size_t) idx * S, f, (size_t) S)) return BEAGLE_SUCCESS;
        const int rc = upload(d_freqs + (size_t) idx * S, f, (size_t) S * sizeof(double));
        if (rc) forget(hostFreqs, (size_t) idx * S, (size_t) S);       // (the device kept the old vector: the next identical call must send again)
        return rc;
    }
    int setWeights(int idx, const double* w)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (idx < 0 || idx >= nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetCategoryWeights: index");
        if (sameAsLast(hostWeights, (size_t) nEigen * K, (size_t) idx * K, w, (size_t) K)) return BEAGLE_SUCCESS;
        const int rc = upload(d_weights + (size_t) idx * K, w, (size_t) K * sizeof(double));
        if (rc) forget(hostWeights, (size_t) idx * K, (size_t) K);
        return rc;
    }
    int setRates(int index, const double* r)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (index < 0 || index > 65535) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "category rates: index");
        if ((size_t) index >= rateSets.size()) rateSets.resize((size_t) index + 1, rateSets[0]);
        for (int k = 0; k < K; ++k) rateSets[index].r[k] = r[k];
        return BEAGLE_SUCCESS;
    }
    int setPatternWeights(const double* w)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        std::vector<double> h((size_t) Ppad, 0.0);
        std::memcpy(h.data(), w, (size_t) P * sizeof(double));
        return upload(d_pweights, h.data(), (size_t) Ppad * sizeof(double));
    }
    void launchMatrices(const MatrixJob64* dj, int count)
    {
        if (S >= 16 && S <= 64 && !noMfma) {
            const unsigned grid = (unsigned) (count * K);
            switch ((S + 15) / 16) {
                case 1: MBAMD_LAUNCH_BARRIER(k64_matrices_mfma<1>, grid, 64, 0, stream, dj, (const double*) d_ev, S, SPAD, K); break;
                case 2: MBAMD_LAUNCH_BARRIER(k64_matrices_mfma<2>, grid, 128, 0, stream, dj, (const double*) d_ev, S, SPAD, K); break;
                case 3: MBAMD_LAUNCH_BARRIER(k64_matrices_mfma<3>, grid, 192, 0, stream, dj, (const double*) d_ev, S, SPAD, K); break;
                default: MBAMD_LAUNCH_BARRIER(k64_matrices_mfma<4>, grid, 256, 0, stream, dj, (const double*) d_ev, S, SPAD, K); break;
            }
            return;
        }
        MBAMD_LAUNCH(k64_matrices, (unsigned) (count * K), 256, 0, stream, dj, (const double*) d_ev, S, SPAD, K);
    }
    // Matrix updates are queued like operation lists: MrBayes updates a codon model's eigen parts one call each (src/mbbeagle.c:1475-1486),
    // and three launches of 200 matrices fill the chip worse than one of 600.  Every other entry point flushes (flushQueue); a second
    // update of a queued matrix, or another category-rate vector, flushes first.
    std::vector<MatrixJob64> matQueue;
    std::vector<char> matQueued;               // per matrix buffer: an update is in the queue
    int matQueueRate = -1;
    int flushMatrices()
    {
        if (matQueue.empty()) return BEAGLE_SUCCESS;
        std::vector<MatrixJob64> jobs;
        jobs.swap(matQueue);
        std::fill(matQueued.begin(), matQueued.end(), 0);
        const int count = (int) jobs.size();
        void* dj = nullptr;
        int rc = stage(jobs.data(), jobs.size() * sizeof(MatrixJob64), &dj);
        if (rc) return rc;
        const size_t need = (size_t) count * K * S;
        if (need > evCap) {
            HIP_TRY(hipStreamSynchronize(stream));
            if (d_ev) (void) hipFree(d_ev);
            d_ev = nullptr;
            evCap = need * 2;
            HIP_TRY(hipMalloc(&d_ev, evCap * sizeof(double)));
        }
        const int total = count * K * S;
        MBAMD_LAUNCH(k64_exponentials, (unsigned) ((total + 255) / 256), 256, 0, stream, (const MatrixJob64*) dj, rateSets[matQueueRate], S, K, total, d_ev);
        launchMatrices((const MatrixJob64*) dj, count);
        HIP_TRY(hipGetLastError());
        return BEAGLE_SUCCESS;
    }
    int updateMatrices(int eigenIdx, int rateIdx, const int* prob, const double* lengths, int count)
    {
        if (!queue.empty()) { const int rcq = flushQueue(); if (rcq) return rcq; }        // (queued operations read the matrices as they are now)
        if (count <= 0) return BEAGLE_SUCCESS;
        if (eigenIdx < 0 || eigenIdx >= nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdateTransitionMatrices: eigen index");
        if (rateIdx < 0 || (size_t) rateIdx >= rateSets.size()) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdateTransitionMatrices: rate index");
        for (int i = 0; i < count; ++i)
            if (prob[i] < 0 || prob[i] >= nMatrices) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdateTransitionMatrices: matrix index");
        if (!matQueue.empty() && (matQueueRate != rateIdx || matQueue.size() + (size_t) count > 60000)) { const int rc = flushMatrices(); if (rc) return rc; }
        if (matQueued.size() != (size_t) nMatrices) matQueued.assign((size_t) nMatrices, 0);
        matQueueRate = rateIdx;
        for (int i = 0; i < count; ++i) {
            if (matQueued[(size_t) prob[i]]) { const int rc = flushMatrices(); if (rc) return rc; }
            matQueued[(size_t) prob[i]] = 1;
            matQueue.push_back({matrixPtr(prob[i]), lengths[i], d_eigen + (size_t) eigenIdx * eigDoubles, 0.0});
        }
        return std::getenv("MBAMD_F64_NO_MATRIX_QUEUE") != nullptr ? flushMatrices() : BEAGLE_SUCCESS;
    }
    // v3: an eigen-system and a category-rate vector per matrix; one launch per run of equal rate vectors
    int updateMatricesMulti(const int* eigenIdx, const int* rateIdx, const int* prob, const double* lengths, int count)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        int i = 0;
        while (i < count) {
            int j = i + 1;
            while (j < count && eigenIdx[j] == eigenIdx[i] && rateIdx[j] == rateIdx[i]) ++j;
            const int rc = updateMatrices(eigenIdx[i], rateIdx[i], prob + i, lengths + i, j - i);
            if (rc) return rc;
            i = j;
        }
        return BEAGLE_SUCCESS;
    }
    // in: [K][S][S] row = from-state (BEAGLE's order)
    int setMatrix(int idx, const double* m)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (idx < 0 || idx >= nMatrices) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetTransitionMatrix: matrix index");
        std::vector<double> h(matDoubles, 0.0);
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < S; ++i)
                for (int j = 0; j < S; ++j) {
                    const double v = m[((size_t) k * S + i) * S + j];
                    h[((size_t) k * S + i) * S + j] = v;
                    h[(size_t) K * S * S + ((size_t) k * S + j) * SPAD + i] = v;
                }
        return upload(matrixPtr(idx), h.data(), matDoubles * sizeof(double));
    }
    int getMatrix(int idx, double* out)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (idx < 0 || idx >= nMatrices) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleGetTransitionMatrix: matrix index");
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(out, matrixPtr(idx), (size_t) K * S * S * sizeof(double), hipMemcpyDeviceToHost));
        return BEAGLE_SUCCESS;
    }

    template <int IB_> void launchPartials(const Op64* ops, int n)
    {
        MBAMD_LAUNCH(k64_partials<IB_>, dim3((unsigned) (Ppad / 64), (unsigned) n, (unsigned) (K * (SPAD / IB_))), 64, 0, stream, ops, S, SPAD, K, Ppad);
    }

    // The walk serves what MrBayes sends for nucleotides: four states, up to eight categories, no pattern partitions, one
    // cumulative buffer for the whole list, no buffer hazards inside the list.  Returns 1 when the list is not of that kind.
    template <int KP> void launchWalk(const Walk64Args& wa)
    {
        auto kern = k64_walk4<KP>;
        const size_t lds = ((size_t) wa.nslots * 4 * 64 + (size_t) 2 * KP * 18) * sizeof(double);
        static char raised[64] = {0};                // per device (and per KP: a static of this template instance)
        if (device >= 0 && device < 64 && !raised[device]) {
            if (hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void) hipGetLastError();
            raised[device] = 1;
        }
        MBAMD_LAUNCH_BARRIER(kern, (unsigned) (Ppad / (64 / KP)), 64, lds, stream, wa);
    }
    int tryWalk4(const QueuedOp* q, int n)
    {
        // (the kernel addresses with 32-bit lane offsets: a plane of partials and the whole exponent array below 4 GiB)
        if (walkOff || S != 4 || K > 8 || !parts.empty() || n < 2 || (bufDoubles >> 29) != 0 || (((size_t) nScale + 1) * Ppad >> 30) != 0 || ((size_t) nMatrices * matDoubles >> 29) != 0) return 1;
        // the walk is one latency chain per wave: it wins when there are enough waves (break-even about 1.2 per SIMD) and on short lists (a
        // root-ward path: one launch instead of one per operation); mid-sized full evaluations stay on the level kernels
        // (measured: profiles/r03_f64_walk.txt).  MBAMD_F64_WALK_ALWAYS=1: every eligible list.
        if (!walkAlways && n > 64 && (long) (Ppad / 64) * K < 1200) return 1;
        std::unique_ptr<StatTimer> st_(new StatTimer(ST_PLAN));      // (MBAMD_STATS: the host side of the walk, up to the upload)
        std::vector<Walk4Op>& wops = walkOps;
        wops.clear();
        std::vector<char> written((size_t) nBuffers, 0), readB((size_t) nBuffers, 0), sc((size_t) std::max(nScale, 1), 0);
        for (int i = 0; i < n; ++i) {
            const BeagleOperation& o = q[i].op;                                       // (indices were checked when the list was queued)
            if (q[i].partition >= 0 || q[i].cum != q[0].cum) return 1;
            const int d = o.destinationPartials, c1 = o.child1Partials, c2 = o.child2Partials;
            const int sw = o.destinationScaleWrite, sr = o.destinationScaleRead;
            if (written[d] || readB[d]) return 1;                                     // buffer hazards: levels
            if (sw != BEAGLE_OP_NONE && sc[sw]) return 1;
            if (sw == BEAGLE_OP_NONE && sr != BEAGLE_OP_NONE && sc[sr] == 2) return 1;
            if (q[0].cum != BEAGLE_OP_NONE && (sw == q[0].cum || sr == q[0].cum)) return 1;
            Walk4Op w;
            w.dst = d; w.c1 = c1; w.c2 = c2; w.m1 = o.child1TransitionMatrix; w.m2 = o.child2TransitionMatrix;
            w.tip1 = q[i].tip1;
            w.tip2 = q[i].tip2;
            w.scaleWrite = sw != BEAGLE_OP_NONE ? sw : -1;
            w.scaleRead = (sw == BEAGLE_OP_NONE && sr != BEAGLE_OP_NONE) ? sr : -1;
            written[d] = 1; readB[c1] = 1; readB[c2] = 1;
            if (sw != BEAGLE_OP_NONE) sc[sw] = 2; else if (sr != BEAGLE_OP_NONE && !sc[sr]) sc[sr] = 1;
            wops.push_back(w);
        }
        const int cumIdx = q[0].cum;
        // launch geometry: a wave owns 64 / KP patterns (KP = K rounded up to a power of two) and is its own workgroup; every wave
        // resident at once where the chip allows, the LDS of a CU split between the waves it hosts; a slot holds one node's
        // 4 x 64 doubles of the wave
        const int KP = K <= 1 ? 1 : (K <= 2 ? 2 : (K <= 4 ? 4 : 8));
        const int slotBytes = 4 * 64 * (int) sizeof(double);
        const long waves = Ppad / (64 / KP);
        const int perCU = (int) std::min(16L, std::max(1L, (waves + 255) / 256));
        const int fixedBytes = 2 * KP * 18 * (int) sizeof(double) + 64;      // the parked matrices (+ allocation granularity)
        int nslots = std::max(2, std::min(24, ((160 * 1024) / perCU - fixedBytes) / slotBytes));
        if (const char* e = std::getenv("MBAMD_F64_WALK_SLOTS")) nslots = std::max(2, std::min((160 * 1024 - fixedBytes) / slotBytes, std::atoi(e)));
        // structure key: who produces whose child, which children are tips (the indices only fill the program)
        std::vector<int> key;
        key.reserve((size_t) n * 3 + 2);
        key.push_back(n); key.push_back(nslots);
        {
            std::vector<int> writer((size_t) nBuffers, -1);
            for (int o = 0; o < n; ++o) {
                key.push_back(wops[o].tip1 ? -1 : writer[wops[o].c1]);
                key.push_back(wops[o].tip2 ? -1 : writer[wops[o].c2]);
                key.push_back((int) wops[o].tip1 | ((int) wops[o].tip2 << 1));
                writer[wops[o].dst] = o;
            }
        }
        if (key != walkKey) {
            Walk4Builder& b = walkBuilder;
            b.maxW = 1; b.maxSlots = nslots; b.maxSlots1 = nslots; b.prefetchDistance = 0; b.memSlots = false;
            b.leadNops = 0; b.unroll = 1; b.tailNops = 0; b.forward = false; b.smallPhase = 1 << 30;
            if (!b.build(wops, walkTemplate)) { walkKey.clear(); return 1; }
            walkKey = key;
        }
        const Walk4Template& t = walkTemplate;
        if (t.W != 1) return 1;
        walkProg.assign((size_t) t.entries, Walk64Entry());
        const unsigned scratch = (unsigned) nScale;                 // the extra exponent row: what entries without a scale buffer "write"
        for (int i = 0; i < t.entries; ++i) {
            const Walk4Template::Entry& te = t.prog[i];
            Walk64Entry& e = walkProg[i];
            std::memset(&e, 0, sizeof e);
            e.scaleR = e.scaleW = scratch;
            unsigned kind1 = 0, kind2 = 0, slot1 = 0, slot2 = 0;
            if (te.op < 0) { e.ctl = 1u << 6; continue; }                     // (dropped below)
            const Walk4Op& w = wops[te.op];
            e.dst = (uint32_t) w.dst; e.m1 = (uint32_t) w.m1; e.m2 = (uint32_t) w.m2;
            if (w.tip1) { kind1 = 2; e.c1 = (uint32_t) stateSlot[w.c1]; }
            else if (te.c1slot == 0xFF) { kind1 = 1; e.c1 = (uint32_t) w.c1; }
            else { kind1 = 0; slot1 = te.c1slot; }
            if (w.tip2) { kind2 = 2; e.c2 = (uint32_t) stateSlot[w.c2]; }
            else if (te.c2slot == 0xFF) { kind2 = 1; e.c2 = (uint32_t) w.c2; }
            else { kind2 = 0; slot2 = te.c2slot; }
            const unsigned mode = w.scaleWrite >= 0 ? 1u : (w.scaleRead >= 0 ? 2u : 0u);
            if (mode == 1u) e.scaleW = (uint32_t) w.scaleWrite;
            if (mode == 2u) e.scaleR = (uint32_t) w.scaleRead;
            e.ctl = kind1 | (kind2 << 2) | (mode << 4) | (slot1 << 8) | (slot2 << 16) | ((unsigned) te.dslot << 24);
        }
        // the kernel's entries all compute (see k64_walk4): drop the no-operation entries of the builder (one wave: they order nothing)
        walkProg.erase(std::remove_if(walkProg.begin(), walkProg.end(), [](const Walk64Entry& e) { return ((e.ctl >> 6) & 1u) != 0; }), walkProg.end());
        if (walkProg.empty()) return 1;
        {   // the kernel fetches entry i's memory children while entry i-1 runs: their producer must be entry i-2 or earlier
            std::vector<int> writtenAt((size_t) nBuffers, -1000);
            for (size_t i = 0; i < walkProg.size(); ++i) {
                const Walk64Entry& e = walkProg[i];
                if (((e.ctl & 3u) == 1u && writtenAt[e.c1] >= (int) i - 1) || (((e.ctl >> 2) & 3u) == 1u && writtenAt[e.c2] >= (int) i - 1)) { walkKey.clear(); return 1; }
                writtenAt[e.dst] = (int) i;
            }
        }
        if (walkVerbose) {
            int mem = 0, tips = 0;
            for (const Walk64Entry& e : walkProg) {
                mem += ((e.ctl & 3u) == 1u) + (((e.ctl >> 2) & 3u) == 1u);
                tips += ((e.ctl & 3u) == 2u) + (((e.ctl >> 2) & 3u) == 2u);
            }
            std::fprintf(stderr, "[mbamd] fp64 walk: %zu entries, %d slots, children: %d compact tips, %d from memory, %zu from LDS\n", walkProg.size(),
                         t.nslots, tips, mem, 2 * walkProg.size() - (size_t) tips - (size_t) mem);
        }
        st_.reset();
        void* dv = nullptr;
        int rc = stage(walkProg.data(), walkProg.size() * sizeof(Walk64Entry), &dv);
        if (rc) return rc;
        Walk64Args wa;
        wa.prog = static_cast<const Walk64Entry*>(dv);
        wa.entries = (int) walkProg.size(); wa.nslots = t.nslots;
        wa.partials = d_partials; wa.bufDoubles = (unsigned) bufDoubles;
        wa.states = d_states;
        wa.matricesT = d_matrices + (size_t) K * S * S; wa.matDoubles = (unsigned) matDoubles;
        wa.scale = d_scale;
        wa.cum = cumIdx != BEAGLE_OP_NONE ? d_scale + (size_t) cumIdx * Ppad : nullptr;
        wa.Ppad = (int) Ppad;
        wa.scratchRow = nScale;
        wa.K = K;
        switch (KP) {
            case 1: launchWalk<1>(wa); break;
            case 2: launchWalk<2>(wa); break;
            case 4: launchWalk<4>(wa); break;
            default: launchWalk<8>(wa); break;
        }
        HIP_TRY(hipGetLastError());
        walkLaunches++;
        return BEAGLE_SUCCESS;
    }
    // One launch per dependency level: an operation goes one level above the last operation that wrote a buffer it reads,
    // read or wrote the buffer it writes, or touched its scale buffer.
    int partitionRange(int partition, int* first, int* last, const char* what) const
    {
        if (partition < 0) { *first = 0; *last = Ppad; return BEAGLE_SUCCESS; }
        if (parts.empty() ? partition != 0 : partition >= (int) parts.size()) return fail(BEAGLE_ERROR_OUT_OF_RANGE, what, "partition index");
        if (parts.empty()) { *first = 0; *last = Ppad; return BEAGLE_SUCCESS; }
        *first = parts[partition].first;
        *last = parts[partition].second;
        return BEAGLE_SUCCESS;
    }
    int setPartitions(int count, const int* ids)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        std::vector<std::pair<int, int>> r;
        for (int c = 0; c < P; ++c) {
            const int p = ids[c];
            if (p < 0 || p >= count) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetPatternPartitions: partition index");
            if ((int) r.size() == p) r.emplace_back(c, c + 1);
            else if ((int) r.size() == p + 1 && r[p].second == c) r[p].second++;
            else return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleSetPatternPartitions: partitions must be contiguous, increasing pattern ranges");
        }
        if ((int) r.size() != count) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetPatternPartitions: empty partition");
        parts = r;
        return BEAGLE_SUCCESS;
    }
    int updatePartials(const BeagleOperation* ops, int n, int cumIdx)
    {
        std::vector<int> part((size_t) std::max(n, 0), -1), cum((size_t) std::max(n, 0), cumIdx);
        return updatePartialsEx(ops, sizeof(BeagleOperation), n, part.data(), cum.data());
    }
    // `stride` bytes between operations (BeagleOperation or BeagleOperationByPartition: the first seven ints are the same);
    // partition[i] < 0: all patterns.
    // Lists are QUEUED, not run: MrBayes submits one list per eigen-system part of a codon model (reference src/mbbeagle.c:1088-1104,
    // with at most a beagleRemoveScaleFactors of the next part's buffers in between), and a launch per dependency level of every
    // list is three times the launches of one launch per level of all of them (codon M3 100 x 5 000: 51 -> 17 launches, 1.37 -> 1.0 ms per evaluation).
    // Every other call of the engine runs the queue first (flushQueue); everything that can fail is checked here, when the list comes.
    int updatePartialsEx(const void* opsRaw, size_t stride, int n, const int* partition, const int* cumOf)
    {
        if (n <= 0) return BEAGLE_SUCCESS;
        // (the matrix updates are complete when the first operation list comes: the device computes them while the host queues and sorts the lists)
        if (!matQueue.empty()) { const int rcm = flushMatrices(); if (rcm) return rcm; }
        const size_t mark = queue.size();
        const int np = std::max<int>(1, (int) parts.size());
        for (int i = 0; i < n; ++i) {
            const BeagleOperation& o = *reinterpret_cast<const BeagleOperation*>(static_cast<const char*>(opsRaw) + (size_t) i * stride);
            const int cumIdx = cumOf[i];
            int rc = BEAGLE_SUCCESS;
            int first = 0, last = Ppad;
            const int d = o.destinationPartials, c1 = o.child1Partials, c2 = o.child2Partials;
            const int sw = o.destinationScaleWrite, sr = o.destinationScaleRead;
            if (cumIdx != BEAGLE_OP_NONE && (cumIdx < 0 || cumIdx >= nScale)) rc = fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: cumulative scale index");
            else if ((rc = partitionRange(partition[i], &first, &last, "beagleUpdatePartialsByPartition")) != BEAGLE_SUCCESS) { }
            else if (d < 0 || d >= nBuffers || c1 < 0 || c1 >= nBuffers || c2 < 0 || c2 >= nBuffers)
                rc = fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: buffer index");
            else if (o.child1TransitionMatrix < 0 || o.child1TransitionMatrix >= nMatrices || o.child2TransitionMatrix < 0 || o.child2TransitionMatrix >= nMatrices)
                rc = fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: matrix index");
            else if (!valid[c1] || !valid[c2]) rc = fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: a child buffer was never written");
            else if (isTip[d]) rc = fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: destination is a compact tip buffer");
            else if ((sw != BEAGLE_OP_NONE && (sw < 0 || sw >= nScale)) || (sr != BEAGLE_OP_NONE && (sr < 0 || sr >= nScale)))
                rc = fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: scale index");
            if (rc != BEAGLE_SUCCESS) { queue.resize(mark); return rc; }          // (nothing of a rejected list runs)
            QueuedOp e;
            e.op = o; e.partition = partition[i]; e.cum = cumIdx;
            e.tip1 = isTip[c1]; e.tip2 = isTip[c2];                                // (what the children are NOW: a later operation may overwrite a tip buffer)
            queue.push_back(e);
            valid[d] = 1;
            isTip[d] = 0;
            if (queuedScale.size() != (size_t) std::max(nScale, 1)) queuedScale.assign((size_t) std::max(nScale, 1), 0);
            if (sw != BEAGLE_OP_NONE) queuedScale[sw] = 1;
            if (sr != BEAGLE_OP_NONE) queuedScale[sr] = 1;
            if (cumIdx != BEAGLE_OP_NONE) queuedScale[cumIdx] = 1;
        }
        (void) np;
        return BEAGLE_SUCCESS;
    }
    // run what updatePartials queued; called first by every other entry point
    // (an error returned from here means the queued lists were DROPPED -- the queue is empty afterwards, whichever entry point
    //  reported it: the client resubmits them, as after any failed beagleUpdatePartials)
    int flushQueue()
    {
        { const int rcm = flushMatrices(); if (rcm) return rcm; }
        if (queue.empty()) return BEAGLE_SUCCESS;
        std::vector<QueuedOp> q;
        q.swap(queue);
        std::fill(queuedScale.begin(), queuedScale.end(), 0);
        return runPartials(q.data(), (int) q.size());
    }
    // Hazards are tracked per (buffer, partition): the same buffer index in two partitions is two disjoint pattern ranges.
    int runPartials(const QueuedOp* qd, int n)
    {
        {
            int rcw = tryWalk4(qd, n);
            if (rcw != 1) return rcw;                      // (1: not a list for the walk -- the level path below takes it)
        }
        const int np = std::max<int>(1, (int) parts.size());
        std::vector<int> level((size_t) n, 0), lastTouchBuf((size_t) nBuffers * np, -1), lastWriteBuf((size_t) nBuffers * np, -1),
            lastTouchScale((size_t) std::max(nScale, 1) * np, -1);
        int nLevels = 0;
        std::vector<Op64> h((size_t) n);
        for (int i = 0; i < n; ++i) {
            const BeagleOperation& o = qd[i].op;
            const int cumIdx = qd[i].cum;
            int first = 0, last = Ppad;
            int rcp = partitionRange(qd[i].partition, &first, &last, "beagleUpdatePartialsByPartition");
            if (rcp) return rcp;
            const int p0 = qd[i].partition < 0 ? 0 : std::min(qd[i].partition, np - 1), p1 = qd[i].partition < 0 ? np : p0 + 1;
            const int d = o.destinationPartials, c1 = o.child1Partials, c2 = o.child2Partials;
            const int sw = o.destinationScaleWrite, sr = o.destinationScaleRead;
            const int sc = sw != BEAGLE_OP_NONE ? sw : sr;
            int lv = 0;
            for (int q = p0; q < p1; ++q) {
                lv = std::max(lv, std::max(std::max(lastWriteBuf[(size_t) c1 * np + q], lastWriteBuf[(size_t) c2 * np + q]), lastTouchBuf[(size_t) d * np + q]) + 1);
                if (sc != BEAGLE_OP_NONE) lv = std::max(lv, lastTouchScale[(size_t) sc * np + q] + 1);
                // (lists of several calls run as one: two operations adding to the same cumulative buffer in one launch are atomic adds)
            }
            level[i] = lv;
            nLevels = std::max(nLevels, lv + 1);
            for (int q = p0; q < p1; ++q) {
                lastWriteBuf[(size_t) d * np + q] = lv;
                lastTouchBuf[(size_t) d * np + q] = std::max(lastTouchBuf[(size_t) d * np + q], lv);
                lastTouchBuf[(size_t) c1 * np + q] = std::max(lastTouchBuf[(size_t) c1 * np + q], lv);
                lastTouchBuf[(size_t) c2 * np + q] = std::max(lastTouchBuf[(size_t) c2 * np + q], lv);
                if (sc != BEAGLE_OP_NONE) lastTouchScale[(size_t) sc * np + q] = lv;
            }
            Op64& q = h[i];
            q.dst = partialsPtr(d);
            q.c1 = qd[i].tip1 ? (const void*) statesPtr(c1) : (const void*) partialsPtr(c1);
            q.c2 = qd[i].tip2 ? (const void*) statesPtr(c2) : (const void*) partialsPtr(c2);
            q.c1_tip = qd[i].tip1;
            q.c2_tip = qd[i].tip2;
            q.m1T = matrixPtr(o.child1TransitionMatrix) + (size_t) K * S * S;
            q.m2T = matrixPtr(o.child2TransitionMatrix) + (size_t) K * S * S;
            q.mode = sw != BEAGLE_OP_NONE ? 1 : sr != BEAGLE_OP_NONE ? 2 : 0;
            q.scale = sc != BEAGLE_OP_NONE ? d_scale + (size_t) sc * Ppad : nullptr;
            q.cum = cumIdx != BEAGLE_OP_NONE ? d_scale + (size_t) cumIdx * Ppad : nullptr;
            q.first = first;
            q.last = last;
            q.pad_ = 0;
        }
        // operations sorted by level (stable), one contiguous run per level
        std::vector<int> order((size_t) n), start((size_t) nLevels + 1, 0);
        for (int i = 0; i < n; ++i) start[(size_t) level[i] + 1]++;
        for (int l = 0; l < nLevels; ++l) start[(size_t) l + 1] += start[l];
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (int i = 0; i < n; ++i) order[(size_t) fill[level[i]]++] = i;
        // (within a level the operations on two compact tips first: they have a kernel of their own)
        std::vector<int> tipsOf((size_t) nLevels, 0);
        for (int l = 0; l < nLevels; ++l) {
            auto mid = std::stable_partition(order.begin() + start[l], order.begin() + start[(size_t) l + 1],
                                             [&](int i) { return h[(size_t) i].c1_tip && h[(size_t) i].c2_tip; });
            tipsOf[l] = (int) (mid - (order.begin() + start[l]));
        }
        std::vector<Op64> sorted((size_t) n);
        for (int i = 0; i < n; ++i) sorted[i] = h[order[i]];
        // A list that is nothing but chains (the root-ward path of a move; one chain per eigen part): one launch of k64_partials_chain
        {
            const int NTr = (S + 15) / 16;
            const size_t ldsBytes = (size_t) 2 * K * ((((S + 3) / 4) + 3) & ~3) * NTr * 64 * sizeof(double);
            if (n >= 2 && S > 16 && S <= 64 && !noMfma && parts.empty() && K >= 1 && K <= 4 && NTr * K <= 8 && ldsBytes <= 65536 &&
                std::getenv("MBAMD_F64_UNFUSED") == nullptr && std::getenv("MBAMD_F64_MFMA_NO_LDS") == nullptr && std::getenv("MBAMD_F64_NO_CHAIN") == nullptr) {
                std::vector<int> root((size_t) n);
                for (int i = 0; i < n; ++i) root[i] = i;
                auto find = [&](int x) { while (root[x] != x) x = root[x] = root[root[x]]; return x; };
                std::unordered_map<const void*, int> owner;
                for (int i = 0; i < n; ++i) {
                    const Op64& q = sorted[(size_t) i];
                    // (the cumulative buffer an operation adds to is a key like its scale buffer: two operations that meet in one --
                    //  one adding atomically, the other storing or reading it as its scale buffer -- must not run as independent chains)
                    const void* keys[5] = {q.dst, q.c1_tip ? nullptr : q.c1, q.c2_tip ? nullptr : q.c2, q.mode != 0 ? (const void*) q.scale : nullptr, (const void*) q.cum};
                    for (const void* key : keys) {
                        if (key == nullptr) continue;
                        auto it = owner.find(key);
                        if (it == owner.end()) owner.emplace(key, i);
                        else { const int a = find(i), b = find(it->second); if (a != b) root[std::max(a, b)] = std::min(a, b); }
                    }
                }
                std::vector<int> chainOf((size_t) n, -1);
                std::vector<std::vector<int>> members;
                for (int i = 0; i < n; ++i) {                 // (`sorted` is level-major: a chain's members come in dependency order)
                    const int r = find(i);
                    if (chainOf[r] < 0) { chainOf[r] = (int) members.size(); members.emplace_back(); }
                    members[(size_t) chainOf[r]].push_back(i);
                }
                bool ok = true;
                std::vector<Op64> chained;
                std::vector<int> chainStart;
                chained.reserve((size_t) n);
                for (const auto& mem : members) {
                    chainStart.push_back((int) chained.size());
                    const double* last = nullptr;
                    std::vector<const void*> written;
                    for (size_t m = 0; m < mem.size() && ok; ++m) {
                        Op64 q = sorted[(size_t) mem[m]];
                        const bool one = !q.c1_tip && q.c1 == (const void*) last, two = !q.c2_tip && q.c2 == (const void*) last;
                        if (m == 0) q.pad_ = 0;
                        else if (one != two) q.pad_ = one ? 1 : 2;
                        else ok = false;                     // not the previous result (or both children are): not a chain
                        // (the other child must come from outside this launch, the scale buffer must not be one written earlier in it)
                        const void* other = m == 0 ? nullptr : (one ? (q.c2_tip ? nullptr : q.c2) : (q.c1_tip ? nullptr : q.c1));
                        for (const void* w : written) ok = ok && w != other && w != (const void*) q.dst && (q.mode != 2 || w != (const void*) q.scale);
                        written.push_back(q.dst);
                        if (q.mode == 1) written.push_back(q.scale);
                        last = q.dst;
                        chained.push_back(q);
                    }
                }
                chainStart.push_back((int) chained.size());
                if (ok) {
                    void *dt = nullptr, *dc = nullptr;
                    int rct = stage(chained.data(), chained.size() * sizeof(Op64), &dt);
                    if (rct) return rct;
                    rct = stage(chainStart.data(), chainStart.size() * sizeof(int), &dc);
                    if (rct) return rct;
                    const dim3 cgrid((unsigned) (Ppad / 64), (unsigned) members.size());
                    const Op64* dto = static_cast<const Op64*>(dt);
                    const int* dco = static_cast<const int*>(dc);
#define MBAMD_F64_CHAIN_CASE(NT_, KF_) MBAMD_LAUNCH_BARRIER((k64_partials_chain<NT_, KF_>), cgrid, 256, ldsBytes, stream, dto, dco, S, SPAD, Ppad)
                    switch (NTr * 8 + K) {
                        case 2 * 8 + 1: MBAMD_F64_CHAIN_CASE(2, 1); break;
                        case 2 * 8 + 2: MBAMD_F64_CHAIN_CASE(2, 2); break;
                        case 2 * 8 + 3: MBAMD_F64_CHAIN_CASE(2, 3); break;
                        case 2 * 8 + 4: MBAMD_F64_CHAIN_CASE(2, 4); break;
                        case 3 * 8 + 1: MBAMD_F64_CHAIN_CASE(3, 1); break;
                        case 4 * 8 + 1: MBAMD_F64_CHAIN_CASE(4, 1); break;
                        default: ok = false; break;
                    }
#undef MBAMD_F64_CHAIN_CASE
                    if (ok) {
                        levelLaunches++;
                        HIP_TRY(hipGetLastError());
                        return BEAGLE_SUCCESS;
                    }
                }
            }
        }
        void* dv = nullptr;
        int rc = stage(sorted.data(), sorted.size() * sizeof(Op64), &dv);
        if (rc) return rc;
        const Op64* dops = static_cast<const Op64*>(dv);
        const bool fused = K == 4 && IB == 4 && S <= IB && std::getenv("MBAMD_F64_UNFUSED") == nullptr;
        for (int l = 0; l < nLevels; ++l) {
            const int first = start[l], cnt = start[(size_t) l + 1] - first;
            if (cnt <= 0) continue;
            levelLaunches++;
            if (fused) {
                const dim3 grid((unsigned) (Ppad / 64), (unsigned) cnt);
                auto kern = k64_partials_fused<4, 4>;
                MBAMD_LAUNCH(kern, grid, 64, 0, stream, dops + first, S, SPAD, Ppad);
                continue;
            }
            if (S >= 16 && S <= 64 && !noMfma) {
                const int NTr = (S + 15) / 16;
                const bool fuse = K >= 1 && K <= 4 && NTr * K <= 8 && std::getenv("MBAMD_F64_UNFUSED") == nullptr;   // all K categories' tiles in registers
                // operations on two compact tips: the gather kernel (fused rescale only, K x ceil(S / 4) <= 32 products per lane)
                const int NSL = (S + 3) / 4 <= 5 ? 5 : (S + 3) / 4 <= 8 ? 8 : 16;
                int ntt = (fuse && NSL * K <= 32 && std::getenv("MBAMD_F64_NO_TIPS_KERNEL") == nullptr) ? tipsOf[l] : 0;
                const size_t tipsLds = (size_t) 2 * K * S * (SPAD | 1) * sizeof(double);
                if (ntt > 0 && tipsLds <= 65536 && std::getenv("MBAMD_F64_TIPS_NO_LDS") == nullptr) {
                    const dim3 tgrid((unsigned) ((Ppad + 255) / 256), (unsigned) ntt);
                    MBAMD_LAUNCH_BARRIER(k64_partials_tips_lds, tgrid, 256, tipsLds, stream, dops + first, S, SPAD, K, Ppad);
                } else if (ntt > 0) {
                    const dim3 tgrid((unsigned) (Ppad / 16), (unsigned) ntt);
#define MBAMD_F64_TIPS_CASE(NSL_, KF_) MBAMD_LAUNCH_BARRIER((k64_partials_tips<NSL_, KF_>), tgrid, 64, 0, stream, dops + first, S, SPAD, Ppad)
                    switch (NSL * 8 + K) {
                        case 5 * 8 + 1: MBAMD_F64_TIPS_CASE(5, 1); break;
                        case 5 * 8 + 2: MBAMD_F64_TIPS_CASE(5, 2); break;
                        case 5 * 8 + 3: MBAMD_F64_TIPS_CASE(5, 3); break;
                        case 5 * 8 + 4: MBAMD_F64_TIPS_CASE(5, 4); break;
                        case 8 * 8 + 1: MBAMD_F64_TIPS_CASE(8, 1); break;
                        case 8 * 8 + 2: MBAMD_F64_TIPS_CASE(8, 2); break;
                        case 8 * 8 + 3: MBAMD_F64_TIPS_CASE(8, 3); break;
                        case 8 * 8 + 4: MBAMD_F64_TIPS_CASE(8, 4); break;
                        case 16 * 8 + 1: MBAMD_F64_TIPS_CASE(16, 1); break;
                        case 16 * 8 + 2: MBAMD_F64_TIPS_CASE(16, 2); break;
                        default: ntt = 0; break;
                    }
#undef MBAMD_F64_TIPS_CASE
                }
                if (ntt == cnt) continue;
                const dim3 grid((unsigned) (Ppad / 16), (unsigned) (cnt - ntt), (unsigned) (fuse ? 1 : K));
                // (matrices through LDS, four waves per workgroup, when both fit into 64 KiB)
                const size_t ldsBytes = (size_t) 2 * (fuse ? K : 1) * ((((S + 3) / 4) + 3) & ~3) * NTr * 64 * sizeof(double);
                // (codon M3 0.90 -> 0.81 ms per evaluation; at 20 states x 4 categories the matrices are 5 KiB each and stay in the L1: 1.19 -> 1.16 ms
                //  with four waves per workgroup since all eight are parked in ONE batch of loads -- a batch per category was 1.38)
                const bool viaLds = NTr >= 2 && ldsBytes <= 65536 && std::getenv("MBAMD_F64_MFMA_NO_LDS") == nullptr;
                // (eight waves per workgroup where that still gives every CU two workgroups; beyond 32 states: with four categories' accumulators
                //  the 128 registers of four waves per SIMD mean spills, 1.55 ms)
                const bool wide = NTr >= 3 && (size_t) ((Ppad + 127) / 128) * (size_t) (cnt - ntt) >= 512 && std::getenv("MBAMD_F64_MFMA_4WAVES") == nullptr;
                const dim3 lgrid((unsigned) (wide ? (Ppad + 127) / 128 : Ppad / 64), (unsigned) (cnt - ntt), (unsigned) (fuse ? 1 : K));
#define MBAMD_F64_MFMA_CASE(NT_, KF_) do { \
                    if (viaLds && wide) MBAMD_LAUNCH_BARRIER((k64_partials_mfma_lds<NT_, KF_, 8>), lgrid, 512, ldsBytes, stream, dops + first + ntt, S, SPAD, Ppad); \
                    else if (viaLds) MBAMD_LAUNCH_BARRIER((k64_partials_mfma_lds<NT_, KF_, 4>), lgrid, 256, ldsBytes, stream, dops + first + ntt, S, SPAD, Ppad); \
                    else MBAMD_LAUNCH_BARRIER((k64_partials_mfma<NT_, KF_>), grid, 64, 0, stream, dops + first + ntt, S, SPAD, Ppad); } while (0)
                const int key = NTr * 8 + (fuse ? K : 0);
                switch (key) {
                    case 1 * 8 + 0: MBAMD_F64_MFMA_CASE(1, 0); break;
                    case 1 * 8 + 1: MBAMD_F64_MFMA_CASE(1, 1); break;
                    case 1 * 8 + 2: MBAMD_F64_MFMA_CASE(1, 2); break;
                    case 1 * 8 + 3: MBAMD_F64_MFMA_CASE(1, 3); break;
                    case 1 * 8 + 4: MBAMD_F64_MFMA_CASE(1, 4); break;
                    case 2 * 8 + 0: MBAMD_F64_MFMA_CASE(2, 0); break;
                    case 2 * 8 + 1: MBAMD_F64_MFMA_CASE(2, 1); break;
                    case 2 * 8 + 2: MBAMD_F64_MFMA_CASE(2, 2); break;
                    case 2 * 8 + 3: MBAMD_F64_MFMA_CASE(2, 3); break;
                    case 2 * 8 + 4: MBAMD_F64_MFMA_CASE(2, 4); break;
                    case 3 * 8 + 0: MBAMD_F64_MFMA_CASE(3, 0); break;
                    case 3 * 8 + 1: MBAMD_F64_MFMA_CASE(3, 1); break;
                    case 3 * 8 + 2: MBAMD_F64_MFMA_CASE(3, 2); break;
                    case 4 * 8 + 0: MBAMD_F64_MFMA_CASE(4, 0); break;
                    case 4 * 8 + 1: MBAMD_F64_MFMA_CASE(4, 1); break;
                    default: MBAMD_F64_MFMA_CASE(4, 2); break;          // (4 * 8 + 2)
                }
#undef MBAMD_F64_MFMA_CASE
                if (fuse) continue;
            } else
            switch (IB) {
                case 4: launchPartials<4>(dops + first, cnt); break;
                case 8: launchPartials<8>(dops + first, cnt); break;
                case 16: launchPartials<16>(dops + first, cnt); break;
                case 20: launchPartials<20>(dops + first, cnt); break;
                default: launchPartials<32>(dops + first, cnt); break;
            }
            bool anyScale = false;
            for (int i = first; i < first + cnt; ++i) anyScale |= sorted[i].mode != 0;
            if (anyScale)
                MBAMD_LAUNCH(k64_rescale, dim3((unsigned) (Ppad / 64), (unsigned) cnt), 64, 0, stream, dops + first, S, K, Ppad);
        }
        HIP_TRY(hipGetLastError());
        return BEAGLE_SUCCESS;
    }

    int resetScale(int idx, int partition = -1)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (idx < 0 || idx >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleResetScaleFactors: index");
        int first = 0, last = Ppad;
        int rc = partitionRange(partition, &first, &last, "beagleResetScaleFactorsByPartition");
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(d_scale + (size_t) idx * Ppad + first, 0, (size_t) (last - first) * sizeof(int32_t), stream));
        return BEAGLE_SUCCESS;
    }
    int accumulateScale(const int* idx, int count, int cumIdx, int sign, int partition = -1)
    {
        {   // (between the lists of a codon model's parts MrBayes removes the NEXT part's scale factors from ITS cumulative buffer:
            //  buffers no queued operation touches -- that may run ahead of the queue)
            bool touches = cumIdx >= 0 && cumIdx < (int) queuedScale.size() && queuedScale[cumIdx];
            for (int i = 0; i < count && !touches; ++i) touches = idx[i] >= 0 && idx[i] < (int) queuedScale.size() && queuedScale[idx[i]];
            if (touches || queuedScale.empty()) { const int rcq = flushQueue(); if (rcq) return rcq; }
        }
        if (count <= 0) return BEAGLE_SUCCESS;
        int first = 0, last = Ppad;
        int rcp = partitionRange(partition, &first, &last, "scale factors by partition");
        if (rcp) return rcp;
        if (cumIdx < 0 || cumIdx >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "scale factors: cumulative index");
        std::vector<const int32_t*> src((size_t) count);
        for (int i = 0; i < count; ++i) {
            if (idx[i] < 0 || idx[i] >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "scale factors: index");
            src[i] = d_scale + (size_t) idx[i] * Ppad;
        }
        void* dv = nullptr;
        int rc = stage(src.data(), src.size() * sizeof(const int32_t*), &dv);
        if (rc) return rc;
        MBAMD_LAUNCH(k64_scale_accumulate, (unsigned) ((last - first + 255) / 256), 256, 0, stream, (const int32_t* const*) dv, count, sign, first, last, d_scale + (size_t) cumIdx * Ppad);
        HIP_TRY(hipGetLastError());
        return BEAGLE_SUCCESS;
    }
    int copyScale(int dst, int src)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (dst < 0 || dst >= nScale || src < 0 || src >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleCopyScaleFactors: index");
        HIP_TRY(hipMemcpyAsync(d_scale + (size_t) dst * Ppad, d_scale + (size_t) src * Ppad, (size_t) Ppad * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
        return BEAGLE_SUCCESS;
    }
    int getScaleExponents(int idx, int* out)            // [K][P]: every category row the same
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (idx < 0 || idx >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "scale factors: index");
        std::vector<int32_t> h((size_t) Ppad);
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(h.data(), d_scale + (size_t) idx * Ppad, (size_t) Ppad * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < P; ++c) out[(size_t) k * P + c] = h[c];
        return BEAGLE_SUCCESS;
    }
    int getScaleFactors(int idx, double* out)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (idx < 0 || idx >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleGetScaleFactors: index");
        std::vector<int32_t> h((size_t) Ppad);
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(h.data(), d_scale + (size_t) idx * Ppad, (size_t) Ppad * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int c = 0; c < P; ++c) out[c] = h[c] * 0.69314718055994530942;
        return BEAGLE_SUCCESS;
    }

    // index arrays are [count][partitionCount] when `partitions` is given (reference src/mbbeagle.c:2781-2800), else [count]
    int logLikelihoods(const int* parent, const int* child, const int* prob, const int* wIdx, const int* fIdx, const int* cumIdx, int count,
                       double* out, const int* partitions = nullptr, int partitionCount = 1, double* outByPartition = nullptr)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (count < 1 || count > MBAMD_MAX_SUBSETS) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "log-likelihood: more than 8 subsets");
        const int pc = partitions ? partitionCount : 1;
        const int nblocks = Ppad / 64;
        const size_t nsums = (size_t) nblocks * pc;
        if (nsums > hSumsCap) {
            if (h_sums) (void) hipHostFree(h_sums);
            h_sums = nullptr;
            hSumsCap = 0;
            HIP_TRY(hipHostMalloc((void**) &h_sums, nsums * sizeof(double), hipHostMallocDefault));
            hSumsCap = nsums;
        }
        double* const h = h_sums;
        std::vector<int> blocksOf((size_t) pc);
        if ((size_t) nblocks * pc > sumsCap) {
            HIP_TRY(hipStreamSynchronize(stream));
            if (d_sums) (void) hipFree(d_sums);
            d_sums = nullptr;
            sumsCap = (size_t) nblocks * pc;
            HIP_TRY(hipMalloc(&d_sums, sumsCap * sizeof(double)));
        }
        for (int d = 0; d < pc; ++d) {
            int first = 0, last = P;
            if (partitions) {
                int rc = partitionRange(partitions[d], &first, &last, "log-likelihood by partition");
                if (rc) return rc;
                last = std::min(last, P);
            }
            IntegrateArgs64 a;
            std::memset(&a, 0, sizeof a);
            a.count = count;
            for (int n = 0; n < count; ++n) {
                const int j = n * pc + d;
                if (parent[j] < 0 || parent[j] >= nBuffers || !valid[parent[j]] || isTip[parent[j]])
                    return fail(BEAGLE_ERROR_OUT_OF_RANGE, "log-likelihood: parent buffer");
                a.parent[n] = partialsPtr(parent[j]);
                if (child) {
                    const int ci = child[j];
                    if (ci < 0 || ci >= nBuffers || !valid[ci] || prob[j] < 0 || prob[j] >= nMatrices)
                        return fail(BEAGLE_ERROR_OUT_OF_RANGE, "edge log-likelihood: child buffer / matrix");
                    a.child[n] = isTip[ci] ? (const void*) statesPtr(ci) : (const void*) partialsPtr(ci);
                    a.child_tip[n] = (uint8_t) isTip[ci];
                    a.matrix[n] = matrixPtr(prob[j]);
                }
                if (wIdx[j] < 0 || wIdx[j] >= nEigen || fIdx[j] < 0 || fIdx[j] >= nEigen)
                    return fail(BEAGLE_ERROR_OUT_OF_RANGE, "log-likelihood: weights / frequencies index");
                a.weights[n] = d_weights + (size_t) wIdx[j] * K;
                a.freqs[n] = d_freqs + (size_t) fIdx[j] * S;
                if (cumIdx && cumIdx[j] != BEAGLE_OP_NONE) {
                    if (cumIdx[j] < 0 || cumIdx[j] >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "log-likelihood: cumulative scale index");
                    a.cum[n] = d_scale + (size_t) cumIdx[j] * Ppad;
                }
            }
            blocksOf[d] = (last + 63) / 64 - first / 64;
            if (blocksOf[d] <= 0) continue;
            if (S >= 16)
                MBAMD_LAUNCH_BARRIER(k64_integrate_wide, (unsigned) blocksOf[d], 512, (size_t) a.count * 8 * 64 * sizeof(double), stream, a, S, K, first, last, Ppad, (const double*) d_pweights, d_site,
                                     d_sums + (size_t) d * nblocks);
            else
                MBAMD_LAUNCH(k64_integrate, (unsigned) blocksOf[d], 64, 0, stream, a, S, K, first, last, Ppad, (const double*) d_pweights, d_site,
                             d_sums + (size_t) d * nblocks);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h, d_sums, nsums * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        double total = 0.0;
        for (int d = 0; d < pc; ++d) {
            double s = 0.0;
            for (int i = 0; i < blocksOf[d]; ++i) s += h[(size_t) d * nblocks + i];
            if (outByPartition) outByPartition[d] = s;
            total += s;
        }
        haveSite = true;
        if (out) *out = total;
        if (!(total == total) || total > 1.79e308 || total < -1.79e308) return BEAGLE_ERROR_FLOATING_POINT;
        return BEAGLE_SUCCESS;
    }
    int getSites(double* out)
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        if (!haveSite) return fail(BEAGLE_ERROR_GENERAL, "beagleGetSiteLogLikelihoods: no log-likelihood was calculated");
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(out, d_site, (size_t) P * sizeof(double), hipMemcpyDeviceToHost));
        return BEAGLE_SUCCESS;
    }
    int synchronize()
    {
        { const int rcq = flushQueue(); if (rcq) return rcq; }
        HIP_TRY(hipStreamSynchronize(stream));
        return BEAGLE_SUCCESS;
    }
};

}  // namespace mbamd
#endif
