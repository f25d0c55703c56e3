// mbamd_kernels_mfma.h -- matrix-core (MFMA) kernels of the general-state path: 20-state amino-acid
// and 61-state codon models (any 5 <= S <= 64).  Written against the device primitives of <mbamd_dev_base.h> /
// <mbamd_dev_walkg_kernel.h>: the TEST-ONLY host emulation compiles and runs these same kernels (an MFMA there is a wave-wide exchange).
//
// Replaces CondLikeDown_Gen[_SSE] / CondLikeDown_NY98[_SSE] + CondLikeScaler_Gen / _NY98
// (reference src/likelihood.c:204-588, 1575-1900, 4939-5070, 5413-5545).
//
// One conditional-likelihood update is, per rate/omega category k, two dense contractions
//     F_c[i][p] = sum_j P_c,k[i][j] * cl_c[k][j][p]        (c = left, right child; p = site pattern)
// followed by the element-wise product F_1 .* F_2 and the per-pattern rescale.  Each contraction
// is mapped on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD):
//     A (32 x 2)  = transition matrix tile   P[32*it + (lane&31)][2t + (lane>>5)]
//     B (2 x 32)  = conditional likelihoods  cl[2t + (lane>>5)][c0 + (lane&31)]
//     D (32 x 32) = F[i][p]: lane holds p = c0 + (lane&31), rows i = 32*it + (r&3) + 8*(r>>2) + 4*(lane>>5)
// The B operand is exactly one coalesced dword load from the tile-major partials layout
// [P_pad/32][K][S][32] (gen_index, mbamd_kernels.h): rows 2t and 2t+1 of a 32-pattern tile are 64
// consecutive floats, one contiguous 256-byte access per wave instruction -- no LDS staging, no transposes.
// The A operand comes pre-packed in MFMA lane order from the transition-matrix kernels (one coalesced
// 256-byte load per MFMA, L1/L2 resident: a matrix set is 10-48 KiB).  A compact tip child needs
// no contraction at all: its factor is column `state` of P, gathered with dwordx4 loads.
// The per-pattern maximum over (k, i) -- the reference's separate CondLikeScaler pass -- is fused
// (register reduction + LDS exchange), and the rescaled result is written exactly once.
//
// Kernels (the host picks per operation list, Instance::runGeneric / flushPending in mbamd_engine.cpp):
//   k_partials_tips        operations on two compact tips (a full evaluation's first dependency level)
//   k_partials_mfma_split  one launch per dependency level, one wave per factor tile (the default)
//   k_partials_mfma_spine  narrow lists / trailing single-operation levels: one launch walks them, software-pipelined
//   k_partials_mfma_serial   plain variant of the spine kernel for run-time state counts
//   k_partials_mfma        one wave per (operation, tile): cross-check (MBAMD_MFMA_WHOLE=1)
//   (k_transition_matrices_mfma, P = U exp(L t) U^-1 on the fp64 matrix cores: mbamd_matrices_mfma.h)
//   k_integrate_lnl_wide   root / edge integration, 8 threads per pattern
#ifndef MBAMD_KERNELS_MFMA_H_
#define MBAMD_KERNELS_MFMA_H_

#include "mbamd_kernels.h"

namespace mbamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));


// floats of the MFMA-packed copy of one category's matrix: NT i-tiles x T j-pairs x 64 lanes
__host__ __device__ inline int mfma_tiles(int S) { return (S + 31) / 32; }
__host__ __device__ inline int mfma_pairs(int S) { return (S + 1) / 2; }

// blockIdx -> (pattern block x, operation y) such that a pattern block always lands on the same
// XCD (block b runs on XCD b % 8): the children written by the previous dependency level for these
// patterns were written through the same XCD's L2.
__device__ __forceinline__ bool xcd_aware_block(int gx, int& x, int& y)
{
    const int id = blockIdx.x;
    const int xcd = id & 7, q = id >> 3;
    const int nxb = (gx + 7) >> 3;
    x = (q % nxb) * 8 + xcd;
    y = q / nxb;
    return x < gx;
}

// F[32 x 32] += P_tile[32 x S] * cl[S x 32] as T = ceil(S/2) chained MFMAs.  ALL operand loads are issued
// before the first MFMA (register arrays + a scheduling barrier): left alone the compiler keeps one load
// pair in flight and every MFMA waits a full memory round trip (measured: 21 000 of 23 000 cycles of a
// 61-state operation).  pa: packed A operands of this (category, row tile), lane-offset; cl: this
// category's rows of the child's 32-pattern tile + lane (tile-major layout: rows 2t and 2t+1 are the 64
// consecutive floats at 64 t, so every B load is one contiguous 256-byte wave access).
template <int SC>
__device__ __forceinline__ f32x16 mfma_contract(const MBAMD_AS_GLOBAL float* __restrict__ pa,
                                                const MBAMD_AS_GLOBAL float* __restrict__ cl, int S_rt, int half, f32x16 acc)
{
    if constexpr (SC > 0) {
        constexpr int T = (SC + 1) / 2, Tfull = SC / 2;
        float a[T], b[T];
#pragma unroll
        for (int t = 0; t < Tfull; ++t) {
            a[t] = pa[(size_t) t * 64];
            b[t] = cl[64 * t];
        }
        if (SC & 1) {                               // last pair: row S does not exist, feed zeros
            a[T - 1] = pa[(size_t) (T - 1) * 64];
            b[T - 1] = 0.0f;
            if (!half) b[T - 1] = cl[64 * Tfull];
        }
        MBD_SCHED_BARRIER();
#pragma unroll
        for (int t = 0; t < T; ++t) acc = mbd_mfma_f32_32x32x2(a[t], b[t], acc);
    } else {
        constexpr int CH = 16;                      // run-time S: chunks of 16 pairs in flight
        const int Tfull = S_rt / 2;
        for (int t0 = 0; t0 < Tfull; t0 += CH) {
            float a[CH], b[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int t = min(t0 + u, Tfull - 1);
                a[u] = pa[(size_t) t * 64];
                b[u] = cl[64 * t];
                if (t0 + u >= Tfull) a[u] = 0.0f;
            }
            MBD_SCHED_BARRIER();
#pragma unroll
            for (int u = 0; u < CH; ++u) acc = mbd_mfma_f32_32x32x2(a[u], b[u], acc);
        }
        if (S_rt & 1) {
            float b = 0.0f;
            if (!half) b = cl[64 * Tfull];
            acc = mbd_mfma_f32_32x32x2(pa[(size_t) Tfull * 64], b, acc);
        }
    }
    return acc;
}

template <int NT, int SC>     // SC: compile-time state count (0 = run-time S)
__device__ __forceinline__ void mfma_child_factor(const void* ptr, int kind, const float* mbase, int S_rt, int SP,
                                                  int K, int k, int Ppad, int c0, int lane, f32x16 (&acc)[NT])
{
    const int S = SC > 0 ? SC : S_rt;
    const int T = (S + 1) / 2;
    const int half = lane >> 5, col = lane & 31;
    if (kind == CHILD_STATES) {
        const unsigned s = as_global(reinterpret_cast<const uint8_t*>(ptr))[c0 + col];
        const bool missing = s >= (unsigned) S;
        const MBAMD_AS_GLOBAL float* row = as_global(mbase) + ((size_t) k * SP + (missing ? 0u : s)) * SP + 4 * half;
#pragma unroll
        for (int it = 0; it < NT; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i0 = 32 * it + 8 * q;                 // + 4*half + (0..3)
                f4 v = *reinterpret_cast<const MBAMD_AS_GLOBAL f4*>(row + i0);
                if (missing) {
                    const int ib = i0 + 4 * half;
                    v.x = (ib + 0 < S) ? 1.0f : 0.0f;
                    v.y = (ib + 1 < S) ? 1.0f : 0.0f;
                    v.z = (ib + 2 < S) ? 1.0f : 0.0f;
                    v.w = (ib + 3 < S) ? 1.0f : 0.0f;
                }
                acc[it][4 * q + 0] = v.x;
                acc[it][4 * q + 1] = v.y;
                acc[it][4 * q + 2] = v.z;
                acc[it][4 * q + 3] = v.w;
            }
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[it][r] = 0.0f;
    // packed A operands of category k: [NT][T][64] after the K transposed SPxSP matrices
    const MBAMD_AS_GLOBAL float* __restrict__ pa =
        as_global(mbase) + (size_t) K * SP * SP + (size_t) k * NT * T * 64 + lane;
    const MBAMD_AS_GLOBAL float* __restrict__ cl =
        as_global(reinterpret_cast<const float*>(ptr)) + gen_index(K, S, k, 0, c0) + lane;
#pragma unroll
    for (int it = 0; it < NT; ++it) acc[it] = mfma_contract<SC>(pa + (size_t) it * T * 64, cl, S, half, acc[it]);
}

// grid: 8 * ceil(gx/8) * count blocks of 256 threads (4 waves, one 32-pattern tile each),
// gx = ceil(P_pad / 128).  KC = compile-time category count.
template <int NT, int SC, int KC>
__global__ void __launch_bounds__(256)
k_partials_mfma(const PartialsOp* __restrict__ ops, int S_rt, int SP, int Ppad, int gx, int32_t* __restrict__ cumulative)
{
    const int S = SC > 0 ? SC : S_rt;
    int bx, by;
    if (!xcd_aware_block(gx, bx, by)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = (bx * 4 + wave) * 32;
    if (c0 >= Ppad) return;
    const MBAMD_AS_CONST PartialsOp* __restrict__ op = as_const(ops) + by;
    const int k1 = op->c1_kind, k2 = op->c2_kind;
    const void* c1 = op->c1;
    const void* c2 = op->c2;
    const float* m1 = op->m1;
    const float* m2 = op->m2;
    const int mode = op->scale_mode;
    const int half = lane >> 5, col = lane & 31;

    f32x16 out[KC][NT];
    float mx = 0.0f;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        f32x16 f2[NT];
        mfma_child_factor<NT, SC>(c1, k1, m1, S, SP, KC, k, Ppad, c0, lane, out[k]);
        mfma_child_factor<NT, SC>(c2, k2, m2, S, SP, KC, k, Ppad, c0, lane, f2);
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = out[k][it][r] * f2[it][r];
                out[k][it][r] = v;
                const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * half;
                mx = fmaxf(mx, (i < S) ? v : 0.0f);
            }
    }
    int e = 0;
    if (mode == SCALE_WRITE) {
        mx = fmaxf(mx, mbd_shfl_xor(mx, 32));
        e = scale_exponent(mx);
        if (half == 0) {
            as_global(op->scale)[c0 + col] = e;
            if (cumulative != nullptr && e != 0) atomicAdd(cumulative + c0 + col, e);
        }
    } else if (mode == SCALE_READ) {
        e = as_global(op->scale)[c0 + col];
    }
    MBAMD_AS_GLOBAL float* __restrict__ dst = as_global(op->dst) + gen_base(KC, S, c0) + col;
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (i < S) {
                    const float v = out[k][it][r];
                    dst[((size_t) k * S + i) * 32] = (mode != SCALE_NONE) ? scale_pow2(v, -e) : v;
                }
            }
}

// ---------------------------------------------------------------------------------------------
// Finer-grained variant: one workgroup per (operation, 32 patterns), one WAVE per factor tile
// (category k, child c, row tile it): 2*K*NT waves.  A wave runs only T = ceil(S/2) dependent
// MFMAs (10 for S=20, 31 for S=61) instead of 2*K*NT*T, so the small dependency levels near the
// root of the tree -- a handful of operations -- still put thousands of waves on the chip.
// The two child waves of an output tile split its 16 accumulator registers (rows): each hands the
// eight registers it does not keep to its partner through LDS ([piece][8][64 lanes] floats, lane-
// contiguous, conflict free), multiplies, reduces the per-pattern maximum through LDS, rescales and stores.
// blockDim.x = 64 * 2*K*NT (<= 512), dynamic LDS = 2*K*NT * (2 KiB exchange + 128 B maxima + 2 KiB store staging).
// ---------------------------------------------------------------------------------------------
// Up to four operation tables per launch: MrBayes issues one beagleUpdatePartials per eigen-system part
// (codon M3: three), mutually independent; the engine defers them and runs each dependency level
// of all parts as ONE launch.
struct OpTables {
    const PartialsOp* ops[MBAMD_MAX_TABLES];
    int32_t* cum[MBAMD_MAX_TABLES];
    int start[MBAMD_MAX_TABLES + 1];       // operation index range [start[t], start[t+1]) belongs to table t
};

// Eight rows of a tip child's factor: column `state` of P (a row of the transposed matrix), registers
// 8c .. 8c+7 of the 32x32 tile layout = rows 32 it + 8 q + 4 half + (0..3), q = 2c, 2c+1.
template <int SC>
__device__ __forceinline__ void mfma_tip_rows(const void* child, const float* mbase, int S, int SP, int k, int it, int c,
                                              int c0, int half, int col, float (&f)[8])
{
    const unsigned s = as_global(reinterpret_cast<const uint8_t*>(child))[c0 + col];
    const bool missing = s >= (unsigned) S;
    const MBAMD_AS_GLOBAL float* row = as_global(mbase) + ((size_t) k * SP + (missing ? 0u : s)) * SP + 4 * half;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i0 = 32 * it + 8 * (2 * c + q);
        f4 v = *reinterpret_cast<const MBAMD_AS_GLOBAL f4*>(row + i0);
        if (missing) {
            const int ib = i0 + 4 * half;
            v.x = (ib + 0 < S) ? 1.0f : 0.0f;
            v.y = (ib + 1 < S) ? 1.0f : 0.0f;
            v.z = (ib + 2 < S) ? 1.0f : 0.0f;
            v.w = (ib + 3 < S) ? 1.0f : 0.0f;
        }
        f[4 * q + 0] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
}

// One operation for one 32-pattern tile, executed by the 2*K*NT waves of a workgroup.  Wave (k, c, it)
// produces registers 8c .. 8c+7 (eight of the 32 rows) of output tile (k, it).
template <int NT, int SC, int KC>
__device__ __forceinline__ void mfma_split_op(const MBAMD_AS_CONST PartialsOp* __restrict__ op, int32_t* __restrict__ cumulative,
                                              int S_rt, int SP, int Ppad, int c0, int wave, int lane, float* tiles, float* smax)
{
    constexpr int NP = 2 * KC * NT;                 // pieces = waves
    float* stg = smax + NP * 32;                    // [NP][16][32] store staging
    const int S = SC > 0 ? SC : S_rt;
    const int k = wave / (2 * NT), c = (wave / NT) & 1, it = wave % NT;
    const int k1 = op->c1_kind, k2 = op->c2_kind;
    const int mode = op->scale_mode;
    const int half = lane >> 5, col = lane & 31;
    const int T = (S + 1) / 2;
    float out[8];
    float mx = 0.0f;

    if (k1 == CHILD_STATES && k2 == CHILD_STATES) {
        // ---- both children are tips (a third of the operations of a full evaluation): no contraction
        // and no tile exchange -- the wave gathers its eight rows of both factors itself
        float f1[8], f2[8];
        mfma_tip_rows<SC>(op->c1, op->m1, S, SP, k, it, c, c0, half, col, f1);
        mfma_tip_rows<SC>(op->c2, op->m2, S, SP, k, it, c, c0, half, col, f2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 8 * c + j;
            const float v = f1[j] * f2[j];
            out[j] = v;
            const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * half;
            mx = fmaxf(mx, (i < S) ? v : 0.0f);
        }
    } else {
        // ---- this wave's factor tile F_c[k][32*it .. 32*it+31][c0 .. c0+31], exchanged through LDS
        const int kind = c ? k2 : k1;
        const void* child = c ? op->c2 : op->c1;
        const float* mbase = c ? op->m2 : op->m1;
        f32x16 acc;
        if (kind == CHILD_STATES) {
            float lo[8], hi[8];
            mfma_tip_rows<SC>(child, mbase, S, SP, k, it, 0, c0, half, col, lo);
            mfma_tip_rows<SC>(child, mbase, S, SP, k, it, 1, c0, half, col, hi);
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j] = lo[j]; acc[8 + j] = hi[j]; }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const MBAMD_AS_GLOBAL float* __restrict__ pa =
                as_global(mbase) + (size_t) KC * SP * SP + ((size_t) (k * NT + it) * T) * 64 + lane;
            const MBAMD_AS_GLOBAL float* __restrict__ cl =
                as_global(reinterpret_cast<const float*>(child)) + gen_index(KC, S, k, 0, c0) + lane;
            acc = mfma_contract<SC>(pa, cl, S, half, acc);
        }
        // the partner wave (other child, same k and it) needs the eight registers this wave does not keep
        float* mine = tiles + (size_t) wave * 8 * 64;
#pragma unroll
        for (int j = 0; j < 8; ++j) mine[j * 64 + lane] = acc[8 * (1 - c) + j];
        MBAMD_SYNC();
        const float* other = tiles + (size_t) ((k * 2 + (1 - c)) * NT + it) * 8 * 64;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 8 * c + j;
            const float v = acc[r] * other[j * 64 + lane];
            out[j] = v;
            const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * half;
            mx = fmaxf(mx, (i < S) ? v : 0.0f);
        }
    }
    int e = 0;
    if (mode == SCALE_WRITE) {
        mx = fmaxf(mx, mbd_shfl_xor(mx, 32));
        if (half == 0) smax[wave * 32 + col] = mx;
        MBAMD_SYNC();
        float m = 0.0f;
#pragma unroll
        for (int p = 0; p < NP; ++p) m = fmaxf(m, smax[p * 32 + col]);
        e = scale_exponent(m);
        if (wave == 0 && half == 0) {
            as_global(op->scale)[c0 + col] = e;
            if (cumulative != nullptr && e != 0) atomicAdd(cumulative + c0 + col, e);
        }
    } else if (mode == SCALE_READ) {
        e = as_global(op->scale)[c0 + col];
    }
    // ---- store: the wave's 16 rows x 32 patterns are 2 KiB contiguous in the tile-major layout; turn the
    // MFMA register layout (lane = pattern) around in LDS so that each store instruction writes 1 KiB
    // contiguous (dwordx4 per lane, 8 rows) instead of two separate 128-byte rows
    float* stage = stg + (size_t) wave * 512;       // [16 local rows][32 patterns]; local row lr = row - 32 it - 16 c
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int lr = (j & 3) + 4 * half + 8 * (j >> 2);
        stage[lr * 32 + col] = (mode != SCALE_NONE) ? scale_pow2(out[j], -e) : out[j];
    }
    MBAMD_WAVE_SYNC();                              // (lanes read each other's rows: the wave's LDS instructions execute in order, the compiler is told)
    MBAMD_AS_GLOBAL float* __restrict__ dst = as_global(op->dst) + gen_index(KC, S, k, 32 * it + 16 * c, c0) + 4 * (lane & 7);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int lr = (lane >> 3) + 8 * u;
        const f4 v = *reinterpret_cast<const f4*>(stage + lr * 32 + 4 * (lane & 7));
        if (32 * it + 16 * c + lr < S) *reinterpret_cast<MBAMD_AS_GLOBAL f4*>(dst + lr * 32) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Operations whose children are BOTH tips with compact states (a third of the operations of a full
// evaluation, the whole first dependency level): no contraction -- the result is the product of two
// matrix columns.  Pure write traffic, so the unit of work is made large: ONE wave produces the whole
// 32-pattern tile (all K categories, all S rows = 10 KiB for 20 states x 4 categories), no
// cross-wave communication, every gather issued before the first use, rows turned around in a
// per-wave LDS strip so that every store is 1 KiB contiguous.
// Lane (col = lane & 31, half = lane >> 5) holds rows 8 q + 4 half + (0..3) of each category.
// grid = (P_pad / 128) * operations, block = 256 (4 waves = 4 tiles), dynamic LDS = 4 * S * 32 floats.
// ---------------------------------------------------------------------------------------------
template <int SC, int KC>
__global__ void __launch_bounds__(256, (SC > 0 ? 6 : (KC > 1 ? 3 : 4)))     // (any state count, two categories: 128 registers spilled 11 of them)
k_partials_tips(OpTables tabs, int S_rt, int SP, int Ppad, int gx4)
{
    constexpr int QC = SC > 0 ? (SC + 7) / 8 : 8;   // 8-row groups per category
    float* const lds_f = mbd_dyn_lds<float>();
    const int S = SC > 0 ? SC : S_rt;
    const int Q = SC > 0 ? QC : (S + 7) / 8;
    const int bx = blockIdx.x % gx4, by = blockIdx.x / gx4;
    const int wave = mbd_wave_index(), lane = threadIdx.x & 63;
    const int c0 = (bx * 4 + wave) * 32;
    if (c0 >= Ppad) return;
    int tsel = 0;
#pragma unroll
    for (int t = 1; t < MBAMD_MAX_TABLES; ++t) tsel += by >= tabs.start[t] ? 1 : 0;
    const MBAMD_AS_CONST PartialsOp* __restrict__ op = as_const(tabs.ops[tsel]) + (by - tabs.start[tsel]);
    int32_t* __restrict__ cumulative = tabs.cum[tsel];
    const int mode = op->scale_mode;
    const int half = lane >> 5, col = lane & 31;
    const unsigned s1 = as_global(reinterpret_cast<const uint8_t*>(op->c1))[c0 + col];
    const unsigned s2 = as_global(reinterpret_cast<const uint8_t*>(op->c2))[c0 + col];
    const bool miss1 = s1 >= (unsigned) S, miss2 = s2 >= (unsigned) S;
    const MBAMD_AS_GLOBAL float* r1 = as_global(op->m1) + (size_t) (miss1 ? 0u : s1) * SP + 4 * half;
    const MBAMD_AS_GLOBAL float* r2 = as_global(op->m2) + (size_t) (miss2 ? 0u : s2) * SP + 4 * half;
    // products of the two gathered columns, category by category (the factors themselves are transient: the
    // register budget decides how many waves a CU keeps in flight, and in-flight waves are what a write stream needs)
    f4 g1[KC][QC];
    float mx = 0.0f;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        f4 ga[QC], gb[QC];
#pragma unroll
        for (int q = 0; q < QC; ++q) {
            const int qq = (SC > 0 || q < Q) ? q : 0;
            ga[q] = *reinterpret_cast<const MBAMD_AS_GLOBAL f4*>(r1 + (size_t) k * SP * SP + 8 * qq);
            gb[q] = *reinterpret_cast<const MBAMD_AS_GLOBAL f4*>(r2 + (size_t) k * SP * SP + 8 * qq);
        }
#pragma unroll
        for (int q = 0; q < QC; ++q) {
            const int ib = 8 * q + 4 * half;
            f4 a = ga[q], b = gb[q];
            if (miss1) a = (f4) (1.0f);
            if (miss2) b = (f4) (1.0f);
            f4 v = a * b;
            v.x = (ib + 0 < S) ? v.x : 0.0f;         // rows >= S (matrix padding / missing-state ones) do not exist
            v.y = (ib + 1 < S) ? v.y : 0.0f;
            v.z = (ib + 2 < S) ? v.z : 0.0f;
            v.w = (ib + 3 < S) ? v.w : 0.0f;
            g1[k][q] = v;
            mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
    }
    int e = 0;
    if (mode == SCALE_WRITE) {
        mx = fmaxf(mx, mbd_shfl_xor(mx, 32));
        e = scale_exponent(mx);
        if (half == 0) {
            as_global(op->scale)[c0 + col] = e;
            if (cumulative != nullptr && e != 0) atomicAdd(cumulative + c0 + col, e);
        }
    } else if (mode == SCALE_READ) {
        e = as_global(op->scale)[c0 + col];
    }
    float* strip = lds_f + (size_t) wave * S * 32;   // [S rows][32 patterns] of one category
    MBAMD_AS_GLOBAL float* __restrict__ dst = as_global(op->dst) + gen_base(KC, S, c0) + 4 * (lane & 7);
#pragma unroll
    for (int k = 0; k < KC; ++k) {
#pragma unroll
        for (int q = 0; q < QC; ++q) {
            const int ib = 8 * q + 4 * half;
            f4 v = g1[k][q];
            if (mode != SCALE_NONE) { v.x = scale_pow2(v.x, -e); v.y = scale_pow2(v.y, -e); v.z = scale_pow2(v.z, -e); v.w = scale_pow2(v.w, -e); }
            if (ib + 0 < S) strip[(ib + 0) * 32 + col] = v.x;
            if (ib + 1 < S) strip[(ib + 1) * 32 + col] = v.y;
            if (ib + 2 < S) strip[(ib + 2) * 32 + col] = v.z;
            if (ib + 3 < S) strip[(ib + 3) * 32 + col] = v.w;
        }
        MBAMD_WAVE_SYNC();                          // (lanes read each other's rows, see mfma_split_op)
#pragma unroll
        for (int u = 0; u < QC; ++u) {
            const int row = (lane >> 3) + 8 * u;
            if (row < S) {
                const f4 v = *reinterpret_cast<const f4*>(strip + row * 32 + 4 * (lane & 7));
                *reinterpret_cast<MBAMD_AS_GLOBAL f4*>(dst + ((size_t) k * S + row) * 32) = v;
            }
        }
        MBAMD_WAVE_SYNC();                          // (the next category overwrites the strip)
    }
}

// grid = (P_pad / 32) * operations of all tables; one dependency level per launch
template <int NT, int SC, int KC>
__global__ void __launch_bounds__(64 * 2 * KC * NT)
k_partials_mfma_split(OpTables tabs, int S_rt, int SP, int Ppad, int gx)
{
    constexpr int NP = 2 * KC * NT;
    float* const lds_f = mbd_dyn_lds<float>();
    float* tiles = lds_f;                           // [NP][8][64]
    float* smax = lds_f + NP * 8 * 64;              // [NP][32]
    const int bx = blockIdx.x % gx, by = blockIdx.x / gx;       // gx = P_pad / 32 tiles
    const int wave = mbd_wave_index(), lane = threadIdx.x & 63;
    int tsel = 0;
#pragma unroll
    for (int t = 1; t < MBAMD_MAX_TABLES; ++t) tsel += by >= tabs.start[t] ? 1 : 0;
    const MBAMD_AS_CONST PartialsOp* __restrict__ op = as_const(tabs.ops[tsel]) + (by - tabs.start[tsel]);
    mfma_split_op<NT, SC, KC>(op, tabs.cum[tsel], S_rt, SP, Ppad, bx * 32, wave, lane, tiles, smax);
}

// Serial variant for narrow lists (the root-ward path a single MCMC move dirties: one operation per
// dependency level): ONE launch, the workgroup that owns a 32-pattern tile executes the operations of
// its table one after the other.  Patterns are independent, so list order inside the workgroup is all
// the synchronisation a dependent operation needs: the result tile written by the previous
// operation is re-read by the same workgroup (same CU, same write-through L1) after the barrier.
// grid = (P_pad / 32) * tables; tabs.start[t] holds the operation COUNT of table t.
template <int NT, int SC, int KC>
__global__ void __launch_bounds__(64 * 2 * KC * NT)
k_partials_mfma_serial(OpTables tabs, int S_rt, int SP, int Ppad, int gx, long long* __restrict__ trace)
{
    constexpr int NP = 2 * KC * NT;
    float* const lds_f = mbd_dyn_lds<float>();
    float* tiles = lds_f;
    float* smax = lds_f + NP * 8 * 64;
    const int bx = blockIdx.x % gx, tsel = blockIdx.x / gx;
    const int wave = mbd_wave_index(), lane = threadIdx.x & 63;
    const MBAMD_AS_CONST PartialsOp* __restrict__ ops = as_const(tabs.ops[tsel]);
    int32_t* __restrict__ cumulative = tabs.cum[tsel];
    const int count = tabs.start[tsel];
    const bool tracing = trace != nullptr && blockIdx.x == 0 && lane == 0;      // MBAMD_WALK_TRACE (timing experiments)
    for (int o = 0; o < count; ++o) {
        if (tracing) trace[((size_t) o * 8 + wave) * 3 + 0] = mbd_clock();
        mfma_split_op<NT, SC, KC>(ops + o, cumulative, S_rt, SP, Ppad, bx * 32, wave, lane, tiles, smax);
        if (tracing) trace[((size_t) o * 8 + wave) * 3 + 1] = mbd_clock();
        MBAMD_SYNC();                            // results visible; tiles / smax reusable
        if (tracing) trace[((size_t) o * 8 + wave) * 3 + 2] = mbd_clock();
    }
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined serial kernel (compile-time state counts): same work split as k_partials_mfma_serial,
// but the latencies that bound a chain of dependent operations are taken off its critical path:
//   * the operands of operation o+1 that do not depend on operation o -- its packed matrices, the
//     child that is not o's result, tip states -- are loaded into a second register set while o computes;
//   * the result of o stays in LDS (a [K][S][32] copy of the tile, which also stages the 1 KiB stores):
//     when o+1 consumes it, its B operand is a ds_read, not a store -> L2 -> load round trip.
// What remains per operation is the MFMA chain, the two LDS exchanges and three barriers.
// ---------------------------------------------------------------------------------------------
// An operation descriptor in scalar registers: four s_load_dwordx4 (the uint8 fields of PartialsOp would
// otherwise be fetched with VECTOR loads, whose in-order counter then makes every descriptor access
// wait for the previous operation's stores).
struct SpineDesc {
    uint64_t dst, c1, c2, m1, m2, scale;
    uint32_t kinds;                 // c1_kind | c2_kind << 8 | ...
    uint32_t modes;                 // dst_slot | scale_mode << 8 | flags << 16
};
__device__ __forceinline__ SpineDesc spine_desc(const MBAMD_AS_CONST PartialsOp* op)
{
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const MBAMD_AS_CONST u4* q = reinterpret_cast<const MBAMD_AS_CONST u4*>(op);
    const u4 a = q[0], b = q[1], c = q[2], e = q[3];
    SpineDesc d;
    d.dst = a.x | ((uint64_t) a.y << 32);
    d.c1 = a.z | ((uint64_t) a.w << 32);
    d.c2 = b.x | ((uint64_t) b.y << 32);
    d.m1 = b.z | ((uint64_t) b.w << 32);
    d.m2 = c.x | ((uint64_t) c.y << 32);
    d.scale = c.z | ((uint64_t) c.w << 32);
    d.kinds = e.x;
    d.modes = e.y;
    return d;
}
static_assert(offsetof(PartialsOp, c1_kind) == 48 && offsetof(PartialsOp, scale_mode) == 53, "SpineDesc unpacking");

template <int SC>
struct SpineOperands {
    static constexpr int T = (SC + 1) / 2;
    float a[T], b[T];               // packed matrix tile / child rows
    unsigned state;                 // tip child: this lane's state code
    int eread;                      // SCALE_READ: this pattern's exponent
};

// Issue the loads of an operation's operands for compute wave (k, c, it): ALWAYS the same 2T + 2 vector
// loads, with harmless addresses where an operand is not needed (tip child, forwarded child, no
// SCALE_READ).  Compute waves issue no other vector-memory instruction, so the in-order counter lets
// operation o wait for exactly its own operands while those of o+1 stay in flight.
// A tip child contributes its state code only: its B operand is the one-hot vector of that state, so a
// tip costs an MFMA chain but no dependent gather.
template <int NT, int SC, int KC>
__device__ __forceinline__ void spine_prefetch(const SpineDesc& d, uint64_t prev1, uint64_t prev2, int SP, int k, int c, int it,
                                               int c0, int lane, SpineOperands<SC>& n)
{
    constexpr int T = (SC + 1) / 2;
    const int col = lane & 31;
    const int kind = (d.kinds >> (8 * c)) & 0xFF;
    const int mode = (d.modes >> 8) & 0xFF;
    const uint64_t child = c ? d.c2 : d.c1;
    const float* mbase = reinterpret_cast<const float*>(c ? d.m2 : d.m1);
    const MBAMD_AS_GLOBAL float* __restrict__ pa =
        as_global(mbase) + (size_t) KC * SP * SP + ((size_t) (k * NT + it) * T) * 64 + lane;
    const bool fromGlobal = kind != CHILD_STATES && child != prev1 && child != prev2;
    const MBAMD_AS_GLOBAL float* __restrict__ cl =
        fromGlobal ? as_global(reinterpret_cast<const float*>(child)) + gen_index(KC, SC, k, 0, c0) + lane : pa;
    const MBAMD_AS_GLOBAL uint8_t* st = (kind == CHILD_STATES) ? as_global(reinterpret_cast<const uint8_t*>(child)) + c0 + col
                                                                : reinterpret_cast<const MBAMD_AS_GLOBAL uint8_t*>(pa);
    const MBAMD_AS_GLOBAL int32_t* er = (mode == SCALE_READ) ? as_global(reinterpret_cast<const int32_t*>(d.scale)) + c0 + col
                                                              : reinterpret_cast<const MBAMD_AS_GLOBAL int32_t*>(pa);
#pragma unroll
    for (int t = 0; t < T; ++t) n.a[t] = pa[(size_t) t * 64];
#pragma unroll
    for (int t = 0; t < T; ++t) n.b[t] = cl[64 * min(t, (SC - 1 - (lane >> 5)) / 2)];   // (odd S: lanes >= 32 have no row S)
    n.state = *st;
    n.eread = *er;
}

// Compute waves: factor tile -> exchange -> product -> maximum -> rescaled result rows into LDS slot `mySlot`.
template <int NT, int SC, int KC>
__device__ __forceinline__ void spine_compute(const SpineDesc& d, uint64_t prev1, uint64_t prev2, int k, int c, int it, int wave,
                                              int lane, const SpineOperands<SC>& cur, float* tiles, float* smax, float* slotPrev1,
                                              float* slotPrev2, float* mySlot)
{
    constexpr int NP = 2 * KC * NT;
    constexpr int S = SC, T = (SC + 1) / 2, Tfull = SC / 2;
    const int half = lane >> 5, col = lane & 31;
    const int mode = (d.modes >> 8) & 0xFF;
    const int kind = (d.kinds >> (8 * c)) & 0xFF;
    const uint64_t child = c ? d.c2 : d.c1;
    float b[T];
    bool missing = false;
    if (kind == CHILD_STATES) {                     // one-hot column selector (a missing state is patched below)
        missing = cur.state >= (unsigned) S;
#pragma unroll
        for (int t = 0; t < T; ++t) b[t] = (cur.state == (unsigned) (2 * t + half)) ? 1.0f : 0.0f;
    } else if (child == prev1 || child == prev2) {  // one of the last two results, still in LDS
        const float* fl = (child == prev1 ? slotPrev1 : slotPrev2) + (size_t) k * S * 32 + lane;
#pragma unroll
        for (int t = 0; t < Tfull; ++t) b[t] = fl[64 * t];
        if (SC & 1) b[T - 1] = half ? 0.0f : fl[64 * Tfull];
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t) b[t] = cur.b[t];
        if (SC & 1) b[T - 1] = half ? 0.0f : b[T - 1];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t) acc = mbd_mfma_f32_32x32x2(cur.a[t], b[t], acc);
    if (kind == CHILD_STATES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * half;
            acc[r] = missing ? ((i < S) ? 1.0f : 0.0f) : acc[r];
        }
    }
    // hand the eight registers this wave does not keep to its partner (other child, same k and it)
    float* mine = tiles + (size_t) wave * 8 * 64;
    float keep[8];
    if (c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { mine[j * 64 + lane] = acc[j]; keep[j] = acc[8 + j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { mine[j * 64 + lane] = acc[8 + j]; keep[j] = acc[j]; }
    }
    MBAMD_SYNC();
    const float* other = tiles + (size_t) ((k * 2 + (1 - c)) * NT + it) * 8 * 64;
    float out[8];
    float mx = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = keep[j] * other[j * 64 + lane];
        out[j] = v;
        const int i = 32 * it + 16 * c + (j & 3) + 8 * (j >> 2) + 4 * half;
        mx = fmaxf(mx, (i < S) ? v : 0.0f);
    }
    int e = 0;
    if (mode == SCALE_WRITE) {
        mx = fmaxf(mx, mbd_shfl_xor(mx, 32));
        if (half == 0) smax[wave * 32 + col] = mx;
        MBAMD_SYNC();
        float m = 0.0f;
#pragma unroll
        for (int p = 0; p < NP; ++p) m = fmaxf(m, smax[p * 32 + col]);
        e = scale_exponent(m);
    } else if (mode == SCALE_READ) {
        e = cur.eread;
    }
    float* frow = mySlot + ((size_t) k * S + 32 * it + 16 * c) * 32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int lr = (j & 3) + 4 * half + 8 * (j >> 2);
        if (32 * it + 16 * c + lr < S) frow[lr * 32 + col] = (mode != SCALE_NONE) ? scale_pow2(out[j], -e) : out[j];
    }
    MBAMD_SYNC();
}

// Writer wave: same barriers; stores the exponents (+ cumulative atomics) and copies the finished tile --
// K*S*32 contiguous floats both in the LDS slot and in the tile-major buffer -- out with 1 KiB stores.
template <int NT, int SC, int KC>
__device__ __forceinline__ void spine_write(const SpineDesc& d, int32_t* __restrict__ cumulative, int c0, int lane,
                                            const float* smax, const float* mySlot)
{
    constexpr int NP = 2 * KC * NT;
    constexpr int F4 = KC * SC * 32 / 4;            // float4s per tile
    const int mode = (d.modes >> 8) & 0xFF;
    MBAMD_SYNC();                                // factor tiles exchanged
    if (mode == SCALE_WRITE) {
        MBAMD_SYNC();                            // maxima posted
        if (lane < 32) {
            float m = 0.0f;
#pragma unroll
            for (int p = 0; p < NP; ++p) m = fmaxf(m, smax[p * 32 + lane]);
            const int e = scale_exponent(m);
            as_global(reinterpret_cast<int32_t*>(d.scale))[c0 + lane] = e;
            if (cumulative != nullptr && e != 0) atomicAdd(cumulative + c0 + lane, e);
        }
    }
    MBAMD_SYNC();                                // result tile complete in LDS
    MBAMD_AS_GLOBAL f4* __restrict__ dst =
        reinterpret_cast<MBAMD_AS_GLOBAL f4*>(as_global(reinterpret_cast<float*>(d.dst)) + gen_base(KC, SC, c0));
    const f4* src = reinterpret_cast<const f4*>(mySlot);
#pragma unroll
    for (int u = 0; u < (F4 + 63) / 64; ++u) {
        const int idx = 64 * u + lane;
        if (idx < F4) dst[idx] = src[idx];
    }
}

// grid = (P_pad / 32) * tables; tabs.start[t] = operations of table t; block = 64 * (2*K*NT + 1);
// dynamic LDS = NP * (2 KiB + 128 B) + 2 * K*S*32 floats.  Pipeline: while operation o computes, the operands
// of o+1 and the descriptor of o+2 are in flight and the writer wave is still storing o-1.
template <int NT, int SC, int KC>
__global__ void __launch_bounds__(64 * (2 * KC * NT + 1))
k_partials_mfma_spine(OpTables tabs, int SP, int gx, long long* __restrict__ trace)
{
    static_assert(SC > 0, "compile-time state count required");
    constexpr int NP = 2 * KC * NT;
    constexpr int TILE = KC * SC * 32;
    float* const lds_f = mbd_dyn_lds<float>();
    float* tiles = lds_f;                           // [NP][8][64]
    float* smax = lds_f + NP * 8 * 64;              // [NP][32]
    float* slots = smax + NP * 32;                  // [2][K][S][32]: the last two results
    const int bx = blockIdx.x % gx, tsel = blockIdx.x / gx;
    const int wave = mbd_wave_index(), lane = threadIdx.x & 63;
    const int c0 = bx * 32;
    const MBAMD_AS_CONST PartialsOp* __restrict__ ops = as_const(tabs.ops[tsel]);
    int32_t* __restrict__ cumulative = tabs.cum[tsel];
    const int count = tabs.start[tsel];
    if (count <= 0) return;
    const bool tracing = trace != nullptr && blockIdx.x == 0 && lane == 0;      // MBAMD_WALK_TRACE (timing experiments)
    if (wave == NP) {                               // ---- writer wave
        for (int o = 0; o < count; ++o) {
            const SpineDesc d = spine_desc(ops + o);
            spine_write<NT, SC, KC>(d, cumulative, c0, lane, smax, slots + (size_t) (o & 1) * TILE);
        }
        return;
    }
    const int k = wave / (2 * NT), c = (wave / NT) & 1, it = wave % NT;
    SpineOperands<SC> A, B;
    SpineDesc d = spine_desc(ops);                                  // operation o
    SpineDesc dn = spine_desc(ops + (count > 1 ? 1 : 0));           // operation o + 1
    uint64_t prev1 = 0, prev2 = 0;                                  // results of o-1, o-2 (LDS slots (o-1)&1, o&1)
    spine_prefetch<NT, SC, KC>(d, prev1, prev2, SP, k, c, it, c0, lane, A);
    for (int o = 0; o < count; o += 2) {
        float* s0 = slots;                           // even operations write slot 0
        float* s1 = slots + TILE;
        {
            if (tracing) trace[((size_t) o * 8 + (wave & 7)) * 3 + 0] = mbd_clock();
            const SpineDesc dnn = spine_desc(ops + min(o + 2, count - 1));
            spine_prefetch<NT, SC, KC>(dn, d.dst, prev1, SP, k, c, it, c0, lane, B);     // (harmless repeat of the last operation at the end)
            spine_compute<NT, SC, KC>(d, prev1, prev2, k, c, it, wave, lane, A, tiles, smax, s1, s0, s0);
            if (tracing) trace[((size_t) o * 8 + (wave & 7)) * 3 + 2] = mbd_clock();
            prev2 = prev1; prev1 = d.dst; d = dn; dn = dnn;
        }
        if (o + 1 >= count) break;
        {
            if (tracing) trace[((size_t) (o + 1) * 8 + (wave & 7)) * 3 + 0] = mbd_clock();
            const SpineDesc dnn = spine_desc(ops + min(o + 3, count - 1));
            spine_prefetch<NT, SC, KC>(dn, d.dst, prev1, SP, k, c, it, c0, lane, A);
            spine_compute<NT, SC, KC>(d, prev1, prev2, k, c, it, wave, lane, B, tiles, smax, s0, s1, s1);
            if (tracing) trace[((size_t) (o + 1) * 8 + (wave & 7)) * 3 + 2] = mbd_clock();
            prev2 = prev1; prev1 = d.dst; d = dn; dn = dnn;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Root / edge integration for the general-state path (Likelihood_Gen / _NY98, reference
// src/likelihood.c:6037-6260, 6671-6800): eight threads per pattern (each takes the states i = g mod 8),
// a workgroup per 32-pattern tile of the tile-major layout -- same arithmetic as k_integrate_lnl<false>,
// an eighth of the serial chain and twice the workgroups.  The root case issues all its loads first.
// grid = P_pad/32, block = 256 (lane & 31 = pattern, threadIdx >> 5 = state group).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_integrate_lnl_wide(IntegrateArgs a, int S, int SP, int K, int P, int Ppad, const double* __restrict__ pattern_weights,
                     double* __restrict__ site, double* __restrict__ wsite)
{
    __shared__ double part[MBAMD_MAX_SUBSETS][8][32];
    const int p = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + p;
    const bool live = c < P;
    for (int n = 0; n < a.count; ++n) {
        double like = 0.0;
        if (live) {
            const float* __restrict__ par = a.parent[n] + gen_base(K, S, c);
            const double* __restrict__ fr = a.freqs[n];
            for (int k = 0; k < K; ++k) {
                double cat = 0.0;
                const float* __restrict__ pk = par + (size_t) k * S * 32;
                if (a.child[n] == nullptr) {
                    for (int i0 = 0; i0 < S; i0 += 64) {
                        float v[8];
                        double f[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = i0 + g + 8 * u;
                            v[u] = (i < S) ? pk[i * 32] : 0.0f;
                            f[u] = (i < S) ? fr[i] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) cat += (double) v[u] * f[u];
                    }
                } else if (a.child_kind[n] == CHILD_STATES) {
                    const unsigned s = reinterpret_cast<const uint8_t*>(a.child[n])[c];
                    const float* __restrict__ mrow = a.matrix[n] + (size_t) k * SP * SP + (size_t) (s < (unsigned) S ? s : 0) * SP;
                    for (int i = g; i < S; i += 8) {
                        const float pc = (s >= (unsigned) S) ? 1.0f : mrow[i];
                        cat += (double) (pk[i * 32] * pc) * fr[i];
                    }
                } else {
                    const float* __restrict__ ch = reinterpret_cast<const float*>(a.child[n]) + gen_index(K, S, k, 0, c);
                    const float* __restrict__ m = a.matrix[n] + (size_t) k * SP * SP;
                    for (int i = g; i < S; i += 8) {
                        float acc = 0.0f;
                        for (int j = 0; j < S; ++j) acc = fmaf(m[(size_t) j * SP + i], ch[j * 32], acc);
                        cat += (double) (pk[i * 32] * acc) * fr[i];
                    }
                }
                like += cat * a.weights[n][k];
            }
        }
        part[n][g][p] = like;
    }
    MBAMD_SYNC();
    if (g != 0) return;                              // lanes 0..31 of wave 0 finish
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
    } else if (c < Ppad) {
        site[c] = 0.0;
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) wl += mbd_shfl_down_32(wl, off);
    if (p == 0) wsite[blockIdx.x] = wl;
}

}  // namespace mbamd
#endif
