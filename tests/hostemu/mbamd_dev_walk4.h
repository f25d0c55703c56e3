// mbamd_dev_walk4.h -- TEST ONLY (tests/hostemu): plain-C++ twins of the device primitives of the 4-state tree-walk kernel
// (mrbayes_amd/csrc/device/mbamd_dev_walk4.h).  Never part of the product.
#ifndef MBAMD_DEV_WALK4_H_
#define MBAMD_DEV_WALK4_H_
namespace mbamd {
struct Walk4Mat { float m[16]; };
__device__ inline Walk4Mat walk4_load_matrix(const float* p) { Walk4Mat r; for (int i = 0; i < 16; ++i) r.m[i] = p[i]; return r; }
__device__ inline Walk4Planes walk4_load_planes(const uint64_t* p) { Walk4Planes r; for (int i = 0; i < 4; ++i) r.p[i] = p[i]; return r; }
__device__ inline f4 walk4_tip_vector(const Walk4Planes& t, unsigned lane)
{
    f4 v;
    v.x = (float) (t.p[0] >> lane & 1u); v.y = (float) (t.p[1] >> lane & 1u);
    v.z = (float) (t.p[2] >> lane & 1u); v.w = (float) (t.p[3] >> lane & 1u);
    return v;
}
__device__ inline void walk4_dma(const f4* base, unsigned lane, f4* slot) { slot[lane] = base[lane]; }
__device__ inline void walk4_dma_exps(const int8_t* base, unsigned lane, int* stage) { stage[lane] = base[lane]; }
__device__ inline void walk4_wait_vm(unsigned) {}
__device__ inline void walk4_barrier() { mbamd_emu_barrier(); }
__device__ inline Walk4Entry walk4_load_entry(const Walk4Entry* p) { return *p; }
__device__ inline Walk4Entry walk4_entry_from_lds(const Walk4Entry* p) { return *p; }
__device__ inline void walk4_program_to_lds(const Walk4Entry* src, Walk4Entry* lds, int entries, unsigned lane)
{
    // (threads run one after the other here: every thread copies the whole program before it reads it)
    (void) lane;
    for (int i = 0; i < entries; ++i) lds[i] = src[i];
}

// f_i = sum_j P(i->j) v_j with the transposed matrix mT[j][i] in scalar registers; the same fma chain
// (j = 0..3, first term a plain product) as the reference's scalar loop order, two rows per v_pk_fma_f32.
__device__ __forceinline__ f4 walk4_matvec(const Walk4Mat& M, f4 v)
{
    f4 r;
    const float* m = M.m;
    r.x = fmaf(m[12], v.w, fmaf(m[8], v.z, fmaf(m[4], v.y, m[0] * v.x)));
    r.y = fmaf(m[13], v.w, fmaf(m[9], v.z, fmaf(m[5], v.y, m[1] * v.x)));
    r.z = fmaf(m[14], v.w, fmaf(m[10], v.z, fmaf(m[6], v.y, m[2] * v.x)));
    r.w = fmaf(m[15], v.w, fmaf(m[11], v.z, fmaf(m[7], v.y, m[3] * v.x)));
    return r;
}


struct Walk4Lds { char* mine; unsigned lane; };
inline Walk4Lds walk4_lds(char* mine, unsigned lane) { return Walk4Lds{mine, lane}; }
inline void walk4_prefetch(const Walk4Lds& L, const f4* src, unsigned dst) { walk4_dma(src, L.lane, reinterpret_cast<f4*>(L.mine + MBAMD_W4_STAGE + dst)); }
inline void walk4_fetch_exps(const Walk4Lds& L, const int8_t* src, unsigned lane, int parity) { walk4_dma_exps(src, lane, reinterpret_cast<int*>(L.mine) + 64 * parity); }
inline void walk4_store(f4* P, int8_t* E, unsigned lane, f4 out, int e) { P[lane] = out; E[lane] = (int8_t) e; }
inline void walk4_store_partials(f4* P, unsigned lane, f4 out) { P[lane] = out; }
inline f4 walk4_load_f4(const f4* p) { return *p; }
inline void walk4_wave_lds_fence() { MBAMD_WAVE_SYNC(); }
inline Walk4Mat walk4_matrix_from_lds(const f4* p) { Walk4Mat r; for (int i = 0; i < 4; ++i) { r.m[4 * i] = p[i].x; r.m[4 * i + 1] = p[i].y; r.m[4 * i + 2] = p[i].z; r.m[4 * i + 3] = p[i].w; } return r; }
}  // namespace mbamd
#endif
