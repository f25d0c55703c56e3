// mbamd_dev_walk4.h (gfx950) -- device primitives of the 4-state tree-walk kernel (mbamd_walk4.h): scalar-path loads into SGPRs,
// lane masks from bitplanes, LDS-DMA, exact vmcnt waits, v_pk_fma_f32 with scalar matrix operands, non-temporal stores.
// The TEST-ONLY host emulation has a plain-C++ header of the same name in front on its include path (tests/hostemu/).
#ifndef MBAMD_DEV_WALK4_H_
#define MBAMD_DEV_WALK4_H_
namespace mbamd {
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u8v __attribute__((ext_vector_type(8)));
typedef unsigned u16v __attribute__((ext_vector_type(16)));
typedef unsigned long ul4v __attribute__((ext_vector_type(4)));
struct Walk4Mat { f16v m; };
// wave-uniform, read-only: constant address space -> s_load_dwordx16 / s_load_dwordx8
__device__ __forceinline__ Walk4Mat walk4_load_matrix(const float* p)
{
    Walk4Mat r;
    r.m = *reinterpret_cast<const MBAMD_AS_CONST f16v*>((uintptr_t) p);
    return r;
}
__device__ __forceinline__ Walk4Entry walk4_load_entry(const Walk4Entry* p)
{
    const u8v v = *reinterpret_cast<const MBAMD_AS_CONST u8v*>((uintptr_t) p);
    Walk4Entry e;
    e.ctl = v[0]; e.dst = v[1]; e.c1 = v[2]; e.c2 = v[3]; e.m1 = v[4]; e.m2 = v[5]; e.ewrite = v[6]; e.eread = v[7];
    return e;
}
// an entry of a program that the wave copied into LDS: every lane reads the same 32 bytes, the first lane's copy goes to scalar registers
// (k_path4: a program in the kernel arguments sits in host-visible memory -- read through the scalar cache entry by entry, every other
//  entry was a round trip to it; one vector load per 32 entries brings the whole program in)
__device__ __forceinline__ Walk4Entry walk4_entry_from_lds(const Walk4Entry* p)
{
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const u4v a = reinterpret_cast<const u4v*>(p)[0], b = reinterpret_cast<const u4v*>(p)[1];
    Walk4Entry e;
    e.ctl = __builtin_amdgcn_readfirstlane(a[0]); e.dst = __builtin_amdgcn_readfirstlane(a[1]);
    e.c1 = __builtin_amdgcn_readfirstlane(a[2]); e.c2 = __builtin_amdgcn_readfirstlane(a[3]);
    e.m1 = __builtin_amdgcn_readfirstlane(b[0]); e.m2 = __builtin_amdgcn_readfirstlane(b[1]);
    e.ewrite = __builtin_amdgcn_readfirstlane(b[2]); e.eread = __builtin_amdgcn_readfirstlane(b[3]);
    return e;
}
// 16 bytes per lane of a program (any memory the device can read, the kernel arguments included) -> LDS
__device__ __forceinline__ void walk4_program_to_lds(const Walk4Entry* src, Walk4Entry* lds, int entries, unsigned lane)
{
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const MBAMD_AS_GLOBAL u4v* s = reinterpret_cast<const MBAMD_AS_GLOBAL u4v*>((uintptr_t) src);
    u4v* d = reinterpret_cast<u4v*>(lds);
    for (int i = (int) lane; i < 2 * entries; i += 64) d[i] = s[i];
    // other LANES wrote what this lane reads next: the hardware needs nothing (a wave's LDS instructions execute in order), the compiler
    // must be told -- to it a thread that stored one piece may read the others as anything
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    MBAMD_WAVE_SYNC();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ Walk4Planes walk4_load_planes(const uint64_t* p)
{
    const ul4v v = *reinterpret_cast<const MBAMD_AS_CONST ul4v*>((uintptr_t) p);
    Walk4Planes r;
    r.p[0] = v[0]; r.p[1] = v[1]; r.p[2] = v[2]; r.p[3] = v[3];
    return r;
}
// a bitplane in a scalar register pair IS a lane mask: one v_cndmask_b32 per state
__device__ __forceinline__ f4 walk4_tip_vector(const Walk4Planes& t, unsigned)
{
    f4 v;
    asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(v.x) : "s"(t.p[0]));
    asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(v.y) : "s"(t.p[1]));
    asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(v.z) : "s"(t.p[2]));
    asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(v.w) : "s"(t.p[3]));
    return v;
}
// LDS-DMA: 64 lanes x 16 bytes (base + lane16 each) straight into the 1 KiB LDS slot at byte address lds_dst
// (lane-linear).  `base` is wave-uniform (scalar registers).  M0 is compiler-reserved: saved and restored inside.
__device__ __forceinline__ void walk4_dma(const f4* base, unsigned lane16, unsigned lds_dst)
{
    // sc0 sc1: served by L2, never by this CU's vector L1 (the line is read once, and it may have been written by
    // another wave of this workgroup a moment ago)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc0 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(base), "s"(lds_dst) : "memory");
}
// one signed byte per lane (base + lane) -> a dword per lane at LDS byte address lds_dst + 4 * lane
__device__ __forceinline__ void walk4_dma_exps(const int8_t* base, unsigned lane, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_sbyte %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane), "s"(base), "s"(lds_dst) : "memory");
}
// wait until at most n vector-memory instructions of this wave are outstanding (s_waitcnt takes an immediate: the
// host rounds n down to one of these values; only entries that read a prefetched child come here)
__device__ __forceinline__ void walk4_wait_vm(unsigned n)
{
#define MBAMD_W4_WAIT(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        MBAMD_W4_WAIT(1) MBAMD_W4_WAIT(2) MBAMD_W4_WAIT(3) MBAMD_W4_WAIT(4) MBAMD_W4_WAIT(5) MBAMD_W4_WAIT(6)
        MBAMD_W4_WAIT(8) MBAMD_W4_WAIT(10) MBAMD_W4_WAIT(12) MBAMD_W4_WAIT(16) MBAMD_W4_WAIT(20) MBAMD_W4_WAIT(24)
        MBAMD_W4_WAIT(32) MBAMD_W4_WAIT(40) MBAMD_W4_WAIT(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef MBAMD_W4_WAIT
}
__device__ __forceinline__ void walk4_barrier()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// f_i = sum_j P(i->j) v_j with the transposed matrix mT[j][i] in scalar registers; the same fma chain
// (j = 0..3, first term a plain product) as the reference's scalar loop order, two rows per v_pk_fma_f32.
__device__ __forceinline__ f4 walk4_matvec(const Walk4Mat& M, f4 v)
{
    f4 r;
    const f16v m = M.m;
    f2v lo = f2v{m[0], m[1]} * f2v{v.x, v.x};
    f2v hi = f2v{m[2], m[3]} * f2v{v.x, v.x};
    lo = __builtin_elementwise_fma(f2v{m[4], m[5]}, f2v{v.y, v.y}, lo);
    hi = __builtin_elementwise_fma(f2v{m[6], m[7]}, f2v{v.y, v.y}, hi);
    lo = __builtin_elementwise_fma(f2v{m[8], m[9]}, f2v{v.z, v.z}, lo);
    hi = __builtin_elementwise_fma(f2v{m[10], m[11]}, f2v{v.z, v.z}, hi);
    lo = __builtin_elementwise_fma(f2v{m[12], m[13]}, f2v{v.w, v.w}, lo);
    hi = __builtin_elementwise_fma(f2v{m[14], m[15]}, f2v{v.w, v.w}, hi);
    r.x = lo[0]; r.y = lo[1]; r.z = hi[0]; r.w = hi[1];
    return r;
}


// this wave's LDS window as the LDS-DMA forms want it: byte addresses in the LDS address space
struct Walk4Lds { unsigned lane16, stage_lds, slots_lds; };
__device__ __forceinline__ Walk4Lds walk4_lds(char* mine, unsigned lane)
{
    Walk4Lds L;
    L.lane16 = lane * 16u;
    L.stage_lds = (unsigned) (uintptr_t) (__attribute__((address_space(3))) char*) mine;
    L.slots_lds = L.stage_lds + MBAMD_W4_STAGE;
    return L;
}
// a child that lives in HBM -> the slot at byte offset dst of this wave's slots
__device__ __forceinline__ void walk4_prefetch(const Walk4Lds& L, const f4* src, unsigned dst) { walk4_dma(src, L.lane16, L.slots_lds + dst); }
// stored exponents of the next entry -> landing area `parity`
__device__ __forceinline__ void walk4_fetch_exps(const Walk4Lds& L, const int8_t* src, unsigned lane, int parity) { walk4_dma_exps(src, lane, L.stage_lds + 256u * (unsigned) parity); }
// 1 KiB contiguous per wave; never waited for.  Non-temporal: the result is not read again in this launch (parents read the
// LDS copy or the register), so it must not push the matrices and programs out of L2
__device__ __forceinline__ void walk4_store(f4* P, int8_t* E, unsigned lane, f4 out, int e)
{
    __builtin_nontemporal_store(out, as_global(P) + lane);
    // (the byte store in the scalar-base + 32-bit lane offset form: the compiler builds a 64-bit address per lane instead)
    asm volatile("global_store_byte %0, %1, %2 nt" :: "v"(lane), "v"(e), "s"(E) : "memory");
}
// a plain 16-byte vector load (k_path4's matrices: eight lanes per entry)
__device__ __forceinline__ f4 walk4_load_f4(const f4* p) { return *reinterpret_cast<const MBAMD_AS_GLOBAL f4*>((uintptr_t) p); }
// other LANES of this wave wrote LDS that this lane reads next (or the reverse): a wave's LDS instructions execute in order, the compiler is told
__device__ __forceinline__ void walk4_wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    MBAMD_WAVE_SYNC();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// a 4 x 4 matrix from LDS, the same address in every lane (broadcast reads): the operand of walk4_matvec in vector registers
__device__ __forceinline__ Walk4Mat walk4_matrix_from_lds(const f4* p)
{
    const f4 a = p[0], b = p[1], c = p[2], d = p[3];
    Walk4Mat r;
    r.m = f16v{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    return r;
}
// the same without exponents (an entry that does not rescale, or divides by stored exponents: nothing to record)
__device__ __forceinline__ void walk4_store_partials(f4* P, unsigned lane, f4 out) { __builtin_nontemporal_store(out, as_global(P) + lane); }
}  // namespace mbamd
#endif
