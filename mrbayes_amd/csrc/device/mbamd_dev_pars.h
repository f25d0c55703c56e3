// mbamd_dev_pars.h (gfx950) -- the chunk descriptors of the parsimony walk (mbamd_parsimony.h): ONE vector load, lane l holding
// dword l; the uniform fields come back out with v_readlane.  Twin for the TEST-ONLY host emulation: tests/hostemu/.
#ifndef MBAMD_DEV_PARS_H_
#define MBAMD_DEV_PARS_H_
namespace mbamd {
struct ParsDesc {
    int v;
    __device__ __forceinline__ int get(int i) const { return __builtin_amdgcn_readlane(v, i); }
};
template <int CH> __device__ __forceinline__ ParsDesc pars_desc_load(const ParsStep* steps, int ch, int lane)
{
    return {reinterpret_cast<const int*>(steps + (size_t) ch * CH)[lane & (CH * 8 - 1)]};
}
}  // namespace mbamd
#endif
