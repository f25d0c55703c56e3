// mbamd_dev_pars.h -- TEST ONLY (tests/hostemu): twin of mrbayes_amd/csrc/device/mbamd_dev_pars.h
#ifndef MBAMD_DEV_PARS_H_
#define MBAMD_DEV_PARS_H_
namespace mbamd {
struct ParsDesc {
    const int* p;
    int get(int i) const { return p[i]; }
};
template <int CH> static inline ParsDesc pars_desc_load(const ParsStep* steps, int ch, int)
{
    return {reinterpret_cast<const int*>(steps + (size_t) ch * CH)};
}
}  // namespace mbamd
#endif
