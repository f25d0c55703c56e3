// mbamd_dev_walk4_args.h -- TEST ONLY (tests/hostemu): twin of mrbayes_amd/csrc/device/mbamd_dev_walk4_args.h
#ifndef MBAMD_DEV_WALK4_ARGS_H_
#define MBAMD_DEV_WALK4_ARGS_H_
namespace mbamd {
__device__ __forceinline__ const Walk4Entry* walk4_program(const Walk4ArgsInline& a) { return a.inl; }
}  // namespace mbamd
#endif
