// mbamd_dev_walk4_args.h (gfx950): a walk program that travels in the kernel arguments is read from the argument block itself
#ifndef MBAMD_DEV_WALK4_ARGS_H_
#define MBAMD_DEV_WALK4_ARGS_H_
namespace mbamd {
// (the address of a by-value kernel parameter would be that of a private copy: read the argument block itself)
__device__ __forceinline__ const Walk4Entry* walk4_program(const Walk4ArgsInline&)
{
    return reinterpret_cast<const Walk4Entry*>((uintptr_t) __builtin_amdgcn_kernarg_segment_ptr() + offsetof(Walk4ArgsInline, inl));
}
}  // namespace mbamd
#endif
