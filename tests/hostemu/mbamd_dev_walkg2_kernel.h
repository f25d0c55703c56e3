// mbamd_dev_walkg2_kernel.h -- TEST ONLY (tests/hostemu): k_walkg2 reads the same arguments, programs, arenas and slot schedule as
// k_walkg; the emulation runs both on the one plain-loop twin of mbamd_dev_walkg_kernel.h (host logic only: programs without
// leading no-ops, the row split's pairs, the PREV / DRAIN flags are ignored by a serial walk).  Never part of the product.
#ifndef MBAMD_DEV_WALKG2_KERNEL_H_
#define MBAMD_DEV_WALKG2_KERNEL_H_
namespace mbamd {
template <int SC, int WMAX, bool PAIR, class ARGS = WalkGArgs>
__global__ void k_walkg2(ARGS AA) { k_walkg<SC, WMAX, 1, 1, ARGS>(AA); }
}  // namespace mbamd
#endif
