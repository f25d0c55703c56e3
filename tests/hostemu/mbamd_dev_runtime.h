// mbamd_dev_runtime.h -- TEST ONLY (tests/hostemu): the "runtime" of the host-emulation build (hip_emu.h).  Never part of the product.
#ifndef MBAMD_DEV_RUNTIME_H_
#define MBAMD_DEV_RUNTIME_H_
#include <memory>
#include "hip_emu.h"
#endif
