// smmc_host.h -- ONE statement of the host-buffer pinning rules, shared by the engine (smmc_capi.cpp), the
// group (smmc_group.cpp) and the C++ drop-in layer (smmc_dropin.cpp).  Not installed.
//
// Round 3 had three copies of these rules that had drifted apart: the group hard-coded the
// threshold and tested only a buffer's first byte for "already pinned", the drop-in treated SMMC_PIN_HOST=chunk
// as "whole".  The reference pins its result with cudaMallocHost (src/simulations.cu:591-592); here the
// caller owns the buffer, so it is page-locked in place (hipHostRegister) for the duration of a call.
#pragma once
#include <stdint.h>

namespace smmc {

// SMMC_PIN_HOST: "0" (anything else than the words below) never; "1" / "whole" (default) the whole buffer up
// front; "chunk" chunk by chunk, one chunk ahead of the copies (every page has one owning chunk).
enum PinPolicy : int { kPinNever = 0, kPinWhole = 1, kPinChunk = 2 };
PinPolicy pin_policy_from_env();

// a result smaller than this is copied through the pageable path: registering costs more than it saves
constexpr uint64_t kPinMinBytes = 32ull << 20;

// Are BOTH ends of [p, p + bytes) in memory the runtime knows as page-locked already (hipHostMalloc'd or
// registered)?  A buffer pinned only at its front is not "pinned".
bool host_range_is_pinned(const void *p, uint64_t bytes);

}  // namespace smmc
