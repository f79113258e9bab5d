// ow_internal.h -- what the translation units of libocean_waves.so share besides the public header: error reporting.
#pragma once
#include <hip/hip_runtime.h>

// The library is built with -fvisibility=hidden: only what include/ocean_waves.h declares is exported.
#pragma GCC visibility push(default)
#include "../../include/ocean_waves.h"
#pragma GCC visibility pop

namespace ow {

// sets the calling thread's ow_last_error() text (printf-style) and returns `st`
ow_status fail(ow_status st, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// replaces the calling thread's ow_last_error() text (a group hands a worker thread's message to its caller)
void set_last_error(const char *message);
// every record finite, tile_length positive, time + delta finite (ow_runtime.hip; sets the error text)
ow_status validate_records(const ow_cascade_params *params, int count, double delta);
// the context's device status word without synchronising (ow_runtime.hip)
ow_status poll_status(ow_context *c);

}  // namespace ow

#define OW_HIP(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return ow::fail(OW_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)
