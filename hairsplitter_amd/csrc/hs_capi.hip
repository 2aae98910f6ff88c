// hs_capi.hip -- the C ABI of include/hairsplitter_hip.h: device memory, kernel launches and the two
// stage drivers. ONE translation unit (the kernels are templates and inline device code shared by all of it), kept in parts:
//   hs_dev_runtime.inc   pools, transfers as kernels, waits, events, kernel accounting
//   hs_capi_kernels.inc  small C ABI + kernel-level entry points
//   hs_cv_backend.inc    stage 3 on the device (hs_cv_batch, HipCvOps)
//   hs_sr_backend.inc    stage 4 on the device (GraphRows, HipSrOps)
//   hs_capi_stage.inc    stage-level C ABI on one device, contig groups (hs_pipeline_*)
//   hs_capi_multi.inc    several GPUs in one process There is no CPU fallback anywhere in this file: without a usable HIP device every entry
// point fails with HS_ENODEVICE.
#include <hip/hip_runtime.h>
#include <malloc.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <condition_variable>
#include <functional>
#include <thread>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hs_host.h"
#include "hs_host_sr.h"
#include "hs_driver.h"
#include "hs_kernels.hip"
#include "hs_kernels_graph.hip"
#include "hs_kernels_cw.hip"
#include "hs_kernels_myers.hip"
#include "hs_kernels_cols.hip"
#include "hs_kernels_loopa.hip"

namespace hs {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace hs

using hs::set_error;

#define HS_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t e__ = (call);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            set_error(std::string(#call) + ": " + hipGetErrorString(e__));                               \
            return e__ == hipErrorNoDevice || e__ == hipErrorInvalidDevice ? HS_ENODEVICE : HS_EHIP;     \
        }                                                                                                \
    } while (0)

#include "hs_dev_runtime.inc"
#include "hs_capi_kernels.inc"
#include "hs_cv_backend.inc"
#include "hs_sr_backend.inc"
#include "hs_capi_stage.inc"
#include "hs_capi_multi.inc"
