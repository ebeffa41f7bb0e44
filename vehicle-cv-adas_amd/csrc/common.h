// common.h -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/adas_hip.h"

namespace adas {
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
// Parameters that a captured pipeline step bakes into its kernel arguments (post / decode / geometry configuration) carry a
// process-wide generation: every setter bumps it, adas_pipeline_* re-captures when it moved since the capture.
unsigned long long config_generation();
void bump_config_generation();
}  // namespace adas

// frames a post-processing handle was created for (post_kernels.hip): adas_pipeline_create checks them against n_streams x micro_batch
struct adas_yolo_post;
struct adas_ufld_decode;
struct adas_lane_geometry;
namespace adas {
int handle_max_batch(const ::adas_yolo_post* h);
int handle_max_batch(const ::adas_ufld_decode* h);
int handle_max_batch(const ::adas_lane_geometry* h);
}  // namespace adas

#define ADAS_HIP_TRY(expr)                                                      \
    do {                                                                        \
        hipError_t e__ = (expr);                                                \
        if (e__ != hipSuccess) return adas::hip_fail(e__, #expr, __FILE__, __LINE__); \
    } while (0)

#define ADAS_REQUIRE(cond, code, ...)      \
    do {                                   \
        if (!(cond)) {                     \
            adas::set_error(__VA_ARGS__);  \
            return (code);                 \
        }                                  \
    } while (0)
