// internal helpers shared by the host-side translation units
#pragma once
#include <hip/hip_runtime.h>
#include <string>

extern thread_local std::string g_vbm_err;
int vbm_set_hip_error(hipError_t e, const char *where);

// test instrumentation (debug_hooks.hip): a spin kernel on `q` when `point` is in the mask of vbm_debug_set_delay
enum vbm_delay_point {
    VBM_DP_JOB_BIG = 0,        // host-built round: before the first kernel of a big batch (its own front stream)
    VBM_DP_JOB_SMALL = 1,      // ... of a small batch (stream of its block type)
    VBM_DP_JOB_STATE = 2,      // before the first kernel of a batch that touches the carried stream state
    VBM_DP_JOB_BACK = 3,       // before a batch's back half (floor fit .. packets)
    VBM_DP_JOB_OUT = 4,        // before a batch's outputs are copied to the caller's buffers
    VBM_DP_FE_FORK = 5,        // front end: between the gathers of a round and the round's fork
    VBM_DP_FE_SHIFT = 6,       // front end: before the buffer shift of a round
    VBM_DP_DEV_BIG_FRONT = 7,  // device-built round: before the big batch's front-half graph
    VBM_DP_DEV_BIG_BACK = 8,   // ... back-half graph
    VBM_DP_DEV_SMALL_FRONT = 9,
    VBM_DP_DEV_SMALL_BACK = 10,
    VBM_DP_DEV_PLAN = 11,      // front end: before a device-built round is planned
    VBM_DP_BATCH_FRONT = 12,   // vbm_analysis_batch2: before the front half
    VBM_DP_BATCH_BACK = 13,    // ... the back half
    VBM_DP_FE_WRITE = 14,      // front end: before the append of a write
    VBM_DP_DEV_OUT = 15,       // device-built round: before a batch's output copy
};
void vbm_debug_delay_point(int point, hipStream_t q);
