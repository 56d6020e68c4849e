// internal helpers shared by the host-side translation units
#pragma once
#include <hip/hip_runtime.h>
#include <string>

extern thread_local std::string g_vbm_err;
int vbm_set_hip_error(hipError_t e, const char *where);
