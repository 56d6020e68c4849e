// internal: host-side encoder setup (setup_host.cpp)
#pragma once
#include <string>
#include "setup.h"

struct vbm_setup_host;
vbm_setup_host *vbm_setup_host_load(const char *common_path, const char *mode_path, std::string &err);
const vbm_setup *vbm_setup_host_view(const vbm_setup_host *H);   // host pointers
int vbm_setup_host_upload(vbm_setup_host *H);                    // 0 or VBM_E*
const vbm_setup *vbm_setup_device(const vbm_setup_host *H);      // device address of the device copy
const vbm_setup *vbm_setup_device_ptrs(const vbm_setup_host *H); // host struct holding device pointers
void vbm_setup_host_free(vbm_setup_host *H);
