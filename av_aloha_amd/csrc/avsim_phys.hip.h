// TEMPORARY stub (replaced by the physics kernels): lets the IK path build and run on the GPU first.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "avsim_model.h"
namespace avs {
struct PhysHost {
    int maxcon = 48, maxefc = 192;
    double *d_qpos_home = nullptr, *d_ctrl_home = nullptr;
    int* d_obj_qadr = nullptr;
    int* d_ncon = nullptr; int* d_cpairs = nullptr; double* d_cdist = nullptr; int* d_diag = nullptr;
    bool init(const Blob& b, int N, bool f64, std::string& err) {
        auto qh = b.f("qpos_home"), ch = b.f("ctrl_home"); auto oa = b.i("objects_qposadr");
        hipMalloc((void**)&d_qpos_home, qh.size() * 8); hipMemcpy(d_qpos_home, qh.data(), qh.size() * 8, hipMemcpyHostToDevice);
        hipMalloc((void**)&d_ctrl_home, ch.size() * 8); hipMemcpy(d_ctrl_home, ch.data(), ch.size() * 8, hipMemcpyHostToDevice);
        hipMalloc((void**)&d_obj_qadr, oa.size() * 4); hipMemcpy(d_obj_qadr, oa.data(), oa.size() * 4, hipMemcpyHostToDevice);
        hipMalloc((void**)&d_ncon, (size_t)N * 4); hipMalloc((void**)&d_cpairs, (size_t)N * maxcon * 8);
        hipMalloc((void**)&d_cdist, (size_t)N * maxcon * 8); hipMalloc((void**)&d_diag, (size_t)N * 16);
        return true;
    }
    void destroy() {}
    bool set_option(const char*, double) { return false; }
    int launch(hipStream_t, int, int, const float*, int, void*, void*, void*, void*, int*, double*, int32_t*, uint8_t*, std::string&) { return 0; }
};
}
