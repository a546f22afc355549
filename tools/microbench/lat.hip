// Cost model probes for k_phys-like code on gfx950: cycles per operation for one wave, with 8 waves per CU (one 512-thread block
// per CU held there by 150 KB of LDS) all doing the same thing.  Build: hipcc --offload-arch=gfx950 -O3 -o lat lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_OPS 64
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, const float* gsrc, int mode, int nidx) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* my = lds + wave * 4096;
    for (int i = lane; i < 4096; i += 64) my[i] = (float)i;
    __syncthreads();
    float x = (float)lane, y = 1.0001f;
    int idx = lane;
    long long t0 = __builtin_readcyclecounter();
    if (mode == 0) {            // dependent v_fma chain
#pragma unroll
        for (int i = 0; i < N_OPS; i++) x = x * y + 1.0f;
    } else if (mode == 1) {     // LDS atomics, distinct addresses (lane -> own word)
#pragma unroll
        for (int i = 0; i < N_OPS; i++) __hip_atomic_fetch_add(my + lane + 64 * (i & 7), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (mode == 2) {     // LDS atomics, all lanes onto `nidx` addresses
#pragma unroll
        for (int i = 0; i < N_OPS; i++) __hip_atomic_fetch_add(my + (lane % nidx) + 64 * (i & 7), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (mode == 3) {     // dependent ds_bpermute chain
#pragma unroll
        for (int i = 0; i < N_OPS; i++) x = __shfl(x, (lane + 1) & 63, 64) + 1.0f;
    } else if (mode == 4) {     // independent ds_bpermute
        float acc = 0;
#pragma unroll
        for (int i = 0; i < N_OPS; i++) acc += __shfl(x, (lane + i) & 63, 64);
        x = acc;
    } else if (mode == 5) {     // dependent LDS read chain
#pragma unroll
        for (int i = 0; i < N_OPS; i++) { idx = (int)my[idx & 4095]; }
        x = (float)idx;
    } else if (mode == 6) {     // dependent global load chain (L2 hits: small array shared by all)
#pragma unroll
        for (int i = 0; i < N_OPS; i++) { idx = (int)gsrc[idx & 1023]; }
        x = (float)idx;
    } else if (mode == 7) {     // plain LDS read-modify-write, distinct addresses
#pragma unroll
        for (int i = 0; i < N_OPS; i++) my[lane + 64 * (i & 7)] += x;
    } else if (mode == 8) {     // v_readlane + fma chain
#pragma unroll
        for (int i = 0; i < N_OPS; i++) x += y * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), i & 63));
    } else if (mode == 9) {     // independent v_fma (4 chains)
        float a = x, b = x + 1, c = x + 2, d = x + 3;
#pragma unroll
        for (int i = 0; i < N_OPS / 4; i++) { a = a * y + 1.0f; b = b * y + 1.0f; c = c * y + 1.0f; d = d * y + 1.0f; }
        x = a + b + c + d;
    } else if (mode == 10) {    // DPP add chain
#pragma unroll
        for (int i = 0; i < N_OPS; i++) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xb1, 0xf, 0xf, false));
    } else if (mode == 11) {    // independent LDS reads (16 per batch)
        float acc = 0;
#pragma unroll
        for (int i = 0; i < N_OPS; i++) acc += my[(lane + 65 * i) & 4095];
        x = acc;
    }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = x + my[lane];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
    const int nb = 256;
    float *out, *gsrc; long long* cyc;
    hipMalloc(&out, nb * 512 * 4); hipMalloc(&cyc, nb * 8 * 8); hipMalloc(&gsrc, 1024 * 4);
    std::vector<float> h(1024); for (int i = 0; i < 1024; i++) h[i] = (float)((i * 37 + 11) & 1023);
    hipMemcpy(gsrc, h.data(), 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const char* names[] = {"dependent v_fma", "ds_add_f32 distinct", "ds_add_f32 onto n addresses", "dependent ds_bpermute", "independent ds_bpermute", "dependent LDS read", "dependent global load (L2)", "LDS += distinct", "v_readlane + fma chain", "4 independent v_fma chains", "dependent DPP add", "independent LDS reads"};
    for (int mode = 0; mode < 12; mode++)
        for (int nidx : {64, 16, 8, 1}) {
            if (mode != 2 && nidx != 64) continue;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k, dim3(nb), dim3(512), 150 * 1024, 0, out, cyc, gsrc, mode, nidx);
                hipDeviceSynchronize();
            }
            std::vector<long long> c(nb * 8); hipMemcpy(c.data(), cyc, nb * 64, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : c) s += (double)v;
            printf("%-32s n=%2d  %.1f cycles per op (8 waves per CU)\n", names[mode], nidx, s / c.size() / N_OPS);
        }
    return 0;
}
