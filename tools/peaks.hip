// tools/peaks.hip -- the two rooflines of SURVEY.md 8(d), MEASURED on the box bench.py runs on: an f32 matrix-core loop (nothing but
// v_mfma_f32_32x32x2_f32 on four independent accumulators per wave, eight waves per CU) and a device-to-device stream copy.  bench.py
// loads tools/libpeaks.so when it is there and reports both beside the nominal 157.3 TF/s / 8 TB/s (`peaks_measured`).  Not part of
// the product library.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/peaks.hip -o tools/libpeaks.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    const float x = 1.0f + threadIdx.x * 1e-6f, y = 1.0f - threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.678f) out[0] = s;                        // (keeps the loop)
}

__global__ __launch_bounds__(256) void copy_loop(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const float __attribute__((ext_vector_type(4)))*>(src + i)),
                                    reinterpret_cast<float __attribute__((ext_vector_type(4)))*>(dst + i));
}

static float timed(hipStream_t s, int reps, void (*launch)(hipStream_t, void*), void* arg) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(s, arg);                                          // warm-up
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) launch(s, arg);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / reps;
}

struct MfmaArg { float* out; int iters, blocks; };
struct CopyArg { const float4* src; float4* dst; int64_t n4; int blocks; };

extern "C" {

// TF/s of the f32 matrix-core loop over the whole chip (0 on failure)
double peak_mfma_f32_tflops(void) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) return 0.0;
    MfmaArg a{nullptr, 20000, pr.multiProcessorCount};
    if (hipMalloc((void**)&a.out, 64) != hipSuccess) return 0.0;
    const float ms = timed(nullptr, 5, [](hipStream_t s, void* p) { auto* q = (MfmaArg*)p; hipLaunchKernelGGL(mfma_loop, dim3(q->blocks), dim3(512), 0, s, q->out, q->iters); }, &a);
    (void)hipFree(a.out);
    const double flop = (double)a.blocks * 8 /* waves */ * a.iters * 4 * (2.0 * 32 * 32 * 2);
    return ms > 0 ? flop / (ms * 1e-3) / 1e12 : 0.0;
}

// GB/s (read + write) of a device-to-device stream copy of `bytes` (0 on failure)
double peak_hbm_copy_gbps(int64_t bytes) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) return 0.0;
    CopyArg a{nullptr, nullptr, bytes / 16, pr.multiProcessorCount * 8};
    if (hipMalloc((void**)&a.src, bytes) != hipSuccess) return 0.0;
    if (hipMalloc((void**)&a.dst, bytes) != hipSuccess) { (void)hipFree((void*)a.src); return 0.0; }
    (void)hipMemset((void*)a.src, 1, bytes);
    const float ms = timed(nullptr, 10, [](hipStream_t s, void* p) { auto* q = (CopyArg*)p; hipLaunchKernelGGL(copy_loop, dim3(q->blocks), dim3(256), 0, s, q->src, q->dst, q->n4); }, &a);
    (void)hipFree((void*)a.src); (void)hipFree(a.dst);
    return ms > 0 ? 2.0 * (double)bytes / (ms * 1e-3) / 1e9 : 0.0;
}

}  // extern "C"
