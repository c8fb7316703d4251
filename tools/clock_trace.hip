// tools/clock_trace.hip -- the shader clock UNDER LOAD, as a trace ("0.88 of nominal" needs evidence or a retraction).
//
// Every launch, one lane per block samples s_memtime (shader-clock ticks, MI355X_MICROARCH.md "tick = shader cycle") and
// s_memrealtime (constant 100 MHz) at its first and last instruction: effective clock of that launch = d(memtime) / d(realtime) * 100 MHz.
// Launches of ~4 ms are repeated back to back for SECONDS seconds per body, so the table shows the clock settling under
// sustained load.  Bodies:
//   mfma32     nothing but v_mfma_f32_32x32x2_f32, four independent accumulators per wave, 8 waves per CU   (tools/peaks.hip's loop,
//              but on RANDOM operands that change from one MFMA to the next)
//   mfma16     the same with v_mfma_f32_16x16x4_f32 (the direct kernels' instruction)
//   mfma32+ld  the MFMA stream with, per 8 MFMAs, two 16-byte global loads (L2-resident 8 MB window), two ds_write_b128 and four
//              ds_read_b128 per lane -- the operand feed of an implicit-GEMM step, results folded into the MFMA operands
//   mfma32+ld+hbm  the same with the loads streaming a 2 GiB buffer (HBM traffic beside the matrix pipe)
// Not part of the product.   hipcc --offload-arch=gfx950 -O3 tools/clock_trace.hip -o tools/clock_trace.bin ; tools/clock_trace.bin [seconds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Stamp { uint64_t c0, c1, r0, r1; };

__global__ void fill_random(float* p, int64_t n) {          // values in [-1, 1)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int)(h & 0xFFFF) * (1.f / 32768.f) - 1.f;
    }
}

template <int BODY>
__global__ __launch_bounds__(512) void body_kernel(Stamp* st, const float4* __restrict__ src, int64_t n4, int iters, float* sink) {
    __shared__ float4 lds[512 * 2];
    const int tid = threadIdx.x;
    uint64_t c0 = 0, r0 = 0;
    if (tid == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    // RANDOM operands, four different ones per lane in rotation: the matrix pipe's inputs toggle from one MFMA to the next as they do
    // on real activations (constant operands draw less power and hold a higher clock -- MI355X_MICROARCH.md, "DVFS give-back")
    auto rnd = [&](unsigned k) { unsigned h = (blockIdx.x * 512u + tid) * 2654435761u + k * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return (float)(int)(h & 0xFFFF) * (1.f / 32768.f) - 1.f; };
    float x = rnd(1), y = rnd(2), z = rnd(3), w = rnd(4);
    float s = 0.f;
    if constexpr (BODY == 1) {
        f32x4 a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {}, a6 = {}, a7 = {};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(z, w, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, z, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, z, a4, 0, 0, 0); a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, w, a5, 0, 0, 0);
            a6 = __builtin_amdgcn_mfma_f32_16x16x4f32(z, x, a6, 0, 0, 0); a7 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, y, a7, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) s += a0[r] + a1[r] + a2[r] + a3[r] + a4[r] + a5[r] + a6[r] + a7[r];
    } else {
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        int64_t idx = ((int64_t)blockIdx.x * 512 + tid) % n4;
        const int64_t step = (int64_t)gridDim.x * 512;
        for (int i = 0; i < iters; ++i) {
            if constexpr (BODY >= 2) {
                if ((i & 1) == 0) {
                    const float4 u = src[idx], v = src[(idx + step / 2) % n4];
                    idx += step; if (idx >= n4) idx -= n4;
                    lds[tid] = u; lds[512 + tid] = v;
                    __syncthreads();
                    const float4 p = lds[(tid + 64) & 511], q = lds[512 + ((tid + 128) & 511)], r = lds[(tid + 192) & 511], ww = lds[512 + ((tid + 256) & 511)];
                    x = x * 0.5f + (p.x + q.y) * 0.25f; y = y * 0.5f + (r.z + ww.w) * 0.25f;
                    __syncthreads();
                }
            }
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(z, w, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, z, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, a3, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    }
    if (s == 12345.678f) sink[0] = s;
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        st[blockIdx.x] = Stamp{c0, (uint64_t)__builtin_readcyclecounter(), r0, (uint64_t)__builtin_amdgcn_s_memrealtime()};
    }
}

template <int BODY>
static void run(const char* name, double seconds, int blocks, const float4* src, int64_t n4, int iters, double flop_per_iter_block) {
    Stamp* st; float* sink;
    (void)hipMalloc((void**)&st, blocks * sizeof(Stamp)); (void)hipMalloc((void**)&sink, 64);
    std::vector<Stamp> h(blocks);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct Row { double t, mhz_med, mhz_min, mhz_max, tf; };
    std::vector<Row> rows;
    double t = 0;
    while (t < seconds) {
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(body_kernel<BODY>, dim3(blocks), dim3(512), 0, nullptr, st, src, n4, iters, sink);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
        std::vector<double> f(blocks);
        for (int b = 0; b < blocks; ++b) f[b] = (double)(h[b].c1 - h[b].c0) / (double)(h[b].r1 - h[b].r0) * 100.0;
        std::sort(f.begin(), f.end());
        t += ms * 1e-3;
        rows.push_back({t, f[blocks / 2], f[0], f[blocks - 1], flop_per_iter_block * iters * blocks / (ms * 1e-3) / 1e12});
    }
    printf("\n== %s: %zu launches of %.2f ms, %d blocks x 512 threads\n   t [s]   clock MHz (median / min / max over blocks)   TF/s\n", name, rows.size(), rows[rows.size() / 2].t / (rows.size() / 2 + 1) * 1e3, blocks);
    const size_t stride = std::max<size_t>(1, rows.size() / 24);
    for (size_t i = 0; i < rows.size(); i += stride) printf("  %6.3f   %7.1f / %7.1f / %7.1f   %7.1f\n", rows[i].t, rows[i].mhz_med, rows[i].mhz_min, rows[i].mhz_max, rows[i].tf);
    // settled = the second half
    double m = 0, tf = 0; size_t n = 0;
    for (size_t i = rows.size() / 2; i < rows.size(); ++i) { m += rows[i].mhz_med; tf += rows[i].tf; ++n; }
    printf("   settled (second half): %.1f MHz = %.3f of 2400, %.1f TF/s = %.3f of 157.3\n", m / n, m / n / 2400.0, tf / n, tf / n / 157.3);
    (void)hipFree(st); (void)hipFree(sink);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int blocks = pr.multiProcessorCount;
    printf("%s, %d CUs, clockRate %d kHz\n", pr.name, blocks, pr.clockRate);
    const int64_t small = 8ll << 20, big = 2ll << 30;
    float4* buf; if (hipMalloc((void**)&buf, big) != hipSuccess) return 1;
    fill_random<<<4096, 256>>>(reinterpret_cast<float*>(buf), big / 4);
    const double f32 = 8.0 * 4 * (2.0 * 32 * 32 * 2), f16 = 8.0 * 8 * (2.0 * 16 * 16 * 4);
    run<0>("mfma32", seconds, blocks, buf, small / 16, 36000, f32);
    run<1>("mfma16", seconds, blocks, buf, small / 16, 36000, f16);
    run<2>("mfma32+ld (L2-resident window)", seconds, blocks, buf, small / 16, 36000, f32);
    run<3>("mfma32+ld+hbm (2 GiB stream)", seconds, blocks, buf, big / 16, 36000, f32);
    (void)hipFree(buf);
    return 0;
}
