// kernels.hip -- the HBM-bound and small kernels around the implicit GEMMs: split-K combine,
// conv2d_transpose to 3 channels, input preprocessing, losses, bias gradients, lrelu', Adam.
#include "launch.h"

namespace ctx {

// ------------------------------------------------------------------------------------------------
// split-K: out = epilogue(sum_s slab[s]); fixed summation order => deterministic
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void splitk_reduce_kernel(const Epi ep, int M, int N, int nprob, int nsplit) {
    const int64_t total = (int64_t)nprob * M * N;
    for (int64_t idx = (int64_t)blockIdx.x * NTHREADS + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * NTHREADS) {
        const int n = (int)(idx % N);
        const int64_t t = idx / N;
        const int m = (int)(t % M), prob = (int)(t / M);
        float v = 0.f;
        for (int s = 0; s < nsplit; ++s) v += ep.slab[(int64_t)s * total + idx];
        int64_t pix;
        if (epi_row(ep, prob, m, pix)) epi_store(ep, prob, pix, n, v);
    }
}

void splitk_reduce(hipStream_t s, const Epi& ep, int M, int N, int nprob, int nsplit) {
    const int64_t total = (int64_t)nprob * M * N;
    int64_t blocks = (total + NTHREADS - 1) / NTHREADS;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(NTHREADS), 0, s, ep, M, N, nprob, nsplit);
}

// ------------------------------------------------------------------------------------------------
// conv2d_transpose 5x5 s2 to 3 output channels (d_h4, arm_shaping.py:1329-1330, :1342-1343).
// N = 3 is far below an MFMA tile, so this one is a direct VALU kernel: a block owns one output
// parity class (py,px) -- all its lanes then use the same taps, whose filter slices sit in LDS and
// are read at wave-uniform addresses (broadcast).  A thread produces 4 horizontally adjacent output
// pixels of its class x 3 channels, so each filter read feeds 4 pixels.
//   out[n, 2i'+py, 2j'+px, c] = b[c] + sum_{sy,sx,k} in[n, i'+py-sy, j'+px-sx, k] * w[1-py+2sy, 1-px+2sx, c, k]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void convt3_fwd_kernel(const ConvT3Args a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [ntaps][3][cb]
    const int py = blockIdx.y >> 1, px = blockIdx.y & 1;
    const int nty = 2 + py, ntx = 2 + px, cb = a.c1 + a.c2;
    for (int idx = threadIdx.x * 4; idx < nty * ntx * 3 * cb; idx += NTHREADS * 4) {
        const int k = idx % cb, t = idx / cb, c = t % 3, tap = t / 3;
        const int sy = tap / ntx, sx = tap - sy * ntx;
        const int ky = 1 - py + 2 * sy, kx = 1 - px + 2 * sx;
        *reinterpret_cast<float4*>(&wl[idx]) = ldg4(a.w + ((int64_t)((ky * 5 + kx) * 3 + c)) * cb + k);
    }
    __syncthreads();
    const int wq = a.ws >> 2;
    const int64_t gid = (int64_t)blockIdx.x * NTHREADS + threadIdx.x;
    if (gid >= (int64_t)a.nimg * a.hs * wq) return;
    const int jq = (int)(gid % wq);
    const int64_t t = gid / wq;
    const int ip = (int)(t % a.hs), n = (int)(t / a.hs);
    float acc[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q][0] = acc[q][1] = acc[q][2] = 0.f;
    for (int sy = 0; sy < nty; ++sy) {
        const int i = ip + py - sy;
        if ((unsigned)i >= (unsigned)a.hs) continue;
        for (int sx = 0; sx < ntx; ++sx) {
            const int jb = 4 * jq + px - sx;
            const float* wt = wl + (sy * ntx + sx) * 3 * cb;
            bool ok[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) ok[q] = (unsigned)(jb + q) < (unsigned)a.ws;
            const float* r1 = a.s1 + (((int64_t)n * a.hs + i) * a.ws + jb) * a.ld1;
            const float* r2 = a.s2 + (((int64_t)(n % a.nmod2) * a.hs + i) * a.ws + jb) * a.ld2;
            for (int k = 0; k < cb; k += 4) {
                const float4 w0 = *reinterpret_cast<const float4*>(wt + k);
                const float4 w1 = *reinterpret_cast<const float4*>(wt + cb + k);
                const float4 w2 = *reinterpret_cast<const float4*>(wt + 2 * cb + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!ok[q]) continue;
                    const float4 x = k < a.c1 ? ldg4(r1 + q * a.ld1 + k) : ldg4(r2 + q * a.ld2 + (k - a.c1));
                    acc[q][0] += x.x * w0.x + x.y * w0.y + x.z * w0.z + x.w * w0.w;
                    acc[q][1] += x.x * w1.x + x.y * w1.y + x.z * w1.z + x.w * w1.w;
                    acc[q][2] += x.x * w2.x + x.y * w2.y + x.z * w2.z + x.w * w2.w;
                }
            }
        }
    }
    const float b0 = a.bias[0], b1 = a.bias[1], b2 = a.bias[2];
    float* o = a.out + (((int64_t)n * (2 * a.hs) + 2 * ip + py) * (2 * a.ws) + 2 * (4 * jq) + px) * 3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        o[q * 6 + 0] = acc[q][0] + b0;
        o[q * 6 + 1] = acc[q][1] + b1;
        o[q * 6 + 2] = acc[q][2] + b2;
    }
}

void convt3_fwd(hipStream_t s, const ConvT3Args& a) {
    const int64_t threads = (int64_t)a.nimg * a.hs * (a.ws / 4);
    dim3 grid((unsigned)((threads + NTHREADS - 1) / NTHREADS), 4);
    const size_t lds = (size_t)9 * 3 * (a.c1 + a.c2) * sizeof(float);
    hipLaunchKernelGGL(convt3_fwd_kernel, grid, dim3(NTHREADS), lds, s, a);
}

// ------------------------------------------------------------------------------------------------
// uint8 frames -> f32 in [-1,1]: convert_image_dtype (x * (1/255)), - 0.5, * 2.0 as three separately
// rounded f32 ops (rllab/sampler/base.py:116-119).  __f*_rn keeps hipcc from contracting them.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float prep_u8(uint8_t x) {
    return __fmul_rn(__fsub_rn(__fmul_rn((float)x, 1.0f / 255.0f), 0.5f), 2.0f);
}

__global__ __launch_bounds__(NTHREADS) void u8_to_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                             int64_t n, int64_t in_period) {
    // 4 elements per thread; in_period: input repeats with this period (row broadcast), multiple of 4
    for (int64_t i = ((int64_t)blockIdx.x * NTHREADS + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * NTHREADS * 4) {
        const uchar4 u = *reinterpret_cast<const uchar4*>(in + (i % in_period));
        *reinterpret_cast<float4*>(out + i) = make_float4(prep_u8(u.x), prep_u8(u.y), prep_u8(u.z), prep_u8(u.w));
    }
}

static unsigned ew_blocks(int64_t work) {
    int64_t b = (work + NTHREADS - 1) / NTHREADS;
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

void u8_to_f32(hipStream_t s, const uint8_t* in, float* out, int64_t n) {
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(ew_blocks(n / 4)), dim3(NTHREADS), 0, s, in, out, n, n);
}

void broadcast_rows_u8_to_f32(hipStream_t s, const uint8_t* in, float* out, int64_t row_elems, int64_t nrows) {
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(ew_blocks(row_elems * nrows / 4)), dim3(NTHREADS), 0, s, in, out,
                       row_elems * nrows, row_elems);
}

// ------------------------------------------------------------------------------------------------
// Losses.  recon_k = tf.nn.l2_loss(tgt - out_k) = sum(d^2)/2 over the WHOLE batch
// (arm_shaping.py:1352-1353); simloss = mean((trans_z - tgtimg_z)^2) * 1e3 (:1345).
// Wavefront shuffle reduction -> one partial per block -> fixed-order final sum in f64.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {   // sh: 4 floats; result valid in thread 0
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

// partial layout: [3][LOSS_BLOCKS]  (recon1, recon2, sim sums of squares)
__global__ __launch_bounds__(NTHREADS) void loss_partial_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                                float* __restrict__ dout, int64_t half,
                                                                const float* __restrict__ tz, const float* __restrict__ tgt_z,
                                                                float* __restrict__ dsim2, int64_t nz, float csim,
                                                                float* __restrict__ partial) {
    __shared__ float sh[4];
    const int64_t stride = (int64_t)gridDim.x * NTHREADS * 4;
    const int64_t start = ((int64_t)blockIdx.x * NTHREADS + threadIdx.x) * 4;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int64_t i = start; i < half; i += stride) {
        const float4 t = ldg4(tgt + i), a = ldg4(out + i), b = ldg4(out + half + i);
        const float4 da = make_float4(a.x - t.x, a.y - t.y, a.z - t.z, a.w - t.w);
        const float4 db = make_float4(b.x - t.x, b.y - t.y, b.z - t.z, b.w - t.w);
        s1 += da.x * da.x + da.y * da.y + da.z * da.z + da.w * da.w;
        s2 += db.x * db.x + db.y * db.y + db.z * db.z + db.w * db.w;
        if (dout) {
            *reinterpret_cast<float4*>(dout + i) = da;
            *reinterpret_cast<float4*>(dout + half + i) = db;
        }
    }
    for (int64_t i = start; i < nz; i += stride) {
        const float4 a = ldg4(tz + i), b = ldg4(tgt_z + i);
        const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        s3 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        if (dsim2) {
            *reinterpret_cast<float4*>(dsim2 + i) = make_float4(csim * d.x, csim * d.y, csim * d.z, csim * d.w);
            *reinterpret_cast<float4*>(dsim2 + nz + i) = make_float4(-csim * d.x, -csim * d.y, -csim * d.z, -csim * d.w);
        }
    }
    const float r1 = block_sum(s1, sh), r2 = block_sum(s2, sh), r3 = block_sum(s3, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = r1;
        partial[LOSS_BLOCKS + blockIdx.x] = r2;
        partial[2 * LOSS_BLOCKS + blockIdx.x] = r3;
    }
}

__global__ void loss_final_kernel(const float* __restrict__ partial, int nblk, double inv_nz, float* __restrict__ scalars) {
    if (threadIdx.x != 0) return;
    double a = 0, b = 0, c = 0;
    for (int i = 0; i < nblk; ++i) { a += partial[i]; b += partial[LOSS_BLOCKS + i]; c += partial[2 * LOSS_BLOCKS + i]; }
    const double r1 = 0.5 * a, r2 = 0.5 * b, sim = c * inv_nz * 1e3;
    scalars[0] = (float)(r1 + r2 + sim);
    scalars[1] = (float)sim;
    scalars[2] = (float)r1;
    scalars[3] = (float)r2;
}

void losses(hipStream_t s, const float* out, const float* tgt, float* dout, int64_t npi, int B, const float* tz,
            const float* tgt_z, float* dsim2, int F, int sim_batch, float* scratch, float* scalars) {
    const int64_t half = npi * B, nz = (int64_t)B * F;
    int64_t blocks = (half / 4 + NTHREADS - 1) / NTHREADS;
    if (blocks > LOSS_BLOCKS) blocks = LOSS_BLOCKS;
    if (blocks < 1) blocks = 1;
    const float csim = (float)(2e3 / ((double)sim_batch * F));
    hipLaunchKernelGGL(loss_partial_kernel, dim3((unsigned)blocks), dim3(NTHREADS), 0, s, out, tgt, dout, half, tz, tgt_z,
                       dsim2, nz, csim, scratch);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, (const float*)scratch, (int)blocks, 1.0 / (double)nz,
                       scalars);
}

// ------------------------------------------------------------------------------------------------
// Bias gradient: column sums of a row-major [rows, C] gradient.  Stage 1: a block covers 64 columns
// x one row slice (4 row lanes, coalesced 256-B row reads); stage 2 adds the slices in fixed order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void colsum_partial_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                                  int64_t rows_per, float* __restrict__ part) {
    __shared__ float sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per;
    int64_t r1 = r0 + rows_per;
    if (r1 > rows) r1 = rows;
    float acc = 0.f;
    if (c < C)
        for (int64_t r = r0 + rl; r < r1; r += 4) acc += x[r * C + c];
    sh[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < C) part[(int64_t)blockIdx.y * C + c] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__global__ __launch_bounds__(NTHREADS) void colsum_final_kernel(const float* __restrict__ part, int nsl, int C,
                                                                float* __restrict__ out) {
    const int c = blockIdx.x * NTHREADS + threadIdx.x;
    if (c >= C) return;
    float acc = 0.f;
    for (int sl = 0; sl < nsl; ++sl) acc += part[(int64_t)sl * C + c];
    out[c] = acc;
}

void colsum(hipStream_t s, const float* x, int64_t rows, int C, float* scratch, float* out) {
    int nsl = (int)((rows + 255) / 256);
    if (nsl > COLSUM_SPLITS) nsl = COLSUM_SPLITS;
    if (nsl < 1) nsl = 1;
    const int64_t rows_per = (rows + nsl - 1) / nsl;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, nsl), dim3(NTHREADS), 0, s, x, rows, C, rows_per, scratch);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + NTHREADS - 1) / NTHREADS), dim3(NTHREADS), 0, s,
                       (const float*)scratch, nsl, C, out);
}

// ------------------------------------------------------------------------------------------------
// lrelu' on the saved output: d/dx max(x, 0.2x) = 1 for x >= 0 else 0.2; sign(y) == sign(x)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void lrelu_mask_kernel(float* __restrict__ g, const float* __restrict__ act, int64_t n) {
    for (int64_t i = ((int64_t)blockIdx.x * NTHREADS + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * NTHREADS * 4) {
        float4 v = ldg4(g + i);
        const float4 a = ldg4(act + i);
        v.x *= a.x >= 0.f ? 1.f : LEAK; v.y *= a.y >= 0.f ? 1.f : LEAK;
        v.z *= a.z >= 0.f ? 1.f : LEAK; v.w *= a.w >= 0.f ? 1.f : LEAK;
        *reinterpret_cast<float4*>(g + i) = v;
    }
}

void lrelu_mask(hipStream_t s, float* g, const float* act, int64_t n) {
    hipLaunchKernelGGL(lrelu_mask_kernel, dim3(ew_blocks(n / 4)), dim3(NTHREADS), 0, s, g, act, n);
}

// ------------------------------------------------------------------------------------------------
// Fused multi-tensor Adam over the flat arena, TF formulation (eps outside the bias correction):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t m / (sqrt(v) + eps)
// 7 arena passes of HBM traffic (read p,g,m,v; write p,m,v) in one launch.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, float lr_t, float b1, float b2,
                                                        float eps) {
    const float c1 = 1.f - b1, c2 = 1.f - b2;
    for (int64_t i = ((int64_t)blockIdx.x * NTHREADS + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * NTHREADS * 4) {
        const float4 gg = ldg4(g + i);
        float4 mm = ldg4(m + i), vv = ldg4(v + i), pp = ldg4(p + i);
#define CTX_ADAM1(f)                                   \
    mm.f = b1 * mm.f + c1 * gg.f;                      \
    vv.f = b2 * vv.f + c2 * (gg.f * gg.f);             \
    pp.f = pp.f - lr_t * mm.f / (sqrtf(vv.f) + eps);
        CTX_ADAM1(x) CTX_ADAM1(y) CTX_ADAM1(z) CTX_ADAM1(w)
#undef CTX_ADAM1
        *reinterpret_cast<float4*>(m + i) = mm;
        *reinterpret_cast<float4*>(v + i) = vv;
        *reinterpret_cast<float4*>(p + i) = pp;
    }
}

void adam(hipStream_t s, float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2,
          float eps) {
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n / 4)), dim3(NTHREADS), 0, s, p, g, m, v, n, lr_t, b1, b2, eps);
}

}  // namespace ctx
