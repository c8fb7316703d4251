// igemm.h -- the one hot kernel of the translator: an LDS-staged, im2col-free implicit GEMM on the
// exact-f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain).
//
// Every dense contraction of ContextSkipNew (gym/envs/mujoco/arm_shaping.py:1272-1354) and of its
// gradient is D[m][n] = sum_k A[m][k] * B[k][n] for some VIEW of NHWC activations / HWIO filters:
//   conv2d 5x5 s2 SAME (arm_shaping.py:21-32)     m=(img,i,j)   k=(ky,kx,cin)        n=cout
//   conv2d_transpose (arm_shaping.py:62-85)        m=(img,y,x) of one output parity class
//                                                  k=(valid taps, cin of [decoder|skip])  n=cout
//   filter gradient of either                      m=c_big  k=(img,i,j)  n=c_small   (x25 taps)
//   linear (arm_shaping.py:48-59) fwd / dX / dW    plain GEMM views
// The views are "loaders": structs that hand the kernel one float4 of the virtual A or B matrix.
// Nothing is materialised in HBM: each block gathers 128-B channel runs of the pixels it needs
// straight into LDS (coalesced), and the 25-tap reuse comes from L2.
//
// Block = 256 threads = 4 waves (2 x 2), block tile (64*MI) x (64*NI), K staged KC = 32 at a time.
// LDS tile formats (both conflict-free, guide section 2 / Guideline 4):
//   KM ("k-minor"): tile[row][KC+4]; a lane reads 4 consecutive k with one ds_read_b128 (row
//       stride 36 dwords spreads a 16-lane group over all 64 banks) and feeds 4 MFMAs.
//   NM ("row-minor"): tile[k][rows]; 32 lanes read 32 consecutive floats with ds_read_b32.
// MFMA k-order inside a chunk: step (q,t), lanes 0-31 supply k = 8q+t, lanes 32-63 k = 8q+4+t --
// the same for A and B, so any permutation is legal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ctx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;
constexpr int LDK = KC + 4;
constexpr int NTHREADS = 256;
constexpr float LEAK = 0.2f;  // arm_shaping.py:18

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ------------------------------------------------------------------------------------------------
// Epilogue: what happens to D.  One struct serves forward (bias + lrelu), backward (skip-gradient
// adds + lrelu' mask from the saved activation), the concat split of the decoder's input gradient
// and split-K partial slabs.
// ------------------------------------------------------------------------------------------------
struct Epi {
    float* out1 = nullptr;          // cols [0, nsplit)
    int64_t ld1 = 0;
    float* out2 = nullptr;          // cols [nsplit, N) -> out2[pix*ld2 + n - nsplit], no mask
    int64_t ld2 = 0;
    int nsplit = 1 << 30;
    const float* bias = nullptr;    // [N]
    const float* add1 = nullptr;    // same pixel mapping as out, own ld
    int64_t lda1 = 0;
    const float* add2 = nullptr;
    int64_t lda2 = 0;
    const float* mask = nullptr;    // saved lrelu OUTPUT; v *= (mask >= 0 ? 1 : 0.2) for n < nsplit
    int64_t ldm = 0;
    int lrelu = 0;                  // v = max(v, 0.2 v) after bias/adds
    int rowmode = 0;                // 0: pix = m; 1: transposed-conv parity class; 2: (ky,e) rows of a C=3 filter
    int hs = 0, ws = 0;             // rowmode 1: small-grid size
    int64_t prob_stride = 0;        // out1 += prob * prob_stride (filter gradient: one tap per problem)
    float* slab = nullptr;          // split-K: raw partials to slab[((split*nprob+prob)*M + m)*N + n]
};

// returns false if the row has no destination
__device__ __forceinline__ bool epi_row(const Epi& e, int prob, int m, int64_t& pix) {
    if (e.rowmode == 0) { pix = m; return true; }
    if (e.rowmode == 1) {
        const int py = prob >> 1, px = prob & 1;
        const int j = m % e.ws, t = m / e.ws, i = t % e.hs, n = t / e.hs;
        pix = ((int64_t)n * (2 * e.hs) + 2 * i + py) * (2 * e.ws) + 2 * j + px;
        return true;
    }
    const int ky = m >> 4, el = m & 15;
    if (ky >= 5 || el == 15) return false;
    pix = ky * 15 + el;
    return true;
}

__device__ __forceinline__ void epi_store(const Epi& e, int prob, int64_t pix, int n, float v) {
    if (e.bias) v += e.bias[n];
    if (e.add1) v += e.add1[pix * e.lda1 + n];
    if (e.add2) v += e.add2[pix * e.lda2 + n];
    if (e.lrelu) v = fmaxf(v, LEAK * v);
    if (n < e.nsplit) {
        if (e.mask) v *= (e.mask[pix * e.ldm + n] >= 0.f) ? 1.f : LEAK;
        e.out1[(int64_t)prob * e.prob_stride + pix * e.ld1 + n] = v;
    } else {
        e.out2[pix * e.ld2 + (n - e.nsplit)] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Loaders.  KM loaders: prep(prob,row,ctx) once, then load(ctx,prob,chunk,k4) -> 4 consecutive k.
//           NM loaders: load(prob,chunk,kk,r4) -> rows r4..r4+3 at k = chunk*32+kk.
// All return zeros outside the virtual matrix (SAME padding, ragged M/N/K).
// ------------------------------------------------------------------------------------------------

// Plain row-major matrix V[row][k], optionally split along k into two buffers (the translate MLP's
// concat([src_z, ctx_z]), arm_shaping.py:1310).
struct KmPlain {
    static constexpr bool KM = true;
    const float* p0; int64_t ld0;
    const float* p1; int64_t ld1;
    int ksplit;      // k < ksplit -> p0, else p1[k - ksplit]
    int R;           // valid rows
    int nchunks;
    struct Ctx { int64_t row; bool ok; };
    __device__ int nchunks_of(int) const { return nchunks; }
    __device__ void prep(int, int row, Ctx& c) const { c.row = row; c.ok = row < R; }
    __device__ float4 load(const Ctx& c, int, int chunk, int k4) const {
        if (!c.ok) return zero4();
        const int k = chunk * KC + k4;
        return k < ksplit ? ldg4(p0 + c.row * ld0 + k) : ldg4(p1 + c.row * ld1 + (k - ksplit));
    }
};

// conv2d forward operand: row m = (img, i, j) of the OUTPUT grid; segment = tap (ky,kx); the run is
// the cin channels of input pixel (2i+ky-1, 2j+kx-1)  [TF SAME for k=5, s=2, even input: pad 1 / 2].
struct KmConvGather {
    static constexpr bool KM = true;
    const float* x; int64_t ldx;   // NHWC input, channel stride ldx
    int hb, wb, hs, ws;            // input (big) and output (small) grids
    int cps;                       // chunks per tap = cin / 32
    int R;                         // imgs * hs * ws
    struct Ctx { int64_t img_base; int i2, j2; bool ok; };
    __device__ int nchunks_of(int) const { return 25 * cps; }
    __device__ void prep(int, int row, Ctx& c) const {
        c.ok = row < R;
        const int j = row % ws, t = row / ws, i = t % hs, n = t / hs;
        c.img_base = (int64_t)n * hb * wb; c.i2 = 2 * i - 1; c.j2 = 2 * j - 1;
    }
    __device__ float4 load(const Ctx& c, int, int chunk, int k4) const {
        const int seg = chunk / cps, kc = (chunk - seg * cps) * KC;
        const int ky = seg / 5, kx = seg - ky * 5;
        const int y = c.i2 + ky, xx = c.j2 + kx;
        if (!c.ok || (unsigned)y >= (unsigned)hb || (unsigned)xx >= (unsigned)wb) return zero4();
        return ldg4(x + (c.img_base + (int64_t)y * wb + xx) * ldx + kc + k4);
    }
};

// conv2d_transpose operand for output parity class prob = (py,px): row m = (img, i', j') with output
// pixel (2i'+py, 2j'+px); valid taps ky = 1-py+2sy (sy < 2+py), input pixel i = i'+py-sy (same in x).
// Input channels come from two tensors [decoder | ctx skip]; the skip is shared by both decoder
// passes, so its image index is img % nmod2 (arm_shaping.py:1323 and :1336 use the same tgtctx_h*).
struct KmConvTGather {
    static constexpr bool KM = true;
    const float* s1; int64_t ld1; int c1;
    const float* s2; int64_t ld2; int nmod2;
    int hs, ws;
    int cps;                       // (c1 + c2) / 32
    int R;
    struct Ctx { int n, i, j; bool ok; };
    __device__ int nchunks_of(int prob) const { return (2 + (prob >> 1)) * (2 + (prob & 1)) * cps; }
    __device__ void prep(int, int row, Ctx& c) const {
        c.ok = row < R;
        c.j = row % ws; const int t = row / ws; c.i = t % hs; c.n = t / hs;
    }
    __device__ float4 load(const Ctx& c, int prob, int chunk, int k4) const {
        const int py = prob >> 1, px = prob & 1, ntx = 2 + px;
        const int seg = chunk / cps, k = (chunk - seg * cps) * KC + k4;
        const int sy = seg / ntx, sx = seg - sy * ntx;
        const int i = c.i + py - sy, j = c.j + px - sx;
        if (!c.ok || (unsigned)i >= (unsigned)hs || (unsigned)j >= (unsigned)ws) return zero4();
        if (k < c1) return ldg4(s1 + (((int64_t)c.n * hs + i) * ws + j) * ld1 + k);
        return ldg4(s2 + (((int64_t)(c.n % nmod2) * hs + i) * ws + j) * ld2 + (k - c1));
    }
};

// conv2d_transpose filter as the B operand: w[ky][kx][a][b] (a = output channel = tile row, b = k).
struct KmConvTWeights {
    static constexpr bool KM = true;
    const float* w; int ca, cb;    // cb = c1 + c2
    int cps;
    struct Ctx { int row; bool ok; };
    __device__ int nchunks_of(int) const { return 0; }
    __device__ void prep(int, int row, Ctx& c) const { c.row = row; c.ok = row < ca; }
    __device__ float4 load(const Ctx& c, int prob, int chunk, int k4) const {
        if (!c.ok) return zero4();
        const int py = prob >> 1, px = prob & 1, ntx = 2 + px;
        const int seg = chunk / cps, k = (chunk - seg * cps) * KC + k4;
        const int sy = seg / ntx, sx = seg - sy * ntx;
        const int ky = 1 - py + 2 * sy, kx = 1 - px + 2 * sx;
        return ldg4(w + ((int64_t)(ky * 5 + kx) * ca + c.row) * cb + k);
    }
};

// conv2d forward operand when cin == 3 (the frame itself, or the decoder's output gradient): for a
// fixed ky the 5 taps x 3 channels of a row are 15 CONTIGUOUS floats starting at pixel (.., 2j-1).
// A chunk holds two ky segments of 16 (15 + one zero); 3 chunks cover ky = 0..4.
struct KmC3Gather {
    static constexpr bool KM = true;
    const float* x;
    int hb, wb, hs, ws;
    int R;
    struct Ctx { int64_t img_base; int i2, j2; bool ok; };
    __device__ int nchunks_of(int) const { return 3; }
    __device__ void prep(int, int row, Ctx& c) const {
        c.ok = row < R;
        const int j = row % ws, t = row / ws, i = t % hs, n = t / hs;
        c.img_base = (int64_t)n * hb * wb; c.i2 = 2 * i - 1; c.j2 = 2 * j - 1;
    }
    __device__ float4 load(const Ctx& c, int, int chunk, int k4) const {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int ky = 2 * chunk + (k4 >> 4);
        const int y = c.i2 + ky;
        if (c.ok && ky < 5 && (unsigned)y < (unsigned)hb) {
            const float* rowp = x + (c.img_base + (int64_t)y * wb) * 3;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int el = (k4 & 15) + u;       // (kx, ch) = (el / 3, el % 3)
                const int xx = c.j2 + el / 3;
                if (el < 15 && (unsigned)xx < (unsigned)wb) v[u] = rowp[(int64_t)c.j2 * 3 + el];
            }
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
};

// Per-pixel channel vectors of the decoder's concat input [s1 | s2]: row m = pixel (img,i,j), k =
// channel; s2 is the ctx skip (image index img % nmod2).  Used by the d_h4 scatter product.
struct KmCat2 {
    static constexpr bool KM = true;
    const float* s1; int64_t ld1; int c1;
    const float* s2; int64_t ld2; int nmod2;
    int hsws;        // pixels per image
    int R;           // pixels
    int cps;         // (c1 + c2) / 32
    struct Ctx { int64_t p1, p2; bool ok; };
    __device__ int nchunks_of(int) const { return cps; }
    __device__ void prep(int, int row, Ctx& c) const {
        c.ok = row < R;
        const int n = row / hsws, rem = row - n * hsws;
        c.p1 = (int64_t)row * ld1;
        c.p2 = ((int64_t)(n % nmod2) * hsws + rem) * ld2;
    }
    __device__ float4 load(const Ctx& c, int, int chunk, int k4) const {
        if (!c.ok) return zero4();
        const int k = chunk * KC + k4;
        return k < c1 ? ldg4(s1 + c.p1 + k) : ldg4(s2 + c.p2 + (k - c1));
    }
};

// Plain k-major matrix V[k][r], optionally split along r into two buffers; ragged K allowed.
struct NmPlain {
    static constexpr bool KM = false;
    const float* p0; int64_t ld0;
    const float* p1; int64_t ld1;
    int rsplit;      // r < rsplit -> p0, else p1[r - rsplit]
    int R;           // valid r
    int K;           // valid k
    __device__ int nchunks_of(int) const { return (K + KC - 1) / KC; }
    __device__ float4 load(int, int chunk, int kk, int r4) const {
        const int k = chunk * KC + kk;
        if (k >= K || r4 >= R) return zero4();
        return r4 < rsplit ? ldg4(p0 + (int64_t)k * ld0 + r4) : ldg4(p1 + (int64_t)k * ld1 + (r4 - rsplit));
    }
};

// cin == 3 conv filter as the B operand of KmC3Gather: k = (ky, el) with el = kx*3 + ch < 15.
struct NmC3Weights {
    static constexpr bool KM = false;
    const float* w; int cb;
    __device__ float4 load(int, int chunk, int kk, int r4) const {
        const int ky = 2 * chunk + (kk >> 4), el = kk & 15;
        if (ky >= 5 || el == 15 || r4 >= cb) return zero4();
        return ldg4(w + (int64_t)(ky * 15 + el) * cb + r4);
    }
};

// Filter-gradient operand: k = output-grid pixel (img,i,j); rows = channels of the BIG tensor at the
// pixel shifted by tap prob = ky*5+kx:  dw[ky,kx,a,b] = sum big[img,2i+ky-1,2j+kx-1,a] * small[img,i,j,b].
struct NmWgradBig {
    static constexpr bool KM = false;
    const float* big; int64_t ldb; int ca;
    int hb, wb, hs, ws;
    int npix;        // imgs * hs * ws
    __device__ int nchunks_of(int) const { return (npix + KC - 1) / KC; }
    __device__ float4 load(int prob, int chunk, int kk, int r4) const {
        const int p = chunk * KC + kk;
        if (p >= npix || r4 >= ca) return zero4();
        const int ky = prob / 5, kx = prob - ky * 5;
        const int j = p % ws, t = p / ws, i = t % hs, n = t / hs;
        const int y = 2 * i + ky - 1, xx = 2 * j + kx - 1;
        if ((unsigned)y >= (unsigned)hb || (unsigned)xx >= (unsigned)wb) return zero4();
        return ldg4(big + (((int64_t)n * hb + y) * wb + xx) * ldb + r4);
    }
};

// Filter-gradient operand, small side: k = pixel, rows = channels of [s1 | s2] (s2 = ctx skip, image
// index img % nmod2).  Single-source tensors pass c1 = total channels.
struct NmWgradSmall {
    static constexpr bool KM = false;
    const float* s1; int64_t ld1; int c1;
    const float* s2; int64_t ld2; int nmod2;
    int cb;          // c1 + c2
    int hsws;        // pixels per image
    int npix;
    __device__ float4 load(int, int chunk, int kk, int r4) const {
        const int p = chunk * KC + kk;
        if (p >= npix || r4 >= cb) return zero4();
        if (r4 < c1) return ldg4(s1 + (int64_t)p * ld1 + r4);
        const int n = p / hsws, rem = p - n * hsws;
        return ldg4(s2 + ((int64_t)(n % nmod2) * hsws + rem) * ld2 + (r4 - c1));
    }
};

// Filter gradient when the big tensor has 3 channels: rows m = ky*16 + el (el = kx*3+ch < 15).
struct NmC3WgradBig {
    static constexpr bool KM = false;
    const float* big;
    int hb, wb, hs, ws;
    int npix;
    __device__ int nchunks_of(int) const { return (npix + KC - 1) / KC; }
    __device__ float4 load(int, int chunk, int kk, int r4) const {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int p = chunk * KC + kk;
        const int ky = r4 >> 4;
        if (p < npix && ky < 5) {
            const int j = p % ws, t = p / ws, i = t % hs, n = t / hs;
            const int y = 2 * i + ky - 1, j2 = 2 * j - 1;
            if ((unsigned)y < (unsigned)hb) {
                const float* rowp = big + (((int64_t)n * hb + y) * wb) * 3;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int el = (r4 & 15) + u;
                    const int xx = j2 + el / 3;
                    if (el < 15 && (unsigned)xx < (unsigned)wb) v[u] = rowp[(int64_t)j2 * 3 + el];
                }
            }
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
};

// ------------------------------------------------------------------------------------------------
// LDS tiles
// ------------------------------------------------------------------------------------------------
template <bool KMF, int TR>
struct Tile {
    static constexpr int FLOATS = KMF ? TR * LDK : KC * TR;
    static constexpr int NPASS = TR / 32;     // float4 per thread per chunk
    // global row (KM) handled by thread tid in pass p
    __device__ static int km_row(int tid, int p) { return (tid >> 3) + 32 * p; }
    __device__ static int km_k4(int tid) { return (tid & 7) * 4; }
    __device__ static int nm_kk(int tid, int p) { return tid / (TR / 4) + (NTHREADS / (TR / 4)) * p; }
    __device__ static int nm_r4(int tid) { return (tid % (TR / 4)) * 4; }
    __device__ static void store(float* s, int tid, int p, float4 v) {
        if (KMF) *reinterpret_cast<float4*>(&s[km_row(tid, p) * LDK + km_k4(tid)]) = v;
        else *reinterpret_cast<float4*>(&s[nm_kk(tid, p) * TR + nm_r4(tid)]) = v;
    }
    // the 4 values lane (row, half h) feeds to MFMA steps (q, 0..3)
    __device__ static void frag(const float* s, int row, int q, int h, float (&f)[4]) {
        if (KMF) {
            const float4 v = *reinterpret_cast<const float4*>(&s[row * LDK + 8 * q + 4 * h]);
            f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) f[t] = s[(8 * q + 4 * h + t) * TR + row];
        }
    }
};

template <class L, int TR, bool KMF = L::KM>
struct Fetch;
template <class L, int TR>
struct Fetch<L, TR, true> {
    typename L::Ctx c[TR / 32];
    __device__ void init(const L& l, int prob, int row0, int tid) {
#pragma unroll
        for (int p = 0; p < TR / 32; ++p) l.prep(prob, row0 + Tile<true, TR>::km_row(tid, p), c[p]);
    }
    __device__ void load(const L& l, int prob, int, int chunk, int tid, float4 (&r)[TR / 32]) const {
#pragma unroll
        for (int p = 0; p < TR / 32; ++p) r[p] = l.load(c[p], prob, chunk, Tile<true, TR>::km_k4(tid));
    }
};
template <class L, int TR>
struct Fetch<L, TR, false> {
    __device__ void init(const L&, int, int, int) {}
    __device__ void load(const L& l, int prob, int row0, int chunk, int tid, float4 (&r)[TR / 32]) const {
#pragma unroll
        for (int p = 0; p < TR / 32; ++p)
            r[p] = l.load(prob, chunk, Tile<false, TR>::nm_kk(tid, p), row0 + Tile<false, TR>::nm_r4(tid));
    }
};

// ------------------------------------------------------------------------------------------------
// The kernel.  grid = (m tiles, n tiles, nprob * nsplit).
// ------------------------------------------------------------------------------------------------
template <class LA, class LB, int MI, int NI>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(const LA la, const LB lb, const Epi ep, int M, int N,
                                                         int nprob, int nsplit) {
    constexpr int TM = 64 * MI, TN = 64 * NI;
    using TA = Tile<LA::KM, TM>;
    using TB = Tile<LB::KM, TN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + TA::FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int prob = blockIdx.z % nprob, split = blockIdx.z / nprob;
    const int nch = la.nchunks_of(prob);
    const int per = (nch + nsplit - 1) / nsplit;
    const int cb = split * per;
    const int ce = (cb + per < nch) ? cb + per : nch;

    Fetch<LA, TM> fa;
    Fetch<LB, TN> fb;
    fa.init(la, prob, m0, tid);
    fb.init(lb, prob, n0, tid);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    float4 ra[TA::NPASS], rb[TB::NPASS];
    if (cb < ce) {
        fa.load(la, prob, m0, cb, tid, ra);
        fb.load(lb, prob, n0, cb, tid, rb);
    }
    for (int c = cb; c < ce; ++c) {
#pragma unroll
        for (int p = 0; p < TA::NPASS; ++p) TA::store(sA, tid, p, ra[p]);
#pragma unroll
        for (int p = 0; p < TB::NPASS; ++p) TB::store(sB, tid, p, rb[p]);
        __syncthreads();
        if (c + 1 < ce) {   // next chunk's HBM/L2 reads fly under this chunk's 64 MFMAs
            fa.load(la, prob, m0, c + 1, tid, ra);
            fb.load(lb, prob, n0, c + 1, tid, rb);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a[MI][4], b[NI][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) TA::frag(sA, wm * 32 * MI + mi * 32 + l31, q, h, a[mi]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) TB::frag(sB, wn * 32 * NI + ni * 32 + l31, q, h, b[ni]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }

    // D layout (32x32 MFMA): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 * MI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= M) continue;
            if (ep.slab) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + wn * 32 * NI + ni * 32 + l31;
                    if (n < N) ep.slab[(((int64_t)split * nprob + prob) * M + m) * N + n] = acc[mi][ni][r];
                }
            } else {
                int64_t pix;
                if (!epi_row(ep, prob, m, pix)) continue;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + wn * 32 * NI + ni * 32 + l31;
                    if (n < N) epi_store(ep, prob, pix, n, acc[mi][ni][r]);
                }
            }
        }
    }
}

}  // namespace ctx
