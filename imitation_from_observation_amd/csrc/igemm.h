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
// Block = 256 threads = 4 waves (2 x 2), block tile (64*MI) x (64*NI), K staged KC = 32 at a time
// through a double-buffered LDS ring with ONE barrier per chunk.  Measured anatomy on MI355X
// (tools/mfma_ablate.hip): the MFMA stream alone runs at the 152 TF/s pipe peak (2.32 GHz); what
// costs is anything that sits BETWEEN a barrier and the first MFMA.  Therefore
//   * loaders are branch-free -- out-of-range lanes (SAME padding, ragged edges) read a zero page, so
//     the whole chunk body is one basic block the scheduler can interleave;
//   * per-row state (base pointer, 25-bit tap-validity mask) is hoisted out of the K loop; per chunk
//     only a wave-uniform offset is added;
//   * the global loads of chunk c+2 and the LDS stores of chunk c+1 are issued in the gaps between
//     the MFMA groups of chunk c (two register sets: 1.5 chunks between a load and its first use).
// What did NOT help (measured, kept out): XCD-contiguous / n-tile-fastest block orders (-8..-20 %),
// pinning the schedule with sched_barrier (-2 %).  What remains (72 % of the pipe peak at the measured
// clock) tracks L2-miss traffic per FLOP, not latency: see DESIGN.md section 6.
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

// (n, i, j) of a flat pixel index; shifts when the grid is a power of two (the 64x64 production
// size), integer division otherwise.  sh < 0 means "not a power of two".
struct PixDiv {
    int ws, hs, ws_sh, hs_sh;
    __device__ __forceinline__ void split(int p, int& n, int& i, int& j) const {
        if (ws_sh >= 0 && hs_sh >= 0) {
            j = p & (ws - 1);
            const int t = p >> ws_sh;
            i = t & (hs - 1);
            n = t >> hs_sh;
        } else {
            j = p % ws;
            const int t = p / ws;
            i = t % hs;
            n = t / hs;
        }
    }
};
inline PixDiv make_pixdiv(int hs, int ws) {
    auto sh = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
    return PixDiv{ws, hs, sh(ws), sh(hs)};
}

// chunk -> (tap segment, channel-slice index) for "slice outer, taps inner" K order; ntap in {4,6,9,25}
// is matched to a literal so the division is a multiply-shift on the scalar unit
__device__ __forceinline__ void tap_slice(int chunk, int ntap, int& seg, int& slice) {
    switch (ntap) {
        case 4: slice = chunk >> 2; break;
        case 6: slice = chunk / 6; break;
        case 9: slice = chunk / 9; break;
        default: slice = chunk / 25; break;
    }
    seg = chunk - slice * ntap;
}

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// branch-free guarded load: `zeros` is >= 16 bytes of device zeros
__device__ __forceinline__ float4 ldg4_or0(const float* p, bool ok, const float* zeros) { return ldg4(ok ? p : zeros); }

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
// Loaders.  KM loaders: prep(prob,row,ctx) once per tile row, then load(ctx,prob,chunk,k4) -> 4
//           consecutive k.  NM loaders: load(prob,chunk,kk,r4) -> rows r4..r4+3 at k = chunk*32+kk.
// All are branch-free and return zeros outside the virtual matrix (SAME padding, ragged M/N/K) by
// redirecting the load to `zeros`.
// ------------------------------------------------------------------------------------------------

// Plain row-major matrix V[row][k], optionally split along k into two buffers (the translate MLP's
// concat([src_z, ctx_z]), arm_shaping.py:1310).
struct KmPlain {
    static constexpr bool KM = true;
    const float* p0; int64_t ld0;
    const float* p1; int64_t ld1;
    int ksplit;      // k < ksplit -> p0, else p1[k - ksplit]; a multiple of KC
    int R;           // valid rows
    int nchunks;
    const float* zeros;
    struct Ctx { const float* r0; const float* r1; bool ok; };
    __device__ int nchunks_of(int) const { return nchunks; }
    __device__ void prep(int, int row, Ctx& c) const {
        c.ok = row < R;
        c.r0 = p0 + (int64_t)row * ld0;
        c.r1 = p1 + (int64_t)row * ld1 - ksplit;
    }
    __device__ float4 load(const Ctx& c, int, int chunk, int k4) const {
        const int k = chunk * KC + k4;
        return ldg4_or0((chunk * KC < ksplit ? c.r0 : c.r1) + k, c.ok, zeros);
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
    const float* zeros;
    struct Ctx { const float* base; unsigned mask; };   // base -> input pixel (2i-1, 2j-1); mask bit = tap valid
    __device__ int nchunks_of(int) const { return 25 * cps; }
    __device__ void prep(int, int row, Ctx& c) const {
        const int j = row % ws, t = row / ws, i = t % hs, n = t / hs;
        const int i2 = 2 * i - 1, j2 = 2 * j - 1;
        c.base = x + (((int64_t)n * hb + i2) * wb + j2) * ldx;
        unsigned m = 0;
#pragma unroll
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx)
                if ((unsigned)(i2 + ky) < (unsigned)hb && (unsigned)(j2 + kx) < (unsigned)wb) m |= 1u << (ky * 5 + kx);
        c.mask = row < R ? m : 0u;
    }
    __device__ float4 load(const Ctx& c, int, int chunk, int k4) const {
        // K order: 32-channel slice outer, the 25 taps inner -- a block re-reads one slice of its input
        // halo (L2-resident) 25 times before moving to the next slice
        int seg, slice;
        tap_slice(chunk, 25, seg, slice);                                 // wave-uniform
        const int kc = slice * KC;
        const int ky = seg / 5, kx = seg - ky * 5;
        const int64_t off = (int64_t)(ky * wb + kx) * ldx + kc;           // wave-uniform
        return ldg4_or0(c.base + off + k4, (c.mask >> seg) & 1u, zeros);
    }
};

// conv2d_transpose operand for output parity class prob = (py,px): row m = (img, i', j') with output
// pixel (2i'+py, 2j'+px); valid taps ky = 1-py+2sy (sy < 2+py), input pixel i = i'+py-sy (same in x).
// Input channels come from two tensors [decoder | ctx skip]; the skip is shared by both decoder
// passes, so its image index is img % nmod2 (arm_shaping.py:1323 and :1336 use the same tgtctx_h*).
struct KmConvTGather {
    static constexpr bool KM = true;
    const float* s1; int64_t ld1; int c1;   // c1 a multiple of KC
    const float* s2; int64_t ld2; int nmod2;
    int hs, ws;
    int cps;                       // (c1 + c2) / 32
    int R;
    const float* zeros;
    struct Ctx { const float* b1; const float* b2; unsigned mask; };   // b* -> input pixel (i'+py, j'+px)
    __device__ int nchunks_of(int prob) const { return (2 + (prob >> 1)) * (2 + (prob & 1)) * cps; }
    __device__ void prep(int prob, int row, Ctx& c) const {
        const int py = prob >> 1, px = prob & 1;
        const int j = row % ws, t = row / ws, i = t % hs, n = t / hs;
        const int64_t pix = (int64_t)(i + py) * ws + (j + px);
        c.b1 = s1 + ((int64_t)n * hs * ws + pix) * ld1;
        c.b2 = s2 + ((int64_t)(n % nmod2) * hs * ws + pix) * ld2 - c1;
        unsigned m = 0;
#pragma unroll
        for (int sy = 0; sy < 3; ++sy)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx)
                if ((unsigned)(i + py - sy) < (unsigned)hs && (unsigned)(j + px - sx) < (unsigned)ws) m |= 1u << (sy * 3 + sx);
        c.mask = row < R ? m : 0u;
    }
    __device__ float4 load(const Ctx& c, int prob, int chunk, int k4) const {
        const int ntx = 2 + (prob & 1), ntap = (2 + (prob >> 1)) * ntx;
        int seg, slice;
        tap_slice(chunk, ntap, seg, slice);                               // wave-uniform; slice outer, taps inner
        const int kc = slice * KC;
        const int sy = ntx == 2 ? seg >> 1 : seg / 3, sx = seg - sy * ntx;
        const int poff = sy * ws + sx;                                    // pixels back from (i'+py, j'+px)
        const float* p = kc < c1 ? c.b1 - (int64_t)poff * ld1 : c.b2 - (int64_t)poff * ld2;
        return ldg4_or0(p + kc + k4, (c.mask >> (sy * 3 + sx)) & 1u, zeros);
    }
};

// conv2d_transpose filter as the B operand: w[ky][kx][a][b] (a = output channel = tile row, b = k).
struct KmConvTWeights {
    static constexpr bool KM = true;
    const float* w; int ca, cb;    // cb = c1 + c2
    int cps;
    const float* zeros;
    struct Ctx { const float* rowp; bool ok; };
    __device__ int nchunks_of(int) const { return 0; }
    __device__ void prep(int, int row, Ctx& c) const { c.ok = row < ca; c.rowp = w + (int64_t)row * cb; }
    __device__ float4 load(const Ctx& c, int prob, int chunk, int k4) const {
        const int py = prob >> 1, px = prob & 1, ntx = 2 + px, ntap = (2 + py) * ntx;
        int seg, slice;
        tap_slice(chunk, ntap, seg, slice);                               // same K order as KmConvTGather
        const int kc = slice * KC;
        const int sy = ntx == 2 ? seg >> 1 : seg / 3, sx = seg - sy * ntx;
        const int ky = 1 - py + 2 * sy, kx = 1 - px + 2 * sx;
        return ldg4_or0(c.rowp + (int64_t)(ky * 5 + kx) * ca * cb + kc + k4, c.ok, zeros);
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
    const float* zeros;
    struct Ctx { const float* p1; const float* p2; bool ok; };
    __device__ int nchunks_of(int) const { return cps; }
    __device__ void prep(int, int row, Ctx& c) const {
        c.ok = row < R;
        const int n = row / hsws, rem = row - n * hsws;
        c.p1 = s1 + (int64_t)row * ld1;
        c.p2 = s2 + ((int64_t)(n % nmod2) * hsws + rem) * ld2 - c1;
    }
    __device__ float4 load(const Ctx& c, int, int chunk, int k4) const {
        const int k = chunk * KC + k4;
        return ldg4_or0((chunk * KC < c1 ? c.p1 : c.p2) + k, c.ok, zeros);
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
    const float* zeros;
    struct Ctx { const float* base; int i2, j2; bool ok; };   // base -> element (2i-1, 2j-1, 0)
    __device__ int nchunks_of(int) const { return 3; }
    __device__ void prep(int, int row, Ctx& c) const {
        c.ok = row < R;
        const int j = row % ws, t = row / ws, i = t % hs, n = t / hs;
        c.i2 = 2 * i - 1; c.j2 = 2 * j - 1;
        c.base = x + (((int64_t)n * hb + c.i2) * wb + c.j2) * 3;
    }
    __device__ float4 load(const Ctx& c, int, int chunk, int k4) const {
        const int ky = 2 * chunk + (k4 >> 4);
        const bool rowok = c.ok && ky < 5 && (unsigned)(c.i2 + ky) < (unsigned)hb;
        const float* rowp = c.base + (int64_t)ky * wb * 3;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int el = (k4 & 15) + u;        // (kx, ch) = (el / 3, el % 3)
            const bool ok = rowok && el < 15 && (unsigned)(c.j2 + el / 3) < (unsigned)wb;
            v[u] = *(ok ? rowp + el : zeros);
        }
        return make_float4(v[0], v[1], v[2], v[3]);
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
    const float* zeros;
    int seglen = 0;  // > 0: K = 25 tap segments of seglen rows, chunk order (slice outer, tap inner) to match
                     //      KmConvGather -- the conv filter [25][cin][cout]
    __device__ int nchunks_of(int) const { return (K + KC - 1) / KC; }
    __device__ float4 load(int, int chunk, int kk, int r4) const {
        int k = chunk * KC + kk;
        if (seglen) {
            int seg, slice;
            tap_slice(chunk, 25, seg, slice);
            k = seg * seglen + slice * KC + kk;
        }
        const float* p = r4 < rsplit ? p0 + (int64_t)k * ld0 + r4 : p1 + (int64_t)k * ld1 + (r4 - rsplit);
        return ldg4_or0(p, k < K && r4 < R, zeros);
    }
};

// cin == 3 conv filter as the B operand of KmC3Gather: k = (ky, el) with el = kx*3 + ch < 15.
struct NmC3Weights {
    static constexpr bool KM = false;
    const float* w; int cb;
    const float* zeros;
    __device__ float4 load(int, int chunk, int kk, int r4) const {
        const int ky = 2 * chunk + (kk >> 4), el = kk & 15;
        return ldg4_or0(w + (int64_t)(ky * 15 + el) * cb + r4, ky < 5 && el < 15 && r4 < cb, zeros);
    }
};

// Filter-gradient operand: k = output-grid pixel (img,i,j); rows = channels of the BIG tensor at the
// pixel shifted by tap prob = ky*5+kx:  dw[ky,kx,a,b] = sum big[img,2i+ky-1,2j+kx-1,a] * small[img,i,j,b].
struct NmWgradBig {
    static constexpr bool KM = false;
    const float* big; int64_t ldb; int ca;
    int hb, wb;
    PixDiv pd;
    int npix;        // imgs * hs * ws
    const float* zeros;
    __device__ int nchunks_of(int) const { return (npix + KC - 1) / KC; }
    __device__ float4 load(int prob, int chunk, int kk, int r4) const {
        const int p = chunk * KC + kk;
        const int ky = prob / 5, kx = prob - ky * 5;
        int n, i, j;
        pd.split(p, n, i, j);
        const int y = 2 * i + ky - 1, xx = 2 * j + kx - 1;
        const bool ok = p < npix && r4 < ca && (unsigned)y < (unsigned)hb && (unsigned)xx < (unsigned)wb;
        return ldg4_or0(big + (((int64_t)n * hb + y) * wb + xx) * ldb + r4, ok, zeros);
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
    int hsws_sh;     // log2(hsws) or -1
    int npix;
    const float* zeros;
    __device__ float4 load(int, int chunk, int kk, int r4) const {
        const int p = chunk * KC + kk;
        const int n = hsws_sh >= 0 ? p >> hsws_sh : p / hsws, rem = p - n * hsws;
        const float* q = r4 < c1 ? s1 + (int64_t)p * ld1 + r4 : s2 + ((int64_t)(n % nmod2) * hsws + rem) * ld2 + (r4 - c1);
        return ldg4_or0(q, p < npix && r4 < cb, zeros);
    }
};

// Filter gradient when the big tensor has 3 channels: rows m = ky*16 + el (el = kx*3+ch < 15).
struct NmC3WgradBig {
    static constexpr bool KM = false;
    const float* big;
    int hb, wb;
    PixDiv pd;
    int npix;
    const float* zeros;
    __device__ int nchunks_of(int) const { return (npix + KC - 1) / KC; }
    __device__ float4 load(int, int chunk, int kk, int r4) const {
        const int p = chunk * KC + kk;
        const int ky = r4 >> 4;
        int n, i, j;
        pd.split(p, n, i, j);
        const int y = 2 * i + ky - 1, j2 = 2 * j - 1;
        const bool rowok = p < npix && ky < 5 && (unsigned)y < (unsigned)hb;
        const float* rowp = big + ((((int64_t)n * hb + y) * wb) + j2) * 3;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int el = (r4 & 15) + u;
            const bool ok = rowok && el < 15 && (unsigned)(j2 + el / 3) < (unsigned)wb;
            v[u] = *(ok ? rowp + el : zeros);
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

// per-thread view of a loader: which float4 of a chunk this thread fetches in pass p
template <class L, int TR, bool KMF = L::KM>
struct Fetch;
template <class L, int TR>
struct Fetch<L, TR, true> {
    typename L::Ctx c[TR / 32];
    __device__ void init(const L& l, int prob, int row0, int tid) {
#pragma unroll
        for (int p = 0; p < TR / 32; ++p) l.prep(prob, row0 + Tile<true, TR>::km_row(tid, p), c[p]);
    }
    __device__ float4 load1(const L& l, int prob, int, int chunk, int tid, int p) const {
        return l.load(c[p], prob, chunk, Tile<true, TR>::km_k4(tid));
    }
};
template <class L, int TR>
struct Fetch<L, TR, false> {
    __device__ void init(const L&, int, int, int) {}
    __device__ float4 load1(const L& l, int prob, int row0, int chunk, int tid, int p) const {
        return l.load(prob, chunk, Tile<false, TR>::nm_kk(tid, p), row0 + Tile<false, TR>::nm_r4(tid));
    }
};

// ------------------------------------------------------------------------------------------------
// The kernel.  1-D grid of gm * gn * nprob * nsplit blocks.
// ------------------------------------------------------------------------------------------------

template <class LA, class LB, int MI, int NI>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(const LA la, const LB lb, const Epi ep, int M, int N,
                                                         int nprob, int nsplit, int gm, int gn) {
    constexpr int TM = 64 * MI, TN = 64 * NI;
    using TA = Tile<LA::KM, TM>;
    using TB = Tile<LB::KM, TN>;
    constexpr int NA = TA::NPASS, NB = TB::NPASS;       // float4 per thread per chunk: 2..4 each
    constexpr int STAGE = TA::FLOATS + TB::FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 stages of [A tile | B tile]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, l31 = lane & 31, h = lane >> 5;

    // 1-D grid, m-tile fastest, then n-tile, then problem, then K-split.  Problems go last-first: the
    // (1,1) parity class of a transposed conv has 9 taps against 4 for (0,0), and the longest blocks
    // must not form the tail.  (XCD-contiguous and n-tile-fastest orders were measured: -8..-20 %.)
    int rest = blockIdx.x;
    const int bx = rest % gm; rest /= gm;
    const int by = rest % gn; rest /= gn;
    const int prob = nprob - 1 - rest % nprob;
    const int split = rest / nprob;
    const int m0 = bx * TM, n0 = by * TN;

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

    if (cb < ce) {
        const int last = ce - 1;
        // Two register sets: while chunk c is multiplied out of LDS stage c&1, set X (chunk c+1, loaded
        // during chunk c-1) is stored to the other stage in the SECOND half of the MFMA stream and set Y
        // is refilled with chunk c+2 in the FIRST half -- 1.5 chunks (>= 6000 cycles) between a load's
        // issue and its first use, which covers an L2 miss to Infinity Cache / HBM.
        float4 xa[NA], xb[NB], ya[NA], yb[NB];
        auto clampc = [&](int c) { return c < last ? c : last; };   // redundant tail reloads are harmless
        // prologue: chunk cb -> stage 0 ; chunk cb+1 -> set X
#pragma unroll
        for (int p = 0; p < NA; ++p) xa[p] = fa.load1(la, prob, m0, cb, tid, p);
#pragma unroll
        for (int p = 0; p < NB; ++p) xb[p] = fb.load1(lb, prob, n0, cb, tid, p);
#pragma unroll
        for (int p = 0; p < NA; ++p) TA::store(smem, tid, p, xa[p]);
#pragma unroll
        for (int p = 0; p < NB; ++p) TB::store(smem + TA::FLOATS, tid, p, xb[p]);
#pragma unroll
        for (int p = 0; p < NA; ++p) xa[p] = fa.load1(la, prob, m0, clampc(cb + 1), tid, p);
#pragma unroll
        for (int p = 0; p < NB; ++p) xb[p] = fb.load1(lb, prob, n0, clampc(cb + 1), tid, p);
        __syncthreads();

        // one chunk: multiply stage `st`; slots 0..7 load chunk c+2 into (la_, lb_); slots 8..15 store
        // (sa_, sb_) = chunk c+1 into the other stage
        auto chunk = [&](int c, int st, float4 (&la_)[NA], float4 (&lb_)[NB], float4 (&sa_)[NA], float4 (&sb_)[NB]) {
            const float* sA = smem + st * STAGE;
            const float* sB = sA + TA::FLOATS;
            float* nA = smem + (st ^ 1) * STAGE;
            float* nB = nA + TA::FLOATS;
            const int c2 = clampc(c + 2);
            float a[2][MI][4], b[2][NI][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) TA::frag(sA, wm * 32 * MI + mi * 32 + l31, 0, h, a[0][mi]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) TB::frag(sB, wn * 32 * NI + ni * 32 + l31, 0, h, b[0][ni]);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int q = g >> 2, t = g & 3;
                if (t == 0 && q < 3) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) TA::frag(sA, wm * 32 * MI + mi * 32 + l31, q + 1, h, a[(q + 1) & 1][mi]);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) TB::frag(sB, wn * 32 * NI + ni * 32 + l31, q + 1, h, b[(q + 1) & 1][ni]);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mi][t], b[q & 1][ni][t], acc[mi][ni], 0, 0, 0);
                if (g < NA) la_[g] = fa.load1(la, prob, m0, c2, tid, g);
                else if (g < NA + NB) lb_[g - NA] = fb.load1(lb, prob, n0, c2, tid, g - NA);
                else if (g >= 8 && g - 8 < NA) TA::store(nA, tid, g - 8, sa_[g - 8]);
                else if (g >= 8 && g - 8 < NA + NB) TB::store(nB, tid, g - 8 - NA, sb_[g - 8 - NA]);
            }
            __syncthreads();
        };
        for (int c = cb; c < ce; c += 2) {
            chunk(c, 0, ya, yb, xa, xb);
            if (c + 1 < ce) chunk(c + 1, 1, xa, xb, ya, yb);
        }
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
