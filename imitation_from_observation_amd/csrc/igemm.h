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
// Block = WM x WN waves, block tile (32*MI*WM) x (32*NI*WN), K staged KC = 32 at a time through a double-buffered
// LDS ring with ONE barrier per chunk.  The shipped 128x128 tile runs EIGHT waves (8 x 64x32 or 8 x 32x64, chosen per
// loader pair in gemm_launch.h): at two blocks per CU that is four waves per SIMD, which hides the load latency
// better than four 64x64 waves did (-6..-20 % per kernel).  Measured anatomy on MI355X (an ablation micro-benchmark of round 1, notebook section 5): the
// MFMA stream alone runs at the 152 TF/s pipe peak (2.32 GHz); what costs is anything that sits BETWEEN a barrier
// and the first MFMA.  Therefore
//   * loaders are branch-free -- out-of-range lanes (SAME padding, ragged edges) get an out-of-range buffer offset
//     and the hardware returns zeros, so the whole chunk body is one basic block the scheduler can interleave;
//   * per-row state (byte offset, tap-validity mask) is hoisted out of the K loop; per chunk only a wave-uniform
//     descriptor base changes;
//   * the global loads of chunk c+2 and the LDS stores of chunk c+1 are issued in the gaps between the MFMA groups
//     of chunk c (two register sets: 1.5 chunks between a load and its first use).
// The "Q" / "R" loaders further down re-order rows and K so that the products with SAME-padding zeros are never
// formed at all.
// What did NOT help (measured, kept out): XCD-contiguous / n-tile-fastest block orders (-8..-20 %), pinning the
// schedule with sched_barrier (-2 %), 256x256 (f32) and 16-wave 256x128 tiles (+3..+14 % time).
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

// chunk -> (tap segment, channel-slice index) for "slice outer, taps inner" K order; the usual ntap values
// is matched to a literal so the division is a multiply-shift on the scalar unit
__device__ __forceinline__ void tap_slice(int chunk, int ntap, int& seg, int& slice) {
    switch (ntap) {
        case 1: slice = chunk; break;
        case 2: slice = chunk >> 1; break;
        case 3: slice = chunk / 3; break;
        case 4: slice = chunk >> 2; break;
        case 6: slice = chunk / 6; break;
        case 9: slice = chunk / 9; break;
        case 25: slice = chunk / 25; break;
        default: slice = chunk / ntap; break;
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
    int64_t add1_mod = 0;           // != 0: add1 has this many pixels and serves two row blocks (pix, pix - add1_mod)
    const float* add2 = nullptr;
    int64_t lda2 = 0;
    const float* mask = nullptr;    // saved lrelu OUTPUT; v *= (mask >= 0 ? 1 : 0.2) for n < nsplit
    int64_t ldm = 0;
    int lrelu = 0;                  // 1: v = max(v, 0.2 v) after bias/adds; 2: plain ReLU (the Inception front end)
    int rowmode = 0;                // 0: pix = m; 1: transposed-conv parity class; 2: rows m = tap*4+ch of a C=3 filter gradient;
                                    // 4 / 5: position-major conv / transposed conv (problem = position, row = image)
    int hs = 0, ws = 0;             // rowmode 1: small-grid size
    int64_t prob_stride = 0;        // out1 += prob * prob_stride (filter gradient: one tap per problem)
    float* slab = nullptr;          // split-K: raw partials to slab[((split*nprob+prob)*M + m)*N + n]
    int xcd_swizzle = 0;            // 1: consecutive work items go to the SAME XCD (its L2): block b does item (b % 8) * ceil(T/8) + b / 8
    int swz_group = 0;              // != 0 (multiple of 8): the swizzle is applied inside consecutive groups of this many items, so
                                    // all XCDs work on the same group (a parity class of a transposed conv) at the same time
    const uint16_t* perm = nullptr; // != null: problem slot pr runs problem perm[pr] (device memory, nprob entries): the launcher's
                                    // load-balanced order for problems of unequal length (gemm_launch.h: balanced_order)
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
    const int tap = m >> 2, ch = m & 3;
    if (tap >= 25 || ch == 3) return false;
    pix = tap * 3 + ch;
    return true;
}

struct RowMap { bool linear; int64_t base, stride; };
__device__ __forceinline__ RowMap epi_rowmap(const Epi& e, int prob) {
    if (e.rowmode == 0) return RowMap{true, 0, 1};
    if (e.rowmode == 4) return RowMap{true, prob, (int64_t)e.hs * e.ws};
    if (e.rowmode == 5) {
        const int np = e.hs * e.ws, c = prob / np, pos = prob - c * np, i = pos / e.ws, j = pos - i * e.ws;
        return RowMap{true, (int64_t)(2 * i + (c >> 1)) * (2 * e.ws) + 2 * j + (c & 1), (int64_t)4 * np};
    }
    return RowMap{false, 0, 0};
}

__device__ __forceinline__ void epi_store(const Epi& e, int prob, int64_t pix, int n, float v) {
    if (e.bias) v += e.bias[n];
    if (e.add1) v += e.add1[((e.add1_mod && pix >= e.add1_mod) ? pix - e.add1_mod : pix) * e.lda1 + n];
    if (e.add2) v += e.add2[pix * e.lda2 + n];
    if (e.lrelu) v = fmaxf(v, (e.lrelu == 2 ? 0.f : LEAK) * v);
    if (n < e.nsplit) {
        if (e.mask) v *= (e.mask[pix * e.ldm + n] >= 0.f) ? 1.f : LEAK;
        e.out1[(int64_t)prob * e.prob_stride + pix * e.ld1 + n] = v;
    } else {
        e.out2[pix * e.ld2 + (n - e.nsplit)] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Loaders.  Measured (rocprofv3 PMC): the first version spent 3-15 ALU instructions per MFMA on
// per-load address / validity math, and an in-order wave cannot hide that behind a 64-cycle MFMA.
// Hence the split used here:
//   Pos  pos(prob, chunk)            wave-uniform, computed ONCE per chunk on the scalar unit: which tap /
//                                    channel slice / source tensor; folded into the BASE of a buffer
//                                    descriptor (so no per-lane add is needed for it)
//   void prep(prob, a, b, Ctx&)      per-lane, loop-invariant: byte offset of this lane's float4 inside
//                                    the tensor (a,b = row,k4 for KM loaders; kk,r4 for NM loaders) and
//                                    its tap-validity bit mask
//   float4 load(Ctx, Pos)            one raw_buffer_load_b128; lanes that fall into SAME padding or past
//                                    a ragged edge get voffset = OOB >= num_records and the hardware
//                                    returns zeros -- no pointer select, no branch, no zero page.
// Descriptors are built with num_records = 2 GiB: every tensor here is far smaller, so the range check
// only ever fires for the OOB marker.
// ------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t OOB = 0x80000000u;

__device__ __forceinline__ rsrc_t make_rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)0x80000000, 0x00020000);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bload4(rsrc_t r, uint32_t voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Plain row-major matrix V[row][k], optionally split along k into two buffers (the translate MLP's
// concat([src_z, ctx_z]), arm_shaping.py:1310).
struct KmPlain {
    static constexpr bool KM = true;
    const float* p0; int64_t ld0;
    const float* p1; int64_t ld1;
    int ksplit;      // k < ksplit -> p0, else p1[k - ksplit]; a multiple of KC
    int R;           // valid rows
    int nchunks;
    const float* zeros;   // unused (kept so host initialisers are uniform)
    struct Pos { rsrc_t rs; bool second; };
    struct Ctx { uint32_t v0, v1; };
    __device__ int nchunks_of(int) const { return nchunks; }
    __device__ Pos pos(int, int chunk) const {
        const int k0 = chunk * KC;
        const bool second = k0 >= ksplit;
        return Pos{make_rsrc(second ? p1 + (k0 - ksplit) : p0 + k0), second};
    }
    __device__ void prep(int, int row, int k4, Ctx& c) const {
        const bool ok = row < R;
        c.v0 = ok ? (uint32_t)(row * ld0 + k4) * 4u : OOB;
        c.v1 = ok ? (uint32_t)(row * ld1 + k4) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, q.second ? c.v1 : c.v0); }
};

// conv2d forward operand: row m = (img, i, j) of the OUTPUT grid; segment = tap (ky,kx); the run is the cin
// channels of input pixel (s*i + ky - pad, s*j + kx - pad): TF SAME for k = 5 is pad 1 (s = 2, even input;
// 2 after) or pad 2 (s = 1).  flip = 1 reads pixel (i + pad - ky, j + pad - kx) instead: the stride-1
// conv2d_transpose written as a correlation (ContextAEReal's d_h2 / d_h4 and the input gradient of its
// stride-1 convs).  Channels may come from two tensors [x | x2] (x2 = ctx skip, image index img % nmod2).
// K order: 32-channel slice outer, the 25 taps inner.
struct KmConvGather {
    static constexpr bool KM = true;
    const float* x; int64_t ldx;   // NHWC input, channel stride ldx
    int hb, wb, hs, ws;            // input (big) and output (small) grids
    int cps;                       // chunks per tap = (c1 + c2) / 32
    int R;                         // imgs * hs * ws
    const float* zeros;
    int tap_outer = 0;             // K order: 0 = channel slice outer / taps inner, 1 = taps outer
    int s = 2, pad = 1, flip = 0;
    const float* x2 = nullptr; int64_t ldx2 = 0; int c1 = 1 << 30; int nmod2 = 1;
    int K = 5;                     // kernel height (5: ContextSkipNew / ContextAEReal, 3: ContextAEInception2)
    int KW = 0, padx = -1;         // kernel width / left pad when they differ from K / pad (1x7, 7x1, 1x3, 3x1 of Inception-v3); K*KW <= 25
    __host__ __device__ int kw() const { return KW ? KW : K; }
    __host__ __device__ int px() const { return padx >= 0 ? padx : pad; }
    __host__ __device__ int ntaps() const { return K * kw(); }
    struct Pos { rsrc_t rs; int seg; bool second; };
    struct Ctx { uint32_t v, v2; unsigned mask; };   // v -> input pixel (s*i, s*j); mask bit = tap valid
    __device__ int nchunks_of(int) const { return ntaps() * cps; }
    __device__ Pos pos(int, int chunk) const {
        int seg, slice;
        if (tap_outer) { seg = chunk / cps; slice = chunk - seg * cps; }
        else tap_slice(chunk, ntaps(), seg, slice);
        const int kwd = kw(), padl = px();
        const int ky = kwd == 5 ? seg / 5 : seg / kwd, kx = seg - ky * kwd;
        const int64_t off = (int64_t)(flip ? pad - ky : ky - pad) * wb + (flip ? padl - kx : kx - padl);
        const int kc = slice * KC;
        const bool second = kc >= c1;
        return Pos{make_rsrc(second ? x2 + off * ldx2 + (kc - c1) : x + off * ldx + kc), seg, second};
    }
    __device__ void prep(int, int row, int k4, Ctx& c) const {
        const int j = row % ws, t = row / ws, i = t % hs, n = t / hs;
        const int64_t pix = (int64_t)(s * i) * wb + s * j;
        c.v = (uint32_t)(((int64_t)n * hb * wb + pix) * ldx + k4) * 4u;
        c.v2 = (uint32_t)(((int64_t)(n % nmod2) * hb * wb + pix) * ldx2 + k4) * 4u;
        unsigned m = 0;
        const int kwd = kw(), padl = px();
        for (int ky = 0; ky < K; ++ky)
            for (int kx = 0; kx < kwd; ++kx) {
                const int y = s * i + (flip ? pad - ky : ky - pad), xx = s * j + (flip ? padl - kx : kx - padl);
                if ((unsigned)y < (unsigned)hb && (unsigned)xx < (unsigned)wb) m |= 1u << (ky * kwd + kx);
            }
        c.mask = row < R ? m : 0u;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const {
        return bload4(q.rs, (c.mask >> q.seg) & 1u ? (q.second ? c.v2 : c.v) : OOB);
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
    // General form (kernel K, SAME pad_before pb of the forward conv over the OUTPUT grid; K = 5: pb = 1, K = 3: pb = 0):
    // class py holds taps ky = par + 2 sy with par = (py + pb) & 1, sy < (K - par + 1) / 2, and input row
    // i = i' + off - sy with off = (py + pb - par) / 2.
    int K = 5, pb = 1;
    struct Pos { rsrc_t rs; int bit; bool second; };
    struct Ctx { uint32_t v1, v2; unsigned mask; };   // v* -> input pixel (i', j')
    __device__ __forceinline__ void cls(int p, int& nt, int& off) const {
        const int par = (p + pb) & 1;
        nt = (K - par + 1) >> 1;
        off = (p + pb - par) >> 1;
    }
    __device__ int nchunks_of(int prob) const {
        int nty, ntx, oy, ox;
        cls(prob >> 1, nty, oy); cls(prob & 1, ntx, ox);
        return nty * ntx * cps;
    }
    __device__ Pos pos(int prob, int chunk) const {
        int nty, ntx, oy, ox;
        cls(prob >> 1, nty, oy); cls(prob & 1, ntx, ox);
        int seg, slice;
        tap_slice(chunk, nty * ntx, seg, slice);
        const int kc = slice * KC;
        const int sy = ntx == 1 ? seg : ntx == 2 ? seg >> 1 : seg / 3, sx = seg - sy * ntx;
        const int64_t d = (int64_t)(oy - sy) * ws + (ox - sx);          // pixel shift of this tap
        const bool second = kc >= c1;
        return Pos{make_rsrc(second ? s2 + d * ld2 + (kc - c1) : s1 + d * ld1 + kc), sy * 3 + sx, second};
    }
    __device__ void prep(int prob, int row, int k4, Ctx& c) const {
        int nty, ntx, oy, ox;
        cls(prob >> 1, nty, oy); cls(prob & 1, ntx, ox);
        const int j = row % ws, t = row / ws, i = t % hs, n = t / hs;
        const int64_t pix = (int64_t)i * ws + j;
        c.v1 = (uint32_t)(((int64_t)n * hs * ws + pix) * ld1 + k4) * 4u;
        c.v2 = (uint32_t)(((int64_t)(n % nmod2) * hs * ws + pix) * ld2 + k4) * 4u;
        unsigned m = 0;
#pragma unroll
        for (int sy = 0; sy < 3; ++sy)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx)
                if ((unsigned)(i + oy - sy) < (unsigned)hs && (unsigned)(j + ox - sx) < (unsigned)ws) m |= 1u << (sy * 3 + sx);
        c.mask = row < R ? m : 0u;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const {
        return bload4(q.rs, (c.mask >> q.bit) & 1u ? (q.second ? c.v2 : c.v1) : OOB);
    }
};

// conv2d_transpose filter as the B operand: w[ky][kx][a][b] (a = output channel = tile row, b = k).
struct KmConvTWeights {
    static constexpr bool KM = true;
    const float* w; int ca, cb;    // cb = c1 + c2
    int cps;
    const float* zeros;
    int flip25 = 0;                // 1: B operand of KmConvGather{flip = 1}: all K*K taps, tap index = seg itself
    int K = 5, pb = 1;             // as in KmConvTGather
    struct Pos { rsrc_t rs; };
    struct Ctx { uint32_t v; };
    __device__ int nchunks_of(int) const { return 0; }
    __device__ Pos pos(int prob, int chunk) const {
        if (flip25) {              // out[y,x,c] = sum in[y+pad-ky, x+pad-kx, k] * w[ky,kx,c,k]: same (ky,kx) on both sides
            int seg, slice;
            tap_slice(chunk, K * K, seg, slice);
            return Pos{make_rsrc(w + (int64_t)seg * ca * cb + slice * KC)};
        }
        const int pary = ((prob >> 1) + pb) & 1, parx = ((prob & 1) + pb) & 1;
        const int nty = (K - pary + 1) >> 1, ntx = (K - parx + 1) >> 1;
        int seg, slice;
        tap_slice(chunk, nty * ntx, seg, slice);
        const int sy = ntx == 1 ? seg : ntx == 2 ? seg >> 1 : seg / 3, sx = seg - sy * ntx;
        const int ky = pary + 2 * sy, kx = parx + 2 * sx;
        return Pos{make_rsrc(w + (int64_t)(ky * K + kx) * ca * cb + slice * KC)};
    }
    __device__ void prep(int, int row, int k4, Ctx& c) const { c.v = row < ca ? (uint32_t)(row * cb + k4) * 4u : OOB; }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, c.v); }
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
    struct Pos { rsrc_t rs; bool second; };
    struct Ctx { uint32_t v1, v2; };
    __device__ int nchunks_of(int) const { return cps; }
    __device__ Pos pos(int, int chunk) const {
        const int k0 = chunk * KC;
        const bool second = k0 >= c1;
        return Pos{make_rsrc(second ? s2 + (k0 - c1) : s1 + k0), second};
    }
    __device__ void prep(int, int row, int k4, Ctx& c) const {
        const bool ok = row < R;
        const int n = row / hsws, rem = row - n * hsws;
        c.v1 = ok ? (uint32_t)((int64_t)row * ld1 + k4) * 4u : OOB;
        c.v2 = ok ? (uint32_t)(((int64_t)(n % nmod2) * hsws + rem) * ld2 + k4) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, q.second ? c.v2 : c.v1); }
};

// cin == 3 operands (the frame itself, or d loss / d out of the 3-channel decoder output).  The tensor is read from
// a 4-CHANNEL COPY x4[pixel][4] (channel 3 = 0; pack3to4_kernel) so that one tap of one pixel is ONE aligned
// 16-byte load.  K order: k = tap * 4 + ch with the 25 taps padded to 32: 4 chunks of 8 taps, tap = 8 * chunk + k4 / 4.
struct KmC3Gather {
    static constexpr bool KM = true;
    const float* x4;
    int hb, wb, hs, ws;
    int R;
    const float* zeros;
    int s = 2, pad = 1;            // (1, 2) for the stride-1 layers of ContextAEReal
    int K = 5;                     // 3: the Inception stem's first conv (3x3 VALID: pad 0)
    struct Pos { rsrc_t rs; int tap0; };
    struct Ctx { int pb, y0, x0, t8; };   // pb = pixel index of (s*i - pad, s*j - pad), may be negative; t8 < 0: row invalid
    __device__ int nchunks_of(int) const { return (K * K + 7) >> 3; }
    __device__ Pos pos(int, int chunk) const { return Pos{make_rsrc(x4), 8 * chunk}; }
    __device__ void prep(int, int row, int k4, Ctx& c) const {
        const int j = row % ws, t = row / ws, i = t % hs, n = t / hs;
        c.y0 = s * i - pad; c.x0 = s * j - pad;
        c.pb = (n * hb + c.y0) * wb + c.x0;
        c.t8 = row < R ? (k4 >> 2) : -64;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const {
        const int tap = q.tap0 + c.t8;
        const int ky = K == 5 ? (tap * 13) >> 6 : (tap * 11) >> 5, kx = tap - K * ky;       // tap / 5, tap / 3 for 0 <= tap < 32
        const bool ok = (unsigned)tap < (unsigned)(K * K) && (unsigned)(c.y0 + ky) < (unsigned)hb && (unsigned)(c.x0 + kx) < (unsigned)wb;
        return bload4(q.rs, ok ? (uint32_t)(c.pb + ky * wb + kx) * 16u : OOB);
    }
};

// Plain k-major matrix V[k][r] (single buffer); ragged K allowed.  seglen > 0: the conv filter
// [25][cin][cout] read in KmConvGather's K order (slice outer, tap inner).
struct NmPlain {
    static constexpr bool KM = false;
    const float* p0; int64_t ld0;
    const float* p1; int64_t ld1;   // unused here (see NmPlain2)
    int rsplit;
    int R;           // valid r
    int K;           // valid k
    const float* zeros;
    int seglen = 0, ntap = 25;
    struct Pos { rsrc_t rs; int kleft; };
    struct Ctx { uint32_t v; int kk; };
    __device__ int nchunks_of(int) const { return (K + KC - 1) / KC; }
    __device__ Pos pos(int, int chunk) const {
        int k0 = chunk * KC;
        if (seglen) {
            int seg, slice;
            tap_slice(chunk, ntap, seg, slice);
            k0 = seg * seglen + slice * KC;
        }
        return Pos{make_rsrc(p0 + (int64_t)k0 * ld0), K - chunk * KC};
    }
    __device__ void prep(int, int kk, int r4, Ctx& c) const {
        c.kk = kk;
        c.v = r4 < R ? (uint32_t)((int64_t)kk * ld0 + r4) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, c.kk < q.kleft ? c.v : OOB); }
};

// V[k][r] = [p0 | p1] split along r at rsplit (the Matrix gradient of trans_h0, whose input is
// concat([src_z, ctx_z])).  A lane belongs to exactly one buffer; both loads are issued and the other
// one returns zeros.
struct NmPlain2 {
    static constexpr bool KM = false;
    const float* p0; int64_t ld0;
    const float* p1; int64_t ld1;
    int rsplit;
    int R;
    int K;
    const float* zeros;
    struct Pos { rsrc_t r0, r1; int kleft; };
    struct Ctx { uint32_t v0, v1; int kk; };
    __device__ int nchunks_of(int) const { return (K + KC - 1) / KC; }
    __device__ Pos pos(int, int chunk) const {
        const int64_t k0 = (int64_t)chunk * KC;
        return Pos{make_rsrc(p0 + k0 * ld0), make_rsrc(p1 + k0 * ld1), K - chunk * KC};
    }
    __device__ void prep(int, int kk, int r4, Ctx& c) const {
        c.kk = kk;
        c.v0 = r4 < rsplit ? (uint32_t)((int64_t)kk * ld0 + r4) * 4u : OOB;
        c.v1 = (r4 >= rsplit && r4 < R) ? (uint32_t)((int64_t)kk * ld1 + (r4 - rsplit)) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const {
        const bool ok = c.kk < q.kleft;
        const float4 a = bload4(q.r0, ok ? c.v0 : OOB), b = bload4(q.r1, ok ? c.v1 : OOB);
        return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
};

// cin == 3 conv filter w[5][5][3][cb] as the B operand of KmC3Gather: k = tap * 4 + ch -> filter row tap * 3 + ch (ch < 3).
struct NmC3Weights {
    static constexpr bool KM = false;
    const float* w; int cb;
    const float* zeros;
    int ntap = 25, cs = 3;         // cs = filter rows per tap (32 where the blob keeps cin padded: the CNN executor's stem)
    struct Pos { rsrc_t rs; int tap0; };
    struct Ctx { uint32_t v; int t; };
    __device__ Pos pos(int, int chunk) const { return Pos{make_rsrc(w + (int64_t)chunk * 8 * cs * cb), 8 * chunk}; }
    __device__ void prep(int, int kk, int r4, Ctx& c) const {
        const int t = kk >> 2, ch = kk & 3;
        c.t = (ch < 3 && r4 < cb) ? t : 64;
        c.v = (uint32_t)((t * cs + ch) * cb + r4) * 4u;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, q.tap0 + c.t < ntap ? c.v : OOB); }
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
    int s = 2, pad = 1, K = 5;
    struct Pos { rsrc_t rs; int k0, ky, kx; };
    struct Ctx { int kk; uint32_t r; };           // r = channel byte offset or OOB
    __device__ int nchunks_of(int) const { return (npix + KC - 1) / KC; }
    __device__ Pos pos(int prob, int chunk) const {
        const int ky = prob / K, kx = prob - ky * K;
        return Pos{make_rsrc(big), chunk * KC, ky, kx};
    }
    __device__ void prep(int, int kk, int r4, Ctx& c) const { c.kk = kk; c.r = r4 < ca ? (uint32_t)r4 * 4u : OOB; }
    __device__ float4 load(const Ctx& c, const Pos& q) const {
        const int p = q.k0 + c.kk;
        int n, i, j;
        pd.split(p, n, i, j);
        const int y = s * i + q.ky - pad, xx = s * j + q.kx - pad;
        const bool ok = p < npix && (unsigned)y < (unsigned)hb && (unsigned)xx < (unsigned)wb;
        const uint32_t v = (uint32_t)(((n * hb + y) * wb + xx) * (int)ldb) * 4u + c.r;   // c.r == OOB keeps it out of range
        return bload4(q.rs, ok ? v : OOB);
    }
};

// Filter-gradient operand, small side, single tensor: k = pixel, rows = channels.
struct NmWgradSmall {
    static constexpr bool KM = false;
    const float* s1; int64_t ld1; int c1;
    const float* s2; int64_t ld2; int nmod2;      // unused here (see NmWgradSmall2)
    int cb;
    int hsws;
    int hsws_sh;
    int npix;
    const float* zeros;
    struct Pos { rsrc_t rs; int kleft; };
    struct Ctx { uint32_t v; int kk; };
    __device__ Pos pos(int, int chunk) const { return Pos{make_rsrc(s1 + (int64_t)chunk * KC * ld1), npix - chunk * KC}; }
    __device__ void prep(int, int kk, int r4, Ctx& c) const {
        c.kk = kk;
        c.v = r4 < cb ? (uint32_t)((int64_t)kk * ld1 + r4) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, c.kk < q.kleft ? c.v : OOB); }
};

// Small side of a decoder filter gradient: channels of [s1 | s2], s2 = ctx skip with image index
// img % nmod2 (img < 2 * nmod2: the two decoder passes).  Both loads are issued; a lane's other one is OOB.
struct NmWgradSmall2 {
    static constexpr bool KM = false;
    const float* s1; int64_t ld1; int c1;
    const float* s2; int64_t ld2; int nmod2;
    int cb;          // c1 + c2
    int hsws;        // pixels per image
    int hsws_sh;
    int npix;
    const float* zeros;
    struct Pos { rsrc_t r1, r2; int k0, kleft; };
    struct Ctx { uint32_t v1, r2; int kk; };       // r2 = channel byte offset inside s2, or OOB
    __device__ Pos pos(int, int chunk) const {
        return Pos{make_rsrc(s1 + (int64_t)chunk * KC * ld1), make_rsrc(s2), chunk * KC, npix - chunk * KC};
    }
    __device__ void prep(int, int kk, int r4, Ctx& c) const {
        c.kk = kk;
        c.v1 = r4 < c1 ? (uint32_t)((int64_t)kk * ld1 + r4) * 4u : OOB;
        c.r2 = (r4 >= c1 && r4 < cb) ? (uint32_t)(r4 - c1) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const {
        const bool ok = c.kk < q.kleft;
        const int p = q.k0 + c.kk, wrap = nmod2 * hsws;
        const int p2 = p >= wrap ? p - wrap : p;                          // pixel of image img % nmod2
        const float4 a = bload4(q.r1, ok ? c.v1 : OOB);
        const float4 b = bload4(q.r2, ok ? (uint32_t)(p2 * (int)ld2) * 4u + c.r2 : OOB);
        return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
};

// ---- power-of-two grids: the filter gradient's K axis re-tiled into 32-pixel PATCHES ---------------
// K (the pixel sum) may be walked in any order.  With hs, ws powers of two a chunk is a patch of R
// stacked rows x C columns (C = min(ws, 32), R = 32 / C) of the image stack [imgs*hs, ws]; lane kk owns
// (di, dj) = (kk / C, kk % C) for the whole kernel, the patch origin (rr0, j0) is wave-uniform.  Because
// hb = 2 hs, row rr of the small stack meets rows 2 rr + ky - 1 of the big stack, so both operands are
// "uniform base + invariant per-lane offset"; only the image-border validity depends on the chunk.
struct PatchGeo {
    int hs, ws, c_sh, C, R, ncol_sh, rows_total;   // ncol_sh = log2(ws / C)
    __device__ __forceinline__ void origin(int chunk, int& rr0, int& j0) const {
        rr0 = (chunk >> ncol_sh) * R;
        j0 = (chunk & ((1 << ncol_sh) - 1)) << c_sh;
    }
    __device__ __forceinline__ int nchunks() const { return ((rows_total + R - 1) / R) << ncol_sh; }
};
inline bool patch_ok(int hs, int ws) { auto p2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; }; return p2(hs) && p2(ws); }
inline PatchGeo make_patch(int nimg, int hs, int ws) {
    auto lg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return s; };
    const int C = ws < 32 ? ws : 32;
    return PatchGeo{hs, ws, lg(C), C, 32 / C, lg(ws / C), nimg * hs};
}

struct NmWgradBigP {
    static constexpr bool KM = false;
    const float* big; int64_t ldb; int ca;
    int wb;
    PatchGeo g;
    const float* zeros;
    struct Pos { rsrc_t rs; int i0, j0, rows_left; bool top, bot, lef, rig; };
    struct Ctx { int di, dj; uint32_t v; };
    __device__ int nchunks_of(int) const { return g.nchunks(); }
    __device__ Pos pos(int prob, int chunk) const {
        const int ky = prob / 5, kx = prob - ky * 5;
        int rr0, j0;
        g.origin(chunk, rr0, j0);
        const float* base = big + ((int64_t)(2 * rr0 + ky - 1) * wb + (2 * j0 + kx - 1)) * ldb;
        return Pos{make_rsrc(base), rr0 & (g.hs - 1), j0, g.rows_total - rr0, ky == 0, ky >= 3, kx == 0, kx >= 3};
    }
    __device__ void prep(int, int kk, int r4, Ctx& c) const {
        c.di = kk >> g.c_sh; c.dj = kk & (g.C - 1);
        c.v = r4 < ca ? (uint32_t)(((2 * c.di) * wb + 2 * c.dj) * (int)ldb + r4) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const {
        const int i = (q.i0 + c.di) & (g.hs - 1), j = q.j0 + c.dj;
        // y = 2i+ky-1 in [0, 2hs): fails for (i == 0, ky == 0) and (i == hs-1, ky >= 3); same in x
        const bool bad = c.di >= q.rows_left || (q.top && i == 0) || (q.bot && i == g.hs - 1) || (q.lef && j == 0) ||
                         (q.rig && j == g.ws - 1);
        return bload4(q.rs, bad ? OOB : c.v);
    }
};

struct NmWgradSmallP {
    static constexpr bool KM = false;
    const float* s1; int64_t ld1; int cb;
    PatchGeo g;
    const float* zeros;
    struct Pos { rsrc_t rs; int rows_left; };
    struct Ctx { int di; uint32_t v; };
    __device__ Pos pos(int, int chunk) const {
        int rr0, j0;
        g.origin(chunk, rr0, j0);
        return Pos{make_rsrc(s1 + ((int64_t)rr0 * g.ws + j0) * ld1), g.rows_total - rr0};
    }
    __device__ void prep(int, int kk, int r4, Ctx& c) const {
        c.di = kk >> g.c_sh;
        const int dj = kk & (g.C - 1);
        c.v = r4 < cb ? (uint32_t)((c.di * g.ws + dj) * (int)ld1 + r4) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, c.di < q.rows_left ? c.v : OOB); }
};

struct NmWgradSmall2P {
    static constexpr bool KM = false;
    const float* s1; int64_t ld1; int c1;
    const float* s2; int64_t ld2; int nmod2;     // s2 = ctx skip, stacked rows wrap at nmod2 * hs
    int cb;
    PatchGeo g;
    const float* zeros;
    struct Pos { rsrc_t r1, r2; int rr0, pix0, rows_left; };
    struct Ctx { int di, pix; uint32_t v1, r2; };
    __device__ Pos pos(int, int chunk) const {
        int rr0, j0;
        g.origin(chunk, rr0, j0);
        const int pix0 = rr0 * g.ws + j0;
        return Pos{make_rsrc(s1 + (int64_t)pix0 * ld1), make_rsrc(s2), rr0, pix0, g.rows_total - rr0};
    }
    __device__ void prep(int, int kk, int r4, Ctx& c) const {
        c.di = kk >> g.c_sh;
        c.pix = c.di * g.ws + (kk & (g.C - 1));
        c.v1 = r4 < c1 ? (uint32_t)(c.pix * (int)ld1 + r4) * 4u : OOB;
        c.r2 = (r4 >= c1 && r4 < cb) ? (uint32_t)(r4 - c1) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const {
        const bool ok = c.di < q.rows_left;
        const int wrap_rows = nmod2 * g.hs;
        const int pix2 = q.pix0 + c.pix - (q.rr0 + c.di >= wrap_rows ? wrap_rows * g.ws : 0);
        const float4 a = bload4(q.r1, ok ? c.v1 : OOB);
        const float4 b = bload4(q.r2, ok ? (uint32_t)(pix2 * (int)ld2) * 4u + c.r2 : OOB);
        return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
};

// ---- rectangle-ordered filter gradient -----------------------------------------------------------------
// K (the sum over output pixels and images) may be walked in any order.  For tap (ky,kx) only the output positions
// whose input pixel (s*i+ky-pad, s*j+kx-pad) lies inside the big grid contribute -- a rectangle [i0,i0+rh) x [j0,j0+rw) --
// and on the 4x4 / 8x8 grids the rest is 28 % / 14 % of all products (SAME-padding zeros).  Here K runs over that
// rectangle only, position-major: k = ((i-i0)*rw + (j-j0)) * nimg + n.  With nimg a multiple of 32 a chunk is 32
// images at ONE position: the position is wave-uniform (folded into the descriptor base) and a lane's offset
// (image kk, channels r4) is loop-invariant, so a load is a single buffer_load with no per-chunk VALU at all.
struct FastDiv {                     // x / d for 0 <= x < 2^31, d >= 1, by multiply-high (Granlund-Montgomery): no divider on the scalar unit
    uint32_t mul; int sh;
    __device__ __forceinline__ int div(int x) const { return (int)((__umulhi(mul, (uint32_t)x) + (uint32_t)x) >> sh); }
};
inline FastDiv make_fastdiv(int d) {
    int sh = 0;
    while ((1 << sh) < d) ++sh;
    return FastDiv{(uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << sh) - (uint64_t)d)) / (uint64_t)d + 1), sh};
}
struct RectGeo {
    int hs, ws, hb, wb, nimg, ipc;   // ipc = nimg / 32: chunks per position
    int s, pad, K;
    int i0[5], rh[5], j0[5], rw[5];  // per ky / per kx
    FastDiv d_ipc, d_rw[5];
    __device__ __forceinline__ int nchunks(int prob) const { const int ky = prob / K, kx = prob - ky * K; return rh[ky] * rw[kx] * ipc; }
    // chunk -> output position (i, j), first image n0, and the tap
    __device__ __forceinline__ void decode(int prob, int chunk, int& ky, int& kx, int& i, int& j, int& n0) const {
        ky = prob / K; kx = prob - ky * K;
        const int pos = d_ipc.div(chunk);
        n0 = (chunk - pos * ipc) * KC;
        const int r = d_rw[kx].div(pos);
        i = i0[ky] + r; j = j0[kx] + (pos - r * rw[kx]);
    }
};
inline bool rect_ok(int nimg) { return nimg > 0 && nimg % KC == 0; }
inline RectGeo make_rect(int nimg, int hs, int ws, int hb, int wb, int s, int pad, int K) {
    RectGeo g{};
    g.hs = hs; g.ws = ws; g.hb = hb; g.wb = wb; g.nimg = nimg; g.ipc = nimg / KC; g.s = s; g.pad = pad; g.K = K;
    auto range = [&](int k, int nbig, int nsmall, int& lo, int& len) {        // 0 <= s*i + k - pad < nbig
        int a = 0, b = nsmall;
        while (a < nsmall && s * a + k - pad < 0) ++a;
        while (b > a && s * (b - 1) + k - pad >= nbig) --b;
        lo = a; len = b - a;
    };
    for (int k = 0; k < 5; ++k) {
        if (k < K) { range(k, hb, hs, g.i0[k], g.rh[k]); range(k, wb, ws, g.j0[k], g.rw[k]); }
        else { g.i0[k] = g.j0[k] = 0; g.rh[k] = g.rw[k] = 0; }
        g.d_rw[k] = make_fastdiv(g.rw[k] > 0 ? g.rw[k] : 1);
    }
    g.d_ipc = make_fastdiv(g.ipc);
    return g;
}

struct NmWgradBigR {
    static constexpr bool KM = false;
    const float* big; int64_t ldb; int ca;
    RectGeo g;
    const float* zeros;
    struct Pos { rsrc_t rs; };
    struct Ctx { uint32_t v; };
    __device__ int nchunks_of(int prob) const { return g.nchunks(prob); }
    __device__ Pos pos(int prob, int chunk) const {
        int ky, kx, i, j, n0;
        g.decode(prob, chunk, ky, kx, i, j, n0);
        return Pos{make_rsrc(big + (((int64_t)n0 * g.hb + (g.s * i + ky - g.pad)) * g.wb + (g.s * j + kx - g.pad)) * ldb)};
    }
    __device__ void prep(int, int kk, int r4, Ctx& c) const { c.v = r4 < ca ? (uint32_t)((int64_t)kk * g.hb * g.wb * ldb + r4) * 4u : OOB; }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, c.v); }
};

struct NmWgradSmallR {
    static constexpr bool KM = false;
    const float* s1; int64_t ld1; int cb;
    RectGeo g;
    const float* zeros;
    struct Pos { rsrc_t rs; };
    struct Ctx { uint32_t v; };
    __device__ Pos pos(int prob, int chunk) const {
        int ky, kx, i, j, n0;
        g.decode(prob, chunk, ky, kx, i, j, n0);
        return Pos{make_rsrc(s1 + (((int64_t)n0 * g.hs + i) * g.ws + j) * ld1)};
    }
    __device__ void prep(int, int kk, int r4, Ctx& c) const { c.v = r4 < cb ? (uint32_t)((int64_t)kk * g.hs * g.ws * ld1 + r4) * 4u : OOB; }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, c.v); }
};

// small side = channels of [s1 | s2]; s2 is the ctx skip shared by both decoder passes (image n % nmod2, nmod2 % 32 == 0)
struct NmWgradSmall2R {
    static constexpr bool KM = false;
    const float* s1; int64_t ld1; int c1;
    const float* s2; int64_t ld2; int nmod2;
    int cb;
    RectGeo g;
    const float* zeros;
    struct Pos { rsrc_t r1, r2; };
    struct Ctx { uint32_t v1, v2; };
    __device__ Pos pos(int prob, int chunk) const {
        int ky, kx, i, j, n0;
        g.decode(prob, chunk, ky, kx, i, j, n0);
        const int n2 = n0 >= nmod2 ? n0 - nmod2 : n0;
        return Pos{make_rsrc(s1 + (((int64_t)n0 * g.hs + i) * g.ws + j) * ld1), make_rsrc(s2 + (((int64_t)n2 * g.hs + i) * g.ws + j) * ld2)};
    }
    __device__ void prep(int, int kk, int r4, Ctx& c) const {
        c.v1 = r4 < c1 ? (uint32_t)((int64_t)kk * g.hs * g.ws * ld1 + r4) * 4u : OOB;
        c.v2 = (r4 >= c1 && r4 < cb) ? (uint32_t)((int64_t)kk * g.hs * g.ws * ld2 + (r4 - c1)) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const {
        const float4 a = bload4(q.r1, c.v1), b = bload4(q.r2, c.v2);
        return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
};

// Filter gradient when the big tensor has 3 channels (read from its 4-channel copy): rows m = tap * 4 + ch (100 of 128
// used), k = output-grid pixel; a lane's float4 is the 4 channels of ONE tap at one pixel.
struct NmC3WgradBig {
    static constexpr bool KM = false;
    const float* big4;
    int hb, wb;
    PixDiv pd;
    int npix;
    const float* zeros;
    int s = 2, pad = 1;
    struct Pos { rsrc_t rs; int k0; };
    struct Ctx { int kk, dy, dx; };               // (dy, dx) = (ky - pad, kx - pad); dy = 1 << 20 marks a padded tap
    __device__ int nchunks_of(int) const { return (npix + KC - 1) / KC; }
    __device__ Pos pos(int, int chunk) const { return Pos{make_rsrc(big4), chunk * KC}; }
    __device__ void prep(int, int kk, int r4, Ctx& c) const {
        const int tap = r4 >> 2, ky = tap / 5, kx = tap - 5 * ky;
        c.kk = kk;
        c.dy = tap < 25 ? ky - pad : (1 << 20);
        c.dx = kx - pad;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const {
        const int p = q.k0 + c.kk;
        int n, i, j;
        pd.split(p, n, i, j);
        const int y = s * i + c.dy, xx = s * j + c.dx;
        const bool ok = p < npix && (unsigned)y < (unsigned)hb && (unsigned)xx < (unsigned)wb;
        return bload4(q.rs, ok ? (uint32_t)((n * hb + y) * wb + xx) * 16u : OOB);
    }
};

// ---- position-major convolutions ------------------------------------------------------------------------
// One PROBLEM per output position; rows m = image index.  Every row of a block then shares the taps that land inside
// the input grid -- a rectangle [ky0,ky1) x [kx0,kx1) of the kernel -- and K runs over those taps only: the products
// with SAME-padding zeros (28 % of a 4x4 layer, 14 % of an 8x8 layer) are never formed, no per-lane validity is left in
// the loop, and a lane's offset is loop-invariant.  The chunk -> (tap, channel slice) decode is done ONCE per chunk
// (KPos) and shared by both operands.  K order: taps outer, 32-channel slices inner.
template <class L, class = void> struct has_kpos { static constexpr bool value = false; };
template <class L> struct has_kpos<L, decltype((void)sizeof(typename L::KPos))> { static constexpr bool value = true; };
struct NoKBlock {};
template <class LA>
__device__ __forceinline__ auto make_kblock(const LA& la, int prob) {      // the block-invariant part of the decode (taps of this position)
    if constexpr (has_kpos<LA>::value) return la.kblock(prob);
    else return NoKBlock{};
}
template <class LA, class LB, class KB>
__device__ __forceinline__ void both_pos(const LA& la, const LB& lb, int prob, int c, const KB& kb, typename LA::Pos& qa, typename LB::Pos& qb) {
    if constexpr (has_kpos<LA>::value) {
        const typename LA::KPos kp = la.kpos(kb, c);
        qa = la.pos(prob, c, kp);
        qb = lb.pos(prob, c, kp);
    } else {
        qa = la.pos(prob, c);
        qb = lb.pos(prob, c);
    }
}
struct TapPos { int ky, kx, slice; };          // kernel tap and channel slice of a chunk
// seg / ntx for seg < 25, 1 <= ntx <= 7
__device__ __forceinline__ int div_small(int seg, int ntx) {
    const int m = ntx == 1 ? 256 : ntx == 2 ? 128 : ntx == 3 ? 86 : ntx == 4 ? 64 : ntx == 5 ? 52 : ntx == 6 ? 43 : 37;
    return (seg * m) >> 8;
}

struct PosGeo {
    int hs, ws, hb, wb, s, pad, K, cps;
    FastDiv d_ws, d_cps;
    int KW = 0, padx = -1;           // kernel width / left pad when they differ from K / pad (Inception's 1x7, 7x1, 1x3, 3x1)
    __host__ __device__ int kw() const { return KW ? KW : K; }
    __host__ __device__ int px() const { return padx >= 0 ? padx : pad; }
    __device__ __forceinline__ void where(int prob, int& i, int& j) const { i = d_ws.div(prob); j = prob - i * ws; }
    // taps k in [k0, k0 + nt) with 0 <= s*i + k - pad < nbig
    __device__ __forceinline__ static void range(int si, int pad, int K, int nbig, int& k0, int& nt) {
        k0 = pad - si > 0 ? pad - si : 0;
        const int k1 = nbig + pad - si < K ? nbig + pad - si : K;
        nt = k1 - k0;
    }
    __device__ __forceinline__ int nchunks(int prob) const {
        int i, j, ky0, nty, kx0, ntx;
        where(prob, i, j);
        range(s * i, pad, K, hb, ky0, nty); range(s * j, px(), kw(), wb, kx0, ntx);
        return nty * ntx * cps;
    }
    struct Blk { int ky0, kx0, ntx; };
    __device__ __forceinline__ Blk kblock(int prob) const {
        int i, j, ky0, nty, kx0, ntx;
        where(prob, i, j);
        range(s * i, pad, K, hb, ky0, nty); range(s * j, px(), kw(), wb, kx0, ntx);
        return Blk{ky0, kx0, ntx};
    }
    __device__ __forceinline__ TapPos kpos(const Blk& b, int chunk) const {
        const int seg = d_cps.div(chunk), r = div_small(seg, b.ntx);
        return TapPos{b.ky0 + r, b.kx0 + (seg - r * b.ntx), chunk - seg * cps};
    }
};
inline PosGeo make_posgeo(int hs, int ws, int hb, int wb, int s, int pad, int K, int cps) {
    return PosGeo{hs, ws, hb, wb, s, pad, K, cps, make_fastdiv(ws), make_fastdiv(cps)};
}
inline int posgeo_min_chunks(const PosGeo& g) {                  // the corner position
    auto nt = [&](int si, int nbig, int pad, int K) { int k0 = pad - si > 0 ? pad - si : 0, k1 = nbig + pad - si < K ? nbig + pad - si : K; return k1 - k0; };
    int best = 1 << 30;
    for (int i : {0, g.hs - 1})
        for (int j : {0, g.ws - 1}) { const int n = nt(g.s * i, g.hb, g.pad, g.K) * nt(g.s * j, g.wb, g.px(), g.kw()) * g.cps; if (n < best) best = n; }
    return best;
}

// conv2d operand: problem = output position (i,j), row = image.  Epilogue rowmode 4.
struct KmConvGatherQ {
    static constexpr bool KM = true;
    const float* x; int64_t ldx;
    PosGeo g;
    int nimg;
    const float* zeros;
    typedef TapPos KPos;
    typedef PosGeo::Blk KBlock;
    struct Pos { rsrc_t rs; };
    struct Ctx { uint32_t v; };
    __device__ int nchunks_of(int prob) const { return g.nchunks(prob); }
    __device__ KBlock kblock(int prob) const { return g.kblock(prob); }
    __device__ KPos kpos(const KBlock& b, int chunk) const { return g.kpos(b, chunk); }
    __device__ Pos pos(int, int, const KPos& t) const {
        return Pos{make_rsrc(x + ((int64_t)(t.ky - g.pad) * g.wb + (t.kx - g.px())) * ldx + t.slice * KC)};
    }
    __device__ void prep(int prob, int row, int k4, Ctx& c) const {
        int i, j;
        g.where(prob, i, j);
        c.v = row < nimg ? (uint32_t)((((int64_t)row * g.hb + g.s * i) * g.wb + g.s * j) * ldx + k4) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, c.v); }
};
// its filter [K][K][cin][cout] as the B operand (rows = cout)
struct NmConvWeightsQ {
    static constexpr bool KM = false;
    const float* w; int cin, cout, K;     // K = kernel WIDTH (row index = ky * K + kx)
    const float* zeros;
    struct Pos { rsrc_t rs; };
    struct Ctx { uint32_t v; };
    __device__ Pos pos(int, int, const TapPos& t) const { return Pos{make_rsrc(w + ((int64_t)(t.ky * K + t.kx) * cin + t.slice * KC) * cout)}; }
    __device__ void prep(int, int kk, int r4, Ctx& c) const { c.v = r4 < cout ? (uint32_t)(kk * cout + r4) * 4u : OOB; }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, c.v); }
};

// conv2d_transpose, stride 2: problem = class * (hs*ws) + input-grid position (i',j') (class = output parity (py,px));
// row = image; valid taps sy in [sy0, sy1) with 0 <= i' + oy - sy < hs (KmConvTGather's notation).  Epilogue rowmode 5.
struct TPosGeo {
    int hs, ws, K, pb, cps;
    FastDiv d_np, d_ws, d_cps;
    struct Where { int py, px, i, j, oy, ox, sy0, nty, sx0, ntx, pary, parx; };
    __device__ __forceinline__ static void cls(int p, int pb, int K, int& par, int& nt, int& off) {
        par = (p + pb) & 1; nt = (K - par + 1) >> 1; off = (p + pb - par) >> 1;
    }
    __device__ __forceinline__ Where where(int prob) const {
        Where w;
        const int c = d_np.div(prob), pos = prob - c * hs * ws;   // (position*4 + class was measured: -30 % fetch traffic, +20 % time)
        w.py = c >> 1; w.px = c & 1;
        w.i = d_ws.div(pos); w.j = pos - w.i * ws;
        int nty, ntx;
        cls(w.py, pb, K, w.pary, nty, w.oy); cls(w.px, pb, K, w.parx, ntx, w.ox);
        // 0 <= i + oy - sy < hs  <=>  i + oy - hs < sy <= i + oy
        w.sy0 = w.i + w.oy - hs + 1 > 0 ? w.i + w.oy - hs + 1 : 0;
        const int sy1 = w.i + w.oy + 1 < nty ? w.i + w.oy + 1 : nty;
        w.nty = sy1 - w.sy0;
        w.sx0 = w.j + w.ox - ws + 1 > 0 ? w.j + w.ox - ws + 1 : 0;
        const int sx1 = w.j + w.ox + 1 < ntx ? w.j + w.ox + 1 : ntx;
        w.ntx = sx1 - w.sx0;
        return w;
    }
    struct KPos { int sy, sx, slice, oy, ox, pary, parx; };
    __device__ __forceinline__ int nchunks(int prob) const { const Where w = where(prob); return w.nty * w.ntx * cps; }
    struct Blk { int sy0, sx0, ntx, oy, ox, pary, parx; };
    __device__ __forceinline__ Blk kblock(int prob) const {
        const Where w = where(prob);
        return Blk{w.sy0, w.sx0, w.ntx, w.oy, w.ox, w.pary, w.parx};
    }
    __device__ __forceinline__ KPos kpos(const Blk& b, int chunk) const {
        const int seg = d_cps.div(chunk), r = div_small(seg, b.ntx);
        return KPos{b.sy0 + r, b.sx0 + (seg - r * b.ntx), chunk - seg * cps, b.oy, b.ox, b.pary, b.parx};
    }
};
inline TPosGeo make_tposgeo(int hs, int ws, int K, int pb, int cps) {
    return TPosGeo{hs, ws, K, pb, cps, make_fastdiv(hs * ws), make_fastdiv(ws), make_fastdiv(cps)};
}
struct KmConvTGatherQ {
    static constexpr bool KM = true;
    const float* s1; int64_t ld1; int c1;   // c1 a multiple of KC
    const float* s2; int64_t ld2; int nmod2;
    TPosGeo g;
    int nimg;
    const float* zeros;
    typedef TPosGeo::KPos KPos;
    typedef TPosGeo::Blk KBlock;
    struct Pos { rsrc_t rs; bool second; };
    struct Ctx { uint32_t v1, v2; };
    __device__ int nchunks_of(int prob) const { return g.nchunks(prob); }
    __device__ KBlock kblock(int prob) const { return g.kblock(prob); }
    __device__ KPos kpos(const KBlock& b, int chunk) const { return g.kpos(b, chunk); }
    __device__ Pos pos(int, int, const KPos& t) const {
        const int64_t d = (int64_t)(t.oy - t.sy) * g.ws + (t.ox - t.sx);
        const int kc = t.slice * KC;
        const bool second = kc >= c1;
        return Pos{make_rsrc(second ? s2 + d * ld2 + (kc - c1) : s1 + d * ld1 + kc), second};
    }
    __device__ void prep(int prob, int row, int k4, Ctx& c) const {
        const TPosGeo::Where w = g.where(prob);
        const int64_t pix = (int64_t)w.i * g.ws + w.j, hw = (int64_t)g.hs * g.ws;
        c.v1 = row < nimg ? (uint32_t)((row * hw + pix) * ld1 + k4) * 4u : OOB;
        c.v2 = row < nimg ? (uint32_t)(((row % nmod2) * hw + pix) * ld2 + k4) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, q.second ? c.v2 : c.v1); }
};
// its filter w[ky][kx][a][b] (a = output channel = tile row, b = k) as the B operand
struct KmConvTWeightsQ {
    static constexpr bool KM = true;
    const float* w; int ca, cb, K;
    const float* zeros;
    struct Pos { rsrc_t rs; };
    struct Ctx { uint32_t v; };
    __device__ Pos pos(int, int, const TPosGeo::KPos& t) const {
        const int ky = t.pary + 2 * t.sy, kx = t.parx + 2 * t.sx;
        return Pos{make_rsrc(w + (int64_t)(ky * K + kx) * ca * cb + t.slice * KC)};
    }
    __device__ void prep(int, int row, int k4, Ctx& c) const { c.v = row < ca ? (uint32_t)(row * cb + k4) * 4u : OOB; }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, c.v); }
};

// conv2d_transpose, stride 1, position-major: out[n,i,j,a] = sum in[n, i + pad - ky, j + pad - kx, k] * w[ky][kx][a][k].
// With ky' = K-1-ky this is PosGeo's correlation (s = 1, hb = hs, wb = ws) with pad' = K-1-pad: only the taps inside the grid are
// chunks (on a 2x2 grid with K 3: 4 of 9; on 1x1: 1 of 9), and the filter tap of chunk (ky', kx') is (K-1-ky', K-1-kx').
// Two sources as in KmConvTGatherQ (the decoder's concat([d, skip]), the skip indexed by image % nmod2).  Epilogue rowmode 4.
struct KmConvT1GatherQ {
    static constexpr bool KM = true;
    const float* s1; int64_t ld1; int c1;   // c1 a multiple of KC
    const float* s2; int64_t ld2; int nmod2;
    PosGeo g;                               // make_posgeo(hs, ws, hs, ws, 1, K - 1 - pad, K, cps)
    int nimg;
    const float* zeros;
    typedef TapPos KPos;
    typedef PosGeo::Blk KBlock;
    struct Pos { rsrc_t rs; bool second; };
    struct Ctx { uint32_t v1, v2; };
    __device__ int nchunks_of(int prob) const { return g.nchunks(prob); }
    __device__ KBlock kblock(int prob) const { return g.kblock(prob); }
    __device__ KPos kpos(const KBlock& b, int chunk) const { return g.kpos(b, chunk); }
    __device__ Pos pos(int, int, const KPos& t) const {
        const int64_t d = (int64_t)(t.ky - g.pad) * g.wb + (t.kx - g.pad);
        const int kc = t.slice * KC;
        const bool second = kc >= c1;
        return Pos{make_rsrc(second ? s2 + d * ld2 + (kc - c1) : s1 + d * ld1 + kc), second};
    }
    __device__ void prep(int prob, int row, int k4, Ctx& c) const {
        int i, j;
        g.where(prob, i, j);
        const int64_t pix = (int64_t)i * g.wb + j, hw = (int64_t)g.hb * g.wb;
        c.v1 = row < nimg ? (uint32_t)((row * hw + pix) * ld1 + k4) * 4u : OOB;
        c.v2 = row < nimg ? (uint32_t)(((row % nmod2) * hw + pix) * ld2 + k4) * 4u : OOB;
    }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, q.second ? c.v2 : c.v1); }
};
// its filter w[ky][kx][a][b] (a = output channel = tile row, b = k) as the B operand
struct KmConvT1WeightsQ {
    static constexpr bool KM = true;
    const float* w; int ca, cb, K;
    const float* zeros;
    struct Pos { rsrc_t rs; };
    struct Ctx { uint32_t v; };
    __device__ Pos pos(int, int, const TapPos& t) const {
        return Pos{make_rsrc(w + (int64_t)((K - 1 - t.ky) * K + (K - 1 - t.kx)) * ca * cb + t.slice * KC)};
    }
    __device__ void prep(int, int row, int k4, Ctx& c) const { c.v = row < ca ? (uint32_t)(row * cb + k4) * 4u : OOB; }
    __device__ float4 load(const Ctx& c, const Pos& q) const { return bload4(q.rs, c.v); }
};

// ------------------------------------------------------------------------------------------------
// LDS tiles.  NT = threads of the block that cooperate on a tile.
// ------------------------------------------------------------------------------------------------
template <bool KMF, int TR, int NT>
struct Tile {
    static constexpr int FLOATS = KMF ? TR * LDK : KC * TR;
    static constexpr int NPASS = TR * KC / 4 / NT;     // float4 per thread per chunk
    static_assert(NPASS >= 1, "tile too small for the block");
    // global row (KM) handled by thread tid in pass p
    __device__ static int km_row(int tid, int p) { return (tid >> 3) + (NT / 8) * p; }
    __device__ static int km_k4(int tid) { return (tid & 7) * 4; }
    __device__ static int nm_kk(int tid, int p) { return tid / (TR / 4) + (NT / (TR / 4)) * p; }
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

// per-thread view of a loader: the loop-invariant state of the float4 this thread fetches in pass p
template <class L, int TR, int NT>
struct Fetch {
    using T = Tile<L::KM, TR, NT>;
    typename L::Ctx c[T::NPASS];
    __device__ void init(const L& l, int prob, int row0, int tid) {
#pragma unroll
        for (int p = 0; p < T::NPASS; ++p) {
            if (L::KM) l.prep(prob, row0 + T::km_row(tid, p), T::km_k4(tid), c[p]);
            else l.prep(prob, T::nm_kk(tid, p), row0 + T::nm_r4(tid), c[p]);
        }
    }
    __device__ float4 load1(const L& l, const typename L::Pos& q, int p) const { return l.load(c[p], q); }
};

// ------------------------------------------------------------------------------------------------
// The kernel.  1-D grid of gm * gn * nprob * nsplit blocks; a block is WM x WN waves, each wave owns
// (32*MI) x (32*NI) of the (32*MI*WM) x (32*NI*WN) block tile.  Shipped shapes:
//   4 waves (2x2), wave 64x64 or smaller  -> 128x128 ... 64x64 tiles   (small M or N)
//   8 waves (4x2), wave 64x128            -> 256x256 tile: half the HBM/L2 bytes per MFMA of 128x128
// ------------------------------------------------------------------------------------------------
template <class LA, class LB, int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const LA la, const LB lb, const Epi ep, int M, int N,
                                                             int nprob, int nsplit, int gm, int gn) {
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = 32 * MI * WM, TN = 32 * NI * WN;
    using TA = Tile<LA::KM, TM, NT>;
    using TB = Tile<LB::KM, TN, NT>;
    constexpr int NA = TA::NPASS, NB = TB::NPASS;       // float4 per thread per chunk
    static_assert(NA + NB <= 16, "at most one load and one store per MFMA group");
    constexpr int STAGE = TA::FLOATS + TB::FLOATS;
    // two register sets (1.5-chunk prefetch distance) only where the register file has room
    constexpr bool TWO_SETS = MI * NI <= 4;
    static_assert(TWO_SETS || NA + NB <= 8, "single register set: stores then loads must fit 16 groups");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 stages of [A tile | B tile]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN, l31 = lane & 31, h = lane >> 5;

    // 1-D grid, m-tile fastest, then n-tile, then problem, then K-split.  Problems go last-first: the
    // (1,1) parity class of a transposed conv has 9 taps against 4 for (0,0), and the longest blocks
    // must not form the tail.  (XCD-contiguous and n-tile-fastest orders were measured: -8..-20 %.)
    int rest = blockIdx.x;
    if (ep.xcd_swizzle) {                               // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of work
        int item;
        if (ep.swz_group) {
            const int per = ep.swz_group >> 3, l = rest >> 3, grp = l / per;
            item = grp * ep.swz_group + (rest & 7) * per + (l - grp * per);
        } else {
            const int per = (int)gridDim.x >> 3;            // the launcher pads the grid to a multiple of 8
            item = (rest & 7) * per + (rest >> 3);
        }
        if (item >= gm * gn * nprob * nsplit) return;
        rest = item;
    }
    const int bx = rest % gm; rest /= gm;
    const int by = rest % gn; rest /= gn;
    // transposed conv (nprob == 4): parity classes have 4/6/6/9 taps.  Dispatching 9,6,4,6 makes the two
    // blocks that share a CU (one from each half of a 2-per-CU round) sum to 13 and 12 taps, not 15 and 10.
    const int pr = rest % nprob;
    const int prob = ep.perm ? (int)ep.perm[pr] : nprob == 4 ? ((0x3201 >> (4 * (3 - pr))) & 15) : nprob - 1 - pr;
    const int split = rest / nprob;
    const int m0 = bx * TM, n0 = by * TN;

    const auto kblk = make_kblock(la, prob);
    const int nch = la.nchunks_of(prob);
    const int per = (nch + nsplit - 1) / nsplit;
    const int cb = split * per;
    const int ce = (cb + per < nch) ? cb + per : nch;

    Fetch<LA, TM, NT> fa;      // per-lane invariants of this thread's float4s
    Fetch<LB, TN, NT> fb;
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
        auto clampc = [&](int c) { return c < last ? c : last; };   // redundant tail reloads are harmless
        float4 xa[NA], xb[NB], ya[TWO_SETS ? NA : 1], yb[TWO_SETS ? NB : 1];
        // prologue: chunk cb -> stage 0 ; chunk cb+1 -> set X
        {
            typename LA::Pos qa;
            typename LB::Pos qb;
            both_pos(la, lb, prob, cb, kblk, qa, qb);
#pragma unroll
            for (int p = 0; p < NA; ++p) xa[p] = fa.load1(la, qa, p);
#pragma unroll
            for (int p = 0; p < NB; ++p) xb[p] = fb.load1(lb, qb, p);
        }
#pragma unroll
        for (int p = 0; p < NA; ++p) TA::store(smem, tid, p, xa[p]);
#pragma unroll
        for (int p = 0; p < NB; ++p) TB::store(smem + TA::FLOATS, tid, p, xb[p]);
        {
            typename LA::Pos qa;
            typename LB::Pos qb;
            both_pos(la, lb, prob, clampc(cb + 1), kblk, qa, qb);
#pragma unroll
            for (int p = 0; p < NA; ++p) xa[p] = fa.load1(la, qa, p);
#pragma unroll
            for (int p = 0; p < NB; ++p) xb[p] = fb.load1(lb, qb, p);
        }
        __syncthreads();

        // One chunk: multiply LDS stage `st` (16 MFMA groups (q,t) of MI*NI instructions).  In the gaps:
        //  TWO_SETS : groups 0..7 load chunk c+2 into (la_, lb_); groups 8..15 store (sa_, sb_) = chunk c+1
        //  one set  : groups 0..7 store (sa_, sb_) = chunk c+1;   groups 8..15 reload the same set with c+2
        auto chunk = [&](int c, int st, float4* la_, float4* lb_, float4* sa_, float4* sb_) {
            const float* sA = smem + st * STAGE;
            const float* sB = sA + TA::FLOATS;
            float* nA = smem + (st ^ 1) * STAGE;
            float* nB = nA + TA::FLOATS;
            // where chunk c+2 lives: wave-uniform, once per chunk, on the scalar unit
            typename LA::Pos qa;
            typename LB::Pos qb;
            both_pos(la, lb, prob, clampc(c + 2), kblk, qa, qb);
            float a[2][MI][4], b[2][NI][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) TA::frag(sA, (wm * MI + mi) * 32 + l31, 0, h, a[0][mi]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) TB::frag(sB, (wn * NI + ni) * 32 + l31, 0, h, b[0][ni]);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int q = g >> 2, t = g & 3;
                if (t == 0 && q < 3) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) TA::frag(sA, (wm * MI + mi) * 32 + l31, q + 1, h, a[(q + 1) & 1][mi]);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) TB::frag(sB, (wn * NI + ni) * 32 + l31, q + 1, h, b[(q + 1) & 1][ni]);
#ifdef CTX_PIN_FRAGS
                    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above the MFMAs (A/B switch)
#endif
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mi][t], b[q & 1][ni][t], acc[mi][ni], 0, 0, 0);
                // gap after group g.  Two sets: loads (into the free set) in groups [0, NA+NB), stores (of the
                // other set) in groups [16-NA-NB, 16) -- they may overlap.  One set: stores first, then reloads.
                constexpr int NLS = NA + NB;
                const int ld = TWO_SETS ? g : g - 8, st_ = TWO_SETS ? g - (16 - NLS) : g;
                if (ld >= 0 && ld < NLS) {
                    if (ld < NA) la_[ld] = fa.load1(la, qa, ld);
                    else lb_[ld - NA] = fb.load1(lb, qb, ld - NA);
                }
                if (st_ >= 0 && st_ < NLS) {
                    if (st_ < NA) TA::store(nA, tid, st_, sa_[st_]);
                    else TB::store(nB, tid, st_ - NA, sb_[st_ - NA]);
                }
            }
            __syncthreads();
        };
        if (TWO_SETS) {
            for (int c = cb; c < ce; c += 2) {
                chunk(c, 0, ya, yb, xa, xb);
                if (c + 1 < ce) chunk(c + 1, 1, xa, xb, ya, yb);
            }
        } else {
            for (int c = cb; c < ce; ++c) chunk(c, (c - cb) & 1, xa, xb, xa, xb);
        }
    }

    // D layout (32x32 MFMA): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    // position-major launches (rowmode 4 / 5): the destination pixel is LINEAR in the row (= image) index; the
    // block-uniform part is resolved once here, outside the unrolled loops
    const RowMap rmap = epi_rowmap(ep, prob);
    if (ep.slab) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= M) continue;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + (wn * NI + ni) * 32 + l31;
                    if (n < N) ep.slab[(((int64_t)split * nprob + prob) * M + m) * N + n] = acc[mi][ni][r];
                }
            }
        }
        return;
    }
    // Everything an element needs from memory (bias, skip-gradient adds, the saved activation behind lrelu') is requested for GR = 4 / NI
    // rows at a time (four loads per term in flight: a larger batch raises the kernel's register count above what its main loop needs, and the small kernels of the side lanes then no longer fit beside it on a SIMD -- measured: conv family -1 % serialised, step +0.03 ms) and only then applied and stored.  epi_store per element -- load, wait, store -- made every one of the 16 MI
    // row steps a store round trip: the s_waitcnt vmcnt(0) in front of a loaded value also waits for the stores issued before it
    // (the ISA of the conv forward kernel had 128 loads, each behind its own vmcnt(0), between its 96 stores -- the bias was re-read
    // per element because the stores in between might alias it).
    constexpr int GR = 4 / NI > 0 ? 4 / NI : 1;
    int nn[NI];
    bool nok[NI];
    float bv[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        nn[ni] = n0 + (wn * NI + ni) * 32 + l31;
        nok[ni] = nn[ni] < N;
        bv[ni] = (ep.bias && nok[ni]) ? ep.bias[nn[ni]] : 0.f;
    }
    const float lk = ep.lrelu == 2 ? 0.f : LEAK;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int g = 0; g < 16 / GR; ++g) {
            int64_t pix[GR];
            bool ok[GR];
            float a1[GR][NI], a2[GR][NI], mk[GR][NI];
#pragma unroll
            for (int j = 0; j < GR; ++j) {
                const int r = GR * g + j;
                const int m = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                ok[j] = m < M;
                pix[j] = 0;
                if (ok[j]) {
                    if (rmap.linear) pix[j] = rmap.base + (int64_t)m * rmap.stride;
                    else ok[j] = epi_row(ep, prob, m, pix[j]);
                }
                if (!ok[j]) pix[j] = 0;                               // (loads below stay inside the tensors)
            }
            if (ep.add1) {
#pragma unroll
                for (int j = 0; j < GR; ++j) {
                    const int64_t pa = (ep.add1_mod && pix[j] >= ep.add1_mod) ? pix[j] - ep.add1_mod : pix[j];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) a1[j][ni] = ep.add1[pa * ep.lda1 + (nok[ni] ? nn[ni] : 0)];
                }
            }
            if (ep.add2) {
#pragma unroll
                for (int j = 0; j < GR; ++j)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) a2[j][ni] = ep.add2[pix[j] * ep.lda2 + (nok[ni] ? nn[ni] : 0)];
            }
            if (ep.mask) {
#pragma unroll
                for (int j = 0; j < GR; ++j)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) mk[j][ni] = ep.mask[pix[j] * ep.ldm + ((nok[ni] && nn[ni] < ep.nsplit) ? nn[ni] : 0)];
            }
#pragma unroll
            for (int j = 0; j < GR; ++j) {
                if (!ok[j]) continue;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (!nok[ni]) continue;
                    float v = acc[mi][ni][GR * g + j] + bv[ni];
                    if (ep.add1) v += a1[j][ni];
                    if (ep.add2) v += a2[j][ni];
                    if (ep.lrelu) v = fmaxf(v, lk * v);
                    if (nn[ni] < ep.nsplit) {
                        if (ep.mask) v *= (mk[j][ni] >= 0.f) ? 1.f : LEAK;
                        ep.out1[(int64_t)prob * ep.prob_stride + pix[j] * ep.ld1 + nn[ni]] = v;
                    } else {
                        ep.out2[pix[j] * ep.ld2 + (nn[ni] - ep.nsplit)] = v;
                    }
                }
            }
        }
    }
}

}  // namespace ctx
