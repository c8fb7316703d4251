// dconv2.h -- round 5: the narrow-channel forward kernel re-built around what the phase stamps of round 5 showed
// (profiles/round5_a_dconv_phase_stamps.txt).  dconv_fwd_kernel (dconv.h) runs its eight waves in lock step through
//     issue the next tile's loads (registers) | MFMA loops | barrier | write the registers to LDS | epilogue | barrier
// and only the second of these uses the matrix pipe: on ContextAEReal's h1_conv forward 16.7 k of 25.1 k cycles per tile, the rest
// being 3.3 k of load ISSUE (96 wave-loads through the address unit with every wave waiting on it), 2.1 k of ds_write_b128, 2.2 k of
// barrier skew and 0.9 k of epilogue.  A second tile buffer would let the loads land beside the MFMA loop, but the halo tile of a
// 32-channel stride-2 layer is 94 KB and its filter 51 KB.
//
// Here the tile is cut along K: a SLICE is 16 input channels of the halo tile (47 KB for that layer), two slice buffers alternate,
// and the slices are filled by LDS-DMA (`global_load_lds_dwordx4`: no staging registers, no ds_write pass, no issue stall -- the
// request is one instruction per KB) while the other buffer is multiplied.  Per slice: one barrier.  The accumulators live across
// the slices of a tile; a 16-channel layer has one slice per tile and simply double-buffers whole tiles.
//
// LDS-DMA writes LINEAR (wave-uniform base + 16 B x lane), so the slice image is dense -- [pixel][16 channels], 64 B per pixel, no
// padding -- and what the padded rows of dconv.h did for the banks is done on the SOURCE side: for a stride-2 layer a tile row is
// stored [even columns | odd columns], so that the 16 pixels a fragment read touches (columns 2x + dx) are consecutive in LDS.
// Out-of-image halo pixels are fetched from a 16-byte zero word.
//
// Work items are UNITS: one 16-pixel row block x one 16-column block of the filter.
//
// A block has TWELVE waves in two ROLES.  Waves 0-7 (two per SIMD) COMPUTE: unit u = wave + 8 * i, all MFMA loops, the epilogue.
// Waves 8-11 (one per SIMD) LOAD: they issue the DMA requests of the next slice -- ~25 address instructions and the DMA instruction,
// ~400 cycles a piece, 47 pieces for a 47 KB slice -- on the vector ALU while the matrix pipe of their SIMD runs the compute waves'
// MFMAs, and meet them at the slice's closing barrier.  Measured on the way here (profiles/round5_a_dconv_phase_stamps.txt):
//   * eight waves doing both in lock step: the requests were 3.2 k of a slice's 12.9 k cycles, in front of the loop, matrix pipe idle;
//   * the requests INSIDE the MFMA loop: nothing moved -- two waves in their MFMA loops already keep a SIMD's pipe 85 % busy, so
//     whatever else those waves do is added, not hidden;
//   * four compute waves + four loaders: the requests vanish from the compute waves' time, but ONE wave per SIMD runs its MFMA loop at
//     53-68 % (every LDS wait and every pipeline fill of a short tap class is exposed).
//
// The epilogue is split: the terms it needs from memory are requested before the class's MFMA loop and folded into the accumulators
// right behind it; the STORES wait until the slice's closing barrier has passed (they are the youngest memory operations then, and
// the next wait for the memory counter is a whole MFMA loop away: no wave ever waits for a store round trip).
#pragma once
#include "dconv.h"

namespace ctx {

constexpr int DC2_NCW = 8, DC2_NLW = 4, DC2_THREADS = 64 * (DC2_NCW + DC2_NLW);      // compute waves, loader waves

struct DcFwd2 {                                   // what dconv_fwd2_kernel needs beyond DcFwd
    int IWP;                                       // pixels per tile row in LDS (S == 2: 2 * IWH, even columns first)
    int IWH;                                       // S == 2: columns per parity half
    int slice_floats;                              // one slice buffer, a multiple of 256 floats (whole 1 KB LDS-DMA pieces)
    unsigned div_mul;                              // pi / IWP = (pi * div_mul) >> 20 for every pixel index of a slice (checked by the launcher)
    const float* zeros;                            // >= 16 bytes of zeros in device memory
};

template <int NSL /* 16-channel slices per tile */, int MI /* units per wave */, bool NB2 /* 32 filter columns (two column blocks) */,
          int NCLS /* 1: conv / stride-1 transposed conv, 4: stride-2 transposed conv */, int OCC>
__global__ __launch_bounds__(DC2_THREADS, 3 * OCC) void dconv_fwd2_kernel(const DcFwd P, const DcFwd2 Q, int ntiles, int nslots) {
    constexpr int NBT = NB2 ? 2 : 1, NP = NBT * 16;
    constexpr int C4T = NSL * 4;                            // float4 rows of the packed filter per tap
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    float* W4 = smem + 2 * Q.slice_floats;
    int* tab = reinterpret_cast<int*>(W4 + (size_t)nslots * NSL * 16 * NP);

    // ---- resident filter: columns [n0, n0 + NP) of the packed image wp[slot * C4T + kq][NPT][4]; tap table
    {
        const float4* src = reinterpret_cast<const float4*>(P.wp);
        float4* dst = reinterpret_cast<float4*>(W4);
        const int total = nslots * C4T * NP;
        for (int i = tid; i < total; i += DC2_THREADS) {
            const int row = i / NP, n = i - row * NP;
            dst[i] = src[(int64_t)row * P.NPT + P.n0 + n];
        }
        if (tid <= nslots) {                                 // byte offset of the tap's pixel relative to a row block's first pixel
            int slot = tid < nslots ? tid : nslots - 1, ci = 0;
            while (ci + 1 < NCLS && slot >= P.cls[ci + 1].pslot0) ++ci;
            DcClass cl;
            switch (ci) { case 0: cl = P.cls[0]; break; case 1: cl = P.cls[1]; break; case 2: cl = P.cls[2]; break; default: cl = P.cls[3]; break; }
            const int e = slot - cl.pslot0, q = (e * cl.mdiv) >> 8, r = e - q * cl.ntx;
            const int dy = cl.dy0 + cl.sgn * q, dx = cl.dx0 + cl.sgn * r;
            tab[tid] = (dy * Q.IWP + (P.S == 2 ? (dx >> 1) + (dx & 1) * Q.IWH : dx)) * 64;
        }
    }

    auto tile_org = [&](int t, int& img, int& ty0, int& tx0) {
        const int txi = t % P.tiles_x; t /= P.tiles_x;
        const int tyi = t % P.tiles_y;
        img = t / P.tiles_y; ty0 = tyi * P.TH; tx0 = txi * P.TW;
    };
    // ---- LDS-DMA of a slice: piece j (1 KB = 16 pixels x 16 channels) is requested by wave j % 8.  fill_begin fixes the slice (tile,
    // channels, buffer), fill_step requests this wave's next piece, if any.
    const int npieces = Q.slice_floats >> 8;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    const float* f_base = nullptr;
    int f_ld = 0, f_iy0 = 0, f_ix0 = 0, fj = npieces;
    unsigned f_dst = 0;
    const bool loader = wv >= DC2_NCW;
    const int lw = wv - DC2_NCW;                             // loader waves: piece j = lw + 4 * i
    auto fill_begin = [&](int t, int s, int buf) {
        int img, ty0, tx0;
        tile_org(t, img, ty0, tx0);
        f_iy0 = P.S * ty0 + P.y_org; f_ix0 = P.S * tx0 + P.x_org;
        const int c = 16 * s + 4 * (lane & 3);               // this lane's first channel
        const bool second = c >= P.c1;
        f_base = second ? P.x2 + (int64_t)(P.nmod2 ? img % P.nmod2 : 0) * P.hin * P.win * P.ld2 + (c - P.c1)
                        : P.x1 + (int64_t)img * P.hin * P.win * P.ld1 + c;
        f_ld = second ? P.ld2 : P.ld1;
        f_dst = lds0 + (unsigned)(buf * Q.slice_floats) * 4u;
        fj = lw;
    };
    auto fill_src = [&](int j) -> const float* {
        const int pi = j * 16 + (lane >> 2);
        const int iy = (int)(((unsigned)pi * Q.div_mul) >> 20), qx = pi - iy * Q.IWP;
        const int ix = P.S == 2 ? (qx < Q.IWH ? 2 * qx : 2 * (qx - Q.IWH) + 1) : qx;
        const int gy = f_iy0 + iy, gx = f_ix0 + ix;
        const bool ok = iy < P.IH && ix < P.IW && (unsigned)gy < (unsigned)P.hin && (unsigned)gx < (unsigned)P.win;
        return ok ? f_base + (int64_t)(gy * P.win + gx) * f_ld : Q.zeros;
    };
    auto fill_step = [&]() {
        if (fj + DC2_NLW < npieces) {                        // two pieces at a time: their address chains interleave
            const float* s0 = fill_src(fj);
            const float* s1 = fill_src(fj + DC2_NLW);
            const unsigned d0 = __builtin_amdgcn_readfirstlane(f_dst + (unsigned)fj * 1024u), d1 = d0 + DC2_NLW * 1024u;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(s0), "s"(d0) : "memory");
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(s1), "s"(d1) : "memory");
            fj += 2 * DC2_NLW;
        } else if (fj < npieces) {                           // (wave-uniform)
            const float* src = fill_src(fj);
            // (inline asm, not __builtin_amdgcn_global_load_lds: the compiler cannot tell the two slice buffers apart and puts
            // s_waitcnt vmcnt(0) -- a wait for THIS request -- in front of the first ds_read of the MFMA loop behind it.  The kernel
            // waits for its DMA itself, once per slice, at the closing barrier; nothing else in it uses m0.  s_nop: an SALU write of
            // m0 needs one wait state before an LDS-DMA reads it -- without it a piece landed at the previous piece's address now and then.)
            const unsigned dst = __builtin_amdgcn_readfirstlane(f_dst + (unsigned)fj * 1024u);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory");
            fj += DC2_NLW;
        }
    };

    // ---- a compute wave's units: u = wv + 8 * mi  ->  row block u / NBT, column block u % NBT
    const int rbw = P.TW >> 4;                               // row blocks per tile row
    const int nunits = P.TH * rbw * NBT;
    int abase[MI], boff[MI], ncol[MI];
    float4 bias_v[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int u = wv + DC2_NCW * mi, rb = NB2 ? u >> 1 : u, nbo = NB2 ? u & 1 : 0;
        const int ty = rb / rbw, tx = (rb - ty * rbw) * 16 + l15;
        abase[mi] = u < nunits ? ((P.S * ty) * Q.IWP + tx) * 64 + kg * 16 : kg * 16;        // bytes (S == 2: column 2 tx + dx sits at tx + its tap's offset)
        boff[mi] = nbo * 256;
        ncol[mi] = P.n0 + nbo * 16 + 4 * kg;
        bias_v[mi] = (P.ep.bias && ncol[mi] < P.N) ? ldg4(P.ep.bias + ncol[mi]) : zero4();
    }
    const int nmy = loader ? 0 : wv < nunits ? (nunits - wv + DC2_NCW - 1) / DC2_NCW : 0;     // units of this wave: MI or MI - 1 (the launcher's tiles), 0 only when MI == 1

    f32x4 acc[NCLS][MI];
    int64_t ppix[NCLS][MI];                                  // output pixel of (class, unit); < 0: none
    const char* w4b = reinterpret_cast<const char*>(W4) + ((size_t)kg * NP + l15) * 16;

    int t = blockIdx.x;
    int q = 0;
    if (loader && t < ntiles) { fill_begin(t, 0, 0); while (fj < npieces) fill_step(); }
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
    __syncthreads();
#ifdef DC_TRACE
    int trace_i = 0;
#define DC2_STAMP(k) do { if (P.trace && lane == 0 && wv < 8 && blockIdx.x < 8 && trace_i < 32) P.trace[(((size_t)blockIdx.x * DC_NW + (wv & 7)) * 32 + trace_i) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define DC2_STAMP(k) do {} while (0)
#endif
    for (; t < ntiles; t += gridDim.x) {
        int img, ty0, tx0;
        tile_org(t, img, ty0, tx0);
#pragma unroll
        for (int s = 0; s < NSL; ++s, ++q) {
            const char* tile = reinterpret_cast<const char*>(smem + (q & 1) * Q.slice_floats);
            const bool last = s == NSL - 1;
            DC2_STAMP(0);
            // the epilogue's memory terms of the tile's last slice: requested in front of the class's MFMA loop and consumed behind it.
            // The compiler does not see the DMA requests (inline asm), so its wait for these loads is a wait for everything in flight,
            // the next slice's DMA included -- which has had most of an MFMA loop by then: the launcher puts the LONGEST tap class first
            // (the nine taps of a stride-2 transposed conv's (1, 1) parity class)
            float4 p1[MI], p2[MI], pm[MI];
            auto pre = [&](int ci, const DcClass& cl) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int u = wv + DC2_NCW * mi, rb = NB2 ? u >> 1 : u;
                    const int ty = rb / rbw, x = tx0 + (rb - ty * rbw) * 16 + l15, y = ty0 + ty;
                    const bool ok = u < nunits && y < P.hlog && x < P.wlog;
                    const int64_t pp = ok ? ((int64_t)img * P.hout + (P.osc * y + cl.oy)) * P.wout + (P.osc * x + cl.ox) : 0;   // (a pixel that exists)
                    ppix[ci][mi] = ok ? pp : -1;
                    const int nc = ncol[mi] < P.N ? ncol[mi] : 0;
                    if (P.ep.add1) {
                        const int64_t pa = (P.ep.add1_mod && pp >= P.ep.add1_mod) ? pp - P.ep.add1_mod : pp;
                        p1[mi] = ldg4(P.ep.add1 + pa * P.ep.lda1 + nc);
                    }
                    if (P.ep.add2) p2[mi] = ldg4(P.ep.add2 + pp * P.ep.lda2 + nc);
                    if (P.ep.mask) pm[mi] = ldg4(P.ep.mask + pp * P.ep.ldm + (nc < P.ep.nsplit ? nc : 0));
                }
            };
            if (loader) {   // the next slice (of this tile, or slice 0 of the block's next tile) goes into the buffer the previous slice was read from
                const int ns = s + 1 < NSL ? s + 1 : 0, nt = s + 1 < NSL ? t : t + (int)gridDim.x;
                if (nt < ntiles) { fill_begin(nt, ns, (q + 1) & 1); while (fj < npieces) fill_step(); }
            }
            DC2_STAMP(1);
            if (!loader) {
            if (last) pre(0, P.cls[0]);
#pragma unroll
            for (int ci = 0; ci < NCLS; ++ci) {
                DcClass cl;
                switch (ci) { case 0: cl = P.cls[0]; break; case 1: cl = P.cls[1]; break; case 2: cl = P.cls[2]; break; default: cl = P.cls[3]; break; }
                if (s == 0) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[ci][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (last && ci > 0) pre(ci, cl);
                const int e0 = cl.pslot0, e1 = cl.pslot0 + cl.ntaps;
                auto run = [&](auto nmi_c) {
                    constexpr int NMI = decltype(nmi_c)::value;
                    constexpr int NBF = NB2 ? NMI : 1;       // B fragments: one per unit, or one shared by all (16 filter columns)
                    float4 a4[2][NMI], b4[2][NBF];
                    int tnext = tab[e0];
                    const char* bp = w4b + ((size_t)(e0 * NSL + s) * 4) * NP * 16;
                    auto fetch = [&](int buf) {
                        const int to = tnext;
#pragma unroll
                        for (int mi = 0; mi < NMI; ++mi) a4[buf][mi] = *reinterpret_cast<const float4*>(tile + abase[mi] + to);
#pragma unroll
                        for (int nb = 0; nb < NBF; ++nb) b4[buf][nb] = *reinterpret_cast<const float4*>(bp + (NB2 ? boff[nb] : 0));
                        bp += (size_t)NSL * 64 * NP;
                    };
                    // a wave with ONE unit would run one dependent MFMA chain (40-cycle latency per 32-cycle issue): its odd taps go to a
                    // second accumulator, added at the end
                    constexpr bool DUAL = NMI == 1;
                    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
                    auto mma = [&](int buf) {
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                            for (int mi = 0; mi < NMI; ++mi) {
                                const float4& A = a4[buf][mi];
                                const float4& B = b4[buf][NB2 ? mi : 0];
                                const float av = tt == 0 ? A.x : tt == 1 ? A.y : tt == 2 ? A.z : A.w;
                                const float bv = tt == 0 ? B.x : tt == 1 ? B.y : tt == 2 ? B.z : B.w;
                                if (DUAL && buf == 1) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, accb, 0, 0, 0);
                                else acc[ci][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[ci][mi], 0, 0, 0);      // D^T: rows = channels, cols = pixels
                            }
                        }
                    };
                    // software pipeline over the class's taps: fragments one tap ahead, the table entry two ahead
                    fetch(0);
                    int e = e0;
                    int tahead = tab[e0 + 1];                // (one entry past the last tap exists)
                    for (; e + 2 < e1; e += 2) {
                        tnext = tahead; tahead = tab[e + 2];
                        fetch(1);
                        __builtin_amdgcn_sched_barrier(0);
                        mma(0);
                        __builtin_amdgcn_sched_barrier(0);
                        tnext = tahead; tahead = tab[e + 3];
                        fetch(0);
                        __builtin_amdgcn_sched_barrier(0);
                        mma(1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (e + 1 < e1) { tnext = tahead; fetch(1); __builtin_amdgcn_sched_barrier(0); mma(0); mma(1); }
                    else mma(0);
                    if constexpr (DUAL) { for (int r = 0; r < 4; ++r) acc[ci][0][r] += accb[r]; }
                };
                if (nmy == MI) run(std::integral_constant<int, MI>{});
                else if constexpr (MI > 1) { if (nmy == MI - 1) run(std::integral_constant<int, MI - 1>{}); }
                if (last) {                                  // fold bias / adds / lrelu / lrelu' into the accumulators (every lane consumes every load)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int n = ncol[mi];
                        float v[4] = {acc[ci][mi][0] + bias_v[mi].x, acc[ci][mi][1] + bias_v[mi].y, acc[ci][mi][2] + bias_v[mi].z, acc[ci][mi][3] + bias_v[mi].w};
                        if (P.ep.add1) { v[0] += p1[mi].x; v[1] += p1[mi].y; v[2] += p1[mi].z; v[3] += p1[mi].w; }
                        if (P.ep.add2) { v[0] += p2[mi].x; v[1] += p2[mi].y; v[2] += p2[mi].z; v[3] += p2[mi].w; }
                        if (P.ep.lrelu) {
                            const float lk = P.ep.lrelu == 2 ? 0.f : LEAK;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lk * v[r]);
                        }
                        if (P.ep.mask) {
                            const bool m1 = n < P.ep.nsplit;
                            v[0] *= (m1 && pm[mi].x < 0.f) ? LEAK : 1.f; v[1] *= (m1 && pm[mi].y < 0.f) ? LEAK : 1.f;
                            v[2] *= (m1 && pm[mi].z < 0.f) ? LEAK : 1.f; v[3] *= (m1 && pm[mi].w < 0.f) ? LEAK : 1.f;
                        }
                        acc[ci][mi] = f32x4{v[0], v[1], v[2], v[3]};
                    }
                }
            }
            }   // (!loader)
            DC2_STAMP(2);
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the next slice has landed (everything older has, too)
            DC2_STAMP(3);
            __syncthreads();                                 // ... for every wave; and every wave is done reading this slice
            DC2_STAMP(4);
            if (last && !loader) {                           // the stores: youngest memory operations, retired under the next MFMA loop
#pragma unroll
                for (int ci = 0; ci < NCLS; ++ci)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int n = ncol[mi];
                        if (ppix[ci][mi] < 0 || n >= P.N) continue;
                        const float4 v = make_float4(acc[ci][mi][0], acc[ci][mi][1], acc[ci][mi][2], acc[ci][mi][3]);
                        if (n < P.ep.nsplit) *reinterpret_cast<float4*>(P.ep.out1 + ppix[ci][mi] * P.ep.ld1 + n) = v;
                        else *reinterpret_cast<float4*>(P.ep.out2 + ppix[ci][mi] * P.ep.ld2 + (n - P.ep.nsplit)) = v;
                    }
            }
            DC2_STAMP(5);
        }
#ifdef DC_TRACE
        DC2_STAMP(6);
        ++trace_i;
#endif
    }
}

}  // namespace ctx
