#!/bin/bash
# Measurement build: per-block cycle stamps inside igemm_kernel (prologue | K loop | epilogue) for the launches with nprob == $IG_TRACE_NPROB
# (default 25: the filter gradients), printed at exit for the last step.   gpurun -- 'bash tools/ig_trace.sh OUTFILE [nprob]'
# Applies the patch at the end of this file's build step (the stamps, macro-guarded) to the (throw-away) GPU-box copy of the tree and rebuilds libctxtrans.so with
# -DIG_TRACE there; the committed sources and the shipped library carry no instrumentation.
OUT=${1:-gpurun_out/ig_trace.txt}; export IG_TRACE_NPROB=${2:-25}
mkdir -p "$(dirname "$OUT")"
patch -p1 > /dev/null <<'IG_TRACE_PATCH_EOF' || { echo "patch failed"; exit 1; }
diff --git a/imitation_from_observation_amd/csrc/gemm_launch.h b/imitation_from_observation_amd/csrc/gemm_launch.h
index 01fc6c6..eaa8b40 100644
--- a/imitation_from_observation_amd/csrc/gemm_launch.h
+++ b/imitation_from_observation_amd/csrc/gemm_launch.h
@@ -8,6 +8,7 @@
 #include <string>
 #include <type_traits>
 #include <typeinfo>
+#include <vector>
 
 #include "launch.h"
 #include "igemm_split.h"
@@ -33,6 +34,52 @@ static void launch_tile_split(hipStream_t s, const LA& a, const LB& b, Epi ep, i
     hipLaunchKernelGGL((igemm_split_kernel<LA, LB, MI, NI, WM, WN>), grid, dim3(NT), lds, s, a, b, ep, M, N, nprob, nsplit, gm, gn);
 }
 
+#ifdef IG_TRACE
+// measurement builds (make EXTRA=-DIG_TRACE; tools/ig_trace.sh): launches with nprob == IG_TRACE_NPROB (environment, default 25: the
+// filter gradients) record per-block cycle stamps; the table is printed at exit
+struct IgTraceRec { unsigned long long* dev; int M, N, nprob, nsplit, tm, tn, nblk; const char* la; };
+static std::vector<IgTraceRec>& ig_trace_recs() { static std::vector<IgTraceRec> v; return v; }
+static void ig_trace_dump() {
+    (void)hipDeviceSynchronize();
+    std::vector<unsigned long long> h(4096 * 8);
+    const size_t keep = ig_trace_recs().size() > 60 ? ig_trace_recs().size() - 60 : 0;      // the last step(s)
+    for (size_t i = keep; i < ig_trace_recs().size(); ++i) {
+        const IgTraceRec& r = ig_trace_recs()[i];
+        (void)hipMemcpy(h.data(), r.dev, h.size() * 8, hipMemcpyDeviceToHost);
+        const int nb = r.nblk < 4096 ? r.nblk : 4096;
+        double pro = 0, loop = 0, epi = 0, chunks = 0, t0min = 1e30, t3max = 0, rt0 = 1e30, rt1 = 0;
+        int n = 0;
+        for (int b = 0; b < nb; ++b) {
+            const unsigned long long* q = &h[(size_t)b * 8];
+            if (!q[3] || !q[1]) continue;
+            pro += (double)(q[1] - q[0]); loop += (double)(q[2] - q[1]); epi += (double)(q[3] - q[2]); chunks += (double)q[4]; ++n;
+            if ((double)q[7] < rt0) rt0 = (double)q[7];
+            if ((double)q[7] > rt1) rt1 = (double)q[7];
+        }
+        if (!n) continue;
+        // launch duration from the blocks' own clocks: start (100 MHz real-time counter) + length (shader cycles at ~2.4 GHz); per XCD the
+        // number of blocks, their summed length and when its last block ended
+        double xend[8] = {0}, xsum[8] = {0};
+        int xn[8] = {0};
+        double tend = 0, total = 0;
+        for (int b = 0; b < nb; ++b) {
+            const unsigned long long* q = &h[(size_t)b * 8];
+            if (!q[3] || !q[1]) continue;
+            const double len = (double)(q[3] - q[0]), e = ((double)q[7] - rt0) * 24.0 + len;
+            const int x = (int)(q[6] & 7);
+            xn[x]++; xsum[x] += len; total += len;
+            if (e > xend[x]) xend[x] = e;
+            if (e > tend) tend = e;
+        }
+        fprintf(stderr, "IG_TRACE %3s launch %3zu  M %5d N %4d nprob %2d nsplit %2d tile %dx%d blocks %5d | per block (cycles): prologue %7.0f  loop %8.0f (%5.1f chunks, %6.0f per chunk)  epilogue %6.0f | "
+                        "launch ~%.0f us, slot occupancy %.2f (512 slots) | per XCD blocks/end(us):", r.la, i, r.M, r.N, r.nprob, r.nsplit, r.tm, r.tn, r.nblk, pro / n, loop / n, chunks / n, loop / (chunks > 0 ? chunks : 1), epi / n,
+                tend / 2400.0, total / (tend * 512.0));
+        for (int x = 0; x < 8; ++x) fprintf(stderr, " %d/%.0f", xn[x], xend[x] / 2400.0);
+        fprintf(stderr, "\n");
+    }
+}
+#endif
+
 template <class LA, class LB, int MI, int NI, int WM, int WN>
 static void launch_tile_f32(hipStream_t s, const LA& a, const LB& b, Epi ep, int M, int N, int nprob, int nsplit) {
     constexpr int NT = 64 * WM * WN, TM = 32 * MI * WM, TN = 32 * NI * WN;
@@ -48,6 +95,20 @@ static void launch_tile_f32(hipStream_t s, const LA& a, const LB& b, Epi ep, int
     }
     if (ep.xcd_swizzle && !ep.swz_group) nblk = (nblk + 7) / 8 * 8;
     dim3 grid((unsigned)nblk);
+#ifdef IG_TRACE
+    {
+        static const int want = getenv("IG_TRACE_NPROB") ? atoi(getenv("IG_TRACE_NPROB")) : 25;
+        if (nprob == want && ig_trace_recs().size() < 4000) {
+            static bool reg = false;
+            if (!reg) { atexit(ig_trace_dump); reg = true; }
+            unsigned long long* d = nullptr;
+            (void)hipMalloc(&d, 4096 * 8 * 8);
+            (void)hipMemsetAsync(d, 0, 4096 * 8 * 8, s);
+            ep.trace = d;
+            ig_trace_recs().push_back(IgTraceRec{d, M, N, nprob, nsplit, TM, TN, (int)nblk, ""});
+        }
+    }
+#endif
     hipLaunchKernelGGL((igemm_kernel<LA, LB, MI, NI, WM, WN>), grid, dim3(NT), lds, s, a, b, ep, M, N, nprob, nsplit, gm, gn);
 }
 
diff --git a/imitation_from_observation_amd/csrc/igemm.h b/imitation_from_observation_amd/csrc/igemm.h
index 75bd0f7..45936af 100644
--- a/imitation_from_observation_amd/csrc/igemm.h
+++ b/imitation_from_observation_amd/csrc/igemm.h
@@ -119,6 +119,9 @@ struct Epi {
     int xcd_swizzle = 0;            // 1: consecutive work items go to the SAME XCD (its L2): block b does item (b % 8) * ceil(T/8) + b / 8
     int swz_group = 0;              // != 0 (multiple of 8): the swizzle is applied inside consecutive groups of this many items, so
                                     // all XCDs work on the same group (a parity class of a transposed conv) at the same time
+#ifdef IG_TRACE
+    unsigned long long* trace = nullptr;   // measurement builds only (tools/ig_trace.sh): s_memtime stamps per block [block][8]
+#endif
     const uint16_t* perm = nullptr; // != null: problem slot pr runs problem perm[pr] (device memory, nprob entries): the launcher's
                                     // load-balanced order for problems of unequal length (gemm_launch.h: balanced_order)
 };
@@ -1149,6 +1152,10 @@ __global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const LA la, const
     const int per = (nch + nsplit - 1) / nsplit;
     const int cb = split * per;
     const int ce = (cb + per < nch) ? cb + per : nch;
+#ifdef IG_TRACE
+    unsigned long long* trc = ep.trace && blockIdx.x < 4096 && threadIdx.x == 0 ? ep.trace + (size_t)blockIdx.x * 8 : nullptr;
+    if (trc) { trc[0] = __builtin_readcyclecounter(); trc[4] = (unsigned long long)(ce - cb); trc[5] = (unsigned long long)prob; unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); trc[6] = xcc & 15; trc[7] = __builtin_amdgcn_s_memrealtime(); }
+#endif
 
     Fetch<LA, TM, NT> fa;      // per-lane invariants of this thread's float4s
     Fetch<LB, TN, NT> fb;
@@ -1191,6 +1198,9 @@ __global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const LA la, const
             for (int p = 0; p < NB; ++p) xb[p] = fb.load1(lb, qb, p);
         }
         __syncthreads();
+#ifdef IG_TRACE
+        if (trc) trc[1] = __builtin_readcyclecounter();
+#endif
 
         // One chunk: multiply LDS stage `st` (16 MFMA groups (q,t) of MI*NI instructions).  In the gaps:
         //  TWO_SETS : groups 0..7 load chunk c+2 into (la_, lb_); groups 8..15 store (sa_, sb_) = chunk c+1
@@ -1251,6 +1261,9 @@ __global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const LA la, const
         }
     }
 
+#ifdef IG_TRACE
+    if (trc) trc[2] = __builtin_readcyclecounter();
+#endif
     // D layout (32x32 MFMA): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
     // position-major launches (rowmode 4 / 5): the destination pixel is LINEAR in the row (= image) index; the
     // block-uniform part is resolved once here, outside the unrolled loops
@@ -1269,6 +1282,9 @@ __global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const LA la, const
                 }
             }
         }
+#ifdef IG_TRACE
+        if (trc) trc[3] = __builtin_readcyclecounter();
+#endif
         return;
     }
     // Everything an element needs from memory (bias, skip-gradient adds, the saved activation behind lrelu') is requested for GR = 4 / NI
IG_TRACE_PATCH_EOF
make -C imitation_from_observation_amd/csrc -j32 EXTRA=-DIG_TRACE > /dev/null 2>&1 || { echo "build failed"; exit 1; }
for ov in 0 1; do
echo "=== CTX_OVERLAP=$ov" >> "$OUT"
CTX_OVERLAP=$ov python - >> "$OUT" 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from imitation_from_observation_amd import Translator
B = 256
g = torch.Generator(device="cuda").manual_seed(0)
fr = [torch.rand((B, 64, 64, 3), device="cuda", generator=g) * 2 - 1 for _ in range(3)]
with Translator(max_batch=B) as tr:
    tr.init_params(1)
    for _ in range(4):
        tr.dev_forward_backward(*(t.data_ptr() for t in fr), B)
        tr.dev_adam(1e-4)
    tr.sync()
PY
done
grep -c IG_TRACE "$OUT"
