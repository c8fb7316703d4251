// options.h -- the library's tuning switches, per handle (include/ctxtrans.h: ctx_set_option / ctx_get_option).
// A handle copies the process defaults at ctx_create / ctx_cnn_create; the defaults are the environment variables CTX_<NAME> read at
// that moment (unset = the built-in default).  The launchers read the CALLING handle's options through a thread-local pointer that the
// C ABI's entry points set (ctx::OptScope), so two handles in one process can run different settings.
#pragma once

namespace ctx {

enum Opt {
    OPT_OVERLAP,       // 1: the step runs on three stream lanes (conv_context chain, filter / bias gradients beside the dx chain); 0: one stream;
                       // -1 (default): by size at create -- off for the table-driven translators on maps under 64 positions; reads back 0 / 1
    OPT_GRAPHS,        // 1: the inference fetches at B <= 64 and the CNN front end replay captured hipGraphs
    OPT_GRAPH_LANES,   // 1: the captured inference forward keeps the stream lanes as graph branches (translate: both encoders side by side)
    OPT_POSMAJOR,      // 1: position-major convolutions (only the taps inside the grid) at >= 64 images
    OPT_XCD_SWIZZLE,   // bits: 1 position-major conv, 2 position-major transposed conv, 4 rectangle-ordered filter gradient: contiguous runs of work per XCD
    OPT_BALANCE,       // bits: position-major conv: 1 load-balanced problem order on <= 16-position grids, 2 on larger grids, 8 Z-order runs on larger
                       // grids instead (wins over 2); 4 load-balanced taps in the filter gradient
    OPT_WCONVT,        // bits: 1 LDS-resident transposed conv (wconvt.hip), 2 row blocks on 4x4 grids, 4 row blocks on 8x8 grids, 8 column-uniform waves,
                       // 16 inference launches of <= 32 images as one product + a gather (launch.h: convt_product)
    OPT_DIRECT3,       // bits: 1 3-channel layers on the direct kernels, 2 c3conv, 4 c3wgrad, 8 d_h4 forward in one pass (convt3), 16 ... on the matrix cores at >= 128 images (convt3m)
    OPT_DCONV,         // bits: 1 ContextAEReal in f32 on the narrow-channel direct kernels (dconv.h), 2 their forward-type launches on the K-sliced
                       // double-buffered LDS-DMA kernel (dconv2.h) where it applies
    OPT_RCHAIN,        // 1: ContextAEReal's FC middle in three launches (rchain.hip)
    OPT_EARLY_ADAM,    // 1: Adam's slices beside the remaining backward in the fused ContextSkipNew steps (bit-identical; -0.06 ms, round 4)
    OPT_CNN_LANES,     // Inception front end: -1 = by precision (lanes in split-bf16 mode only), 0 / 1 = off / on
    OPT_CNN_DCONV,     // Inception front end, f32: 1 = layers of <= 32 input and output channels (Conv2d_2a_3x3) on the direct kernels of dconv.h;
                       // 0 = implicit GEMM
    OPT_CNN_STEM4,     // 1: the front end's 3-channel first conv on the 4-channel gather
    OPT_TRACE_LAUNCH,  // 1: one stderr line per distinct implicit-GEMM launch shape (diagnostics)
    OPT_ADAM_PRIO,     // HIP priority of the early-Adam stream, read at create: 1 low (its own hardware queue), 0 normal, -1 high; 2 (default) = 1 for
                       // exact-f32 handles, 0 for split-bf16 ones; reads back resolved
    OPT_COUNT
};

struct Options { int v[OPT_COUNT]; };

const char* opt_name(int i);                 // lower-case name, e.g. "wconvt"; the environment variable is CTX_ + upper case
int opt_find(const char* name);              // index or -1 (case-insensitive, with or without the CTX_ prefix)
Options options_from_env();                  // built-in defaults overridden by the environment, read now
extern thread_local const Options* g_opt;    // the calling handle's options (null: options_from_env() of the first use)
int opt(Opt o);

struct OptScope {                            // entry points: `OptScope os(&h->opt);`
    const Options* prev;
    explicit OptScope(const Options* o) : prev(g_opt) { g_opt = o; }
    ~OptScope() { g_opt = prev; }
};

}  // namespace ctx
