/*
 * ctxtrans.h -- C ABI of libctxtrans.so: the context-translation encoder/decoder
 * ("translator") of imitation_from_observation as hand-written HIP kernels for gfx950.
 *
 * The reference has no FFI for this path; its boundary is the TensorFlow feed/fetch contract used
 * at four call sites plus Saver.restore/save.  Every entry point below replaces one of them
 * (paths relative to the reference root):
 *
 *   ctx_create            Model().build(placeholder)            rllab/sampler/base.py:134-138,
 *                                                               scripts/train_script.py:118-121
 *   ctx_param_* / set / get   tf.train.Saver var list, restore/save   base.py:144-145,
 *                                                               train_script.py:133,181
 *   ctx_init_params       tf.global_variables_initializer       train_script.py:129
 *   ctx_translate         sess.run([translated_z, out], {image:[src,[ctx]*B,[ctx]*B]})
 *                                                               base.py:216-218
 *   ctx_encode            sess.run([input_z, image_trans], ...) base.py:234-235
 *   ctx_train_step        sess.run([optimizer, loss, simloss, recon1, recon2], ...)
 *                                                               train_script.py:163,167
 *   ctx_eval              sess.run([loss, simloss, recon1, recon2, out, out2], ...)
 *                                                               train_script.py:176,192-193
 *   ctx_dev_*             the same train step split into device-resident phases so a host can put
 *                         an RCCL gradient all-reduce between backward and Adam (new: the
 *                         reference has no multi-GPU path)
 *   ctx_dp_*              that all-reduce itself, on RCCL, behind this ABI (no torch needed)
 *
 * Conventions: every function returns 0 on success or a negative CTX_E_* code; the message is
 * available from ctx_last_error().  No C++ exception crosses the ABI.  Host buffers are caller
 * owned and only touched during the call.  Layouts are the reference's: frames NHWC uint8 / f32,
 * parameters flat f32 in ctx_param_info order (TF variable names and shapes).  A handle is bound
 * to one device and is thread-compatible, not thread-safe (one caller at a time, like the single
 * tf.Session user).  There is NO CPU fallback: without a usable gfx950 device ctx_create fails.
 */
#ifndef CTXTRANS_H
#define CTXTRANS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTX_ABI_VERSION 4   /* 2: ctx_config carries strides / kernels / filters / keep_prob / loss_mode; 3: per-handle options,
                               ctx_dp_train_step_sampled / ctx_dp_eval_sampled, ctx_prof_entry.useful_frac; 4: ctx_dev_frames */

enum {
    CTX_OK = 0,
    CTX_E_INVALID = -1,   /* bad argument / unsupported configuration */
    CTX_E_DEVICE = -2,    /* HIP runtime error (no device, launch failure, ...) */
    CTX_E_NOMEM = -3,     /* device allocation failed */
    CTX_E_STATE = -4      /* call sequence error (e.g. adam before backward) */
};

enum {
    CTX_VARIANT_SKIPNEW = 0, /* ContextSkipNew, gym/envs/mujoco/arm_shaping.py:1260-1354 */
    CTX_VARIANT_INCEPTION2 = 2, /* ContextAEInception2(strides, kernels, filters), arm_shaping.py:1786-1894 (mode
                                   'oursinception').  ctx_config.strides / kernels / filters are the constructor's lists
                                   (:1787-1803: s1..s4, k1..k4, f1..f4; the decoder mirrors them); all-zero lists mean the
                                   sampler's instantiation strides [1,2,1,2], kernels [3,3,3,3], filters [16d,16d,8d,8d]
                                   (d = df_dim = 64: rllab/sampler/base.py:126).  Strides 1 | 2, kernels 1..5 (k x k),
                                   filters multiples of 32.  Inputs are Inception-v3 Mixed_7c FEATURE MAPS, f32
                                   [B, H, W, C] with C a multiple of 32 (2048; H = W = 2 for 125x125 frames, 8 for
                                   299x299); out = decode + tgtctx.  The uint8 entry points are refused: use *_f32. */
    CTX_VARIANT_REAL = 1     /* ContextAEReal, arm_shaping.py:1599-1684 (sampler names 'real', 'sweep'): shared
                                encoder, filters 32/16/16/8, strides 1/2/1/2; H, W multiples of 4, featsize (100)
                                a multiple of 4, df_dim ignored, keep_prob = 1 */
};

/* Arithmetic of the convolutions / linear layers (everything else -- epilogues, losses, reductions, Adam,
 * parameters, activations in HBM -- is f32 in both modes):
 *   CTX_PREC_F32     v_mfma_f32_32x32x2_f32: bitwise an fmaf chain.
 *   CTX_PREC_BF16X3  every f32 operand is split on the fly into bf16 hi + bf16 lo and a*b is evaluated as
 *                    hi*hi + hi*lo + lo*hi on the bf16 matrix cores with f32 accumulation: ~2^-16 relative error
 *                    per product, inside the 1e-3 budget of the path, at 16/3 the f32 matrix rate. */
enum { CTX_PREC_F32 = 0, CTX_PREC_BF16X3 = 1 };

typedef struct ctx_config {
    int32_t variant;    /* CTX_VARIANT_* */
    int32_t H, W, C;    /* frame size; H, W multiples of 16 (arm_shaping.py:1314-1319); C == 3 */
    int32_t df_dim;     /* encoder/decoder base width (df_dim == gf_dim == 64 in the reference);
                           multiple of 32 */
    int32_t featsize;   /* 1024 in the reference (arm_shaping.py:1277); multiple of 32 */
    int32_t max_batch;  /* largest B any later call will pass */
    int32_t precision;  /* CTX_PREC_*: arithmetic of the matrix contractions (0 = exact f32, the default) */
    /* ---- ABI 2 (zero-initialise for the defaults) ---- */
    int32_t strides[4]; /* CTX_VARIANT_INCEPTION2: s1..s4 of the constructor (0,0,0,0 = 1,2,1,2) */
    int32_t kernels[4]; /*                          k1..k4 (0 = 3) */
    int32_t filters[4]; /*                          f1..f4 (0 = 16d,16d,8d,8d) */
    float keep_prob;    /* CTX_VARIANT_REAL: tf.nn.dropout keep probability of the TRAINING graph (arm_shaping.py:1637-1661; the
                           module-level default is 1.0, :1476; ablations_code/ablations.py:544 feeds 0.5).  0 or 1 = no dropout.
                           Inference fetches (ctx_translate / ctx_encode) and ctx_eval never drop (keep_prob = 1 is what the
                           sampler's graph has). */
    int32_t loss_terms; /* CTX_LOSS_* bits: which terms make up `loss`, i.e. what Adam minimises (0 = all three) */
} ctx_config;

/* `loss` of the training graph.  The reference's trainer minimises recon1 + recon2 + simloss (arm_shaping.py:1354); its ablation
 * script switches terms off (ablations_code/ablations.py:175-182, 278-285):  "None" = all (7),  "L2" = recon1 + recon2 (3),
 * "L2L3" = recon1 (1),  "L1" = recon2 + simloss (6).  All four scalars are reported whatever the mask. */
enum { CTX_LOSS_RECON1 = 1, CTX_LOSS_RECON2 = 2, CTX_LOSS_SIM = 4 };

typedef struct ctx_handle ctx_handle;

/* ---- lifetime ------------------------------------------------------------------------------ */
int ctx_abi_version(void);
/* Allocates everything on `device` with hipMalloc and creates a private stream. */
int ctx_create(const ctx_config* cfg, int device, ctx_handle** out);
/* Same, on caller-owned resources: `stream` is a hipStream_t (NULL = private stream), `arena` is
 * device memory of ctx_arena_bytes(cfg) bytes laid out [params | grads | adam_m | adam_v], each
 * ctx_param_total floats (NULL = allocate).  Lets a host framework own the gradient buffer it
 * hands to its collective library. */
int ctx_create_ex(const ctx_config* cfg, int device, void* stream, void* arena, ctx_handle** out);
void ctx_destroy(ctx_handle* h);
/* Message of the last failed call on `h` (or of the last failed ctx_create when h == NULL). */
const char* ctx_last_error(const ctx_handle* h);

/* ---- tuning switches, per handle ---------------------------------------------------------------
 * Every switch is an int.  A handle starts from the built-in defaults overridden by the environment variables CTX_<NAME> (upper
 * case) as they stand at ctx_create; ctx_set_option changes it for that handle only (two handles of one process may differ).
 * Names (ctx_option_count / ctx_option_name enumerate them):
 *   overlap     -1   1 = the step runs on three stream lanes (conv_context chain; filter / bias gradients beside the dx chain); 0 = one stream;
 *                    -1 = decided at create (off for the table-driven translators on maps under 64 positions) and reads back as 0 / 1
 *   graphs       1   the inference fetches at B <= 64 replay captured hipGraphs
 *   graph_lanes  1   ... with the stream lanes captured as graph branches (translate: the two encoders run side by side)
 *   posmajor     1   position-major convolutions (only the taps inside the grid) from 64 images up
 *   xcd_swizzle  7   bits: contiguous runs of work per XCD for 1 the position-major conv, 2 the transposed conv, 4 the filter gradient
 *   balance      9   bits: problem order of the position-major conv: 1 load-balanced runs on grids of <= 16 positions, 2 on larger grids,
 *                    8 Z-order (2-D compact) runs on larger grids instead (wins over 2); 4 load-balanced taps in the filter gradient
 *   wconvt      31   bits: 1 LDS-resident transposed conv, 2 / 4 row blocks on 4x4 / 8x8 grids, 8 column-uniform waves (4x4),
 *                    16 inference launches of <= 32 images as one product + a gather
 *   direct3     31   bits: 1 3-channel layers on the direct kernels, 2 c3conv, 4 c3wgrad, 8 d_h4 forward in one pass, 16 d_h4 forward on the
 *                    matrix cores at >= 128 images (convt3m.hip)  [fixed at create]
 *   dconv        3   bits: 1 ContextAEReal in f32 on the narrow-channel direct kernels, 2 the K-sliced LDS-DMA forward kernel     [fixed at create]
 *   rchain       1   ContextAEReal's FC middle in three launches
 *   early_adam   1   Adam's slices beside the remaining backward in the fused ContextSkipNew steps (bit-identical; -0.06 ms)
 *   cnn_lanes   -1   Inception front end: branch lanes; -1 = in the split-bf16 mode only      (ctx_cnn handles: environment at create) [fixed at create]
 *   cnn_dconv    1   Inception front end (f32): layers of <= 32 input and output channels on the direct kernels (dconv.h); 0 = implicit GEMM  [fixed at create]
 *   cnn_stem4    1   Inception front end: the 3-channel first conv on the 4-channel gather    (ctx_cnn handles: environment at create) [fixed at create]
 *   trace_launch 0   one stderr line per distinct implicit-GEMM launch shape
 *   adam_prio    2   HIP priority of the early-Adam stream (1 low: its own hardware queue; 0 normal; -1 high; 2 = low for exact-f32 handles,
 *                    normal for split-bf16 ones, reads back resolved)  [fixed at create]
 * Results never depend on a switch beyond f32 summation order -- tested value by value (tests/test_gpu_options.py, against the default
 * switches on the bench's launch shapes; the defaults themselves are what every oracle suite runs):
 *   ContextSkipNew 64x64 B = 256:  overlap 0 | posmajor 0 | xcd_swizzle 0 1 2 3 4 5 6 | balance 0 1 2 3 4 5 8 13 | wconvt 0 1 3 5 7 15 23 29 |
 *                                  direct3 0 1 3 5 7 9 15 23 | early_adam 0 | adam_prio -1 0 1;  graph_lanes 0 1 and early_adam 0 1 also in
 *                                  tests/test_gpu_parity.py against the oracle
 *   ContextAEReal 36x64 B = 64:    overlap 0 1 | dconv 0 1 5 7 | rchain 0 | direct3 0 15 | posmajor 0
 * Combinations of two non-default switches are not enumerated; cnn_* (front end) and graphs 0 run in their own suites' defaults only.
 * Not options: CTX_DEBUG_POISON=1 (debugging aid: every device
 * buffer a handle allocates is filled with 0xFF bytes -- float NaN -- so that a read of never-written memory shows on every run),
 * CTX_RCCL_LIB (path of the librccl to dlopen, read by the
 * first ctx_dp_* call of the process). */
int ctx_option_count(void);
const char* ctx_option_name(int index);                               /* NULL past the end */
int ctx_get_option(const ctx_handle* h, const char* name, int* value);
int ctx_set_option(ctx_handle* h, const char* name, int value);       /* CTX_E_INVALID: unknown name; CTX_E_STATE: fixed at create */

/* ---- parameter inventory (TF variable names) ------------------------------------------------ */
int64_t ctx_param_total_for(const ctx_config* cfg); /* number of f32 parameters, <0 on error */
int64_t ctx_arena_bytes(const ctx_config* cfg);     /* 4 * 4 * ctx_param_total_for */
int64_t ctx_param_total(const ctx_handle* h);
int ctx_param_count(const ctx_handle* h);           /* number of tensors */
/* name: e.g. "conv/h0_conv/w"; shape: up to 4 dims, unused = 1; offset: in floats into the arena */
int ctx_param_info(const ctx_handle* h, int index, const char** name, int* ndim, int64_t shape[4],
                   int64_t* offset);
int ctx_set_params(ctx_handle* h, const float* flat, size_t n);
int ctx_get_params(ctx_handle* h, float* flat, size_t n);
int ctx_get_grads(ctx_handle* h, float* flat, size_t n); /* gradient of the last backward */
/* Adam slots + step counter (Saver saves them too, train_script.py:133). */
int ctx_set_adam_state(ctx_handle* h, const float* m, const float* v, size_t n, int64_t step);
int ctx_get_adam_state(ctx_handle* h, float* m, float* v, size_t n, int64_t* step);
/* conv w: truncated normal(0.02); deconv w, FC Matrix: normal(0.02); biases 0
 * (arm_shaping.py:25-29, 52-55, 67-68, 79).  Also zeroes the Adam state. */
int ctx_init_params(ctx_handle* h, uint64_t seed);

/* ---- inference: the rllab reward hook's two fetches ------------------------------------------ */
/* translate(obs_src, obs_tgt0) -> (pred_frame, feat).
 * src  [B,H,W,3] uint8; ctx0 [H,W,3] uint8 (ctx_batched == 0, broadcast to B like base.py's
 * [context]*batch_size) or [B,H,W,3] (ctx_batched != 0).
 * pred [B,H,W,3] f32 = model.out; feat [B,featsize] f32 = model.translated_z.  Either may be NULL. */
int ctx_translate(ctx_handle* h, const uint8_t* src, const uint8_t* ctx0, int ctx_batched, int B,
                  float* pred, float* feat);
/* frames [B,H,W,3] uint8 -> feat [B,featsize] = model.input_z; frames_f32 (nullable) [B,H,W,3] =
 * image_trans[0] = (x/255 - 0.5)*2 -- the device's bits either way: up to 2^20 elements (the hook's batch of 25 and a few paths) they are written
 * by the host from the 256-entry table of the same three f32 operations while the device encodes (no download), beyond that copied back. */
int ctx_encode(ctx_handle* h, const uint8_t* frames, int B, float* feat, float* frames_f32);
/* The same two fetches on float inputs [B,H,W,C]: frames already scaled to [-1,1], or -- for
 * CTX_VARIANT_INCEPTION2, where image_trans IS the feature tensor (base.py:127-132) -- Mixed_7c feature maps. */
int ctx_translate_f32(ctx_handle* h, const float* src, const float* ctx0, int ctx_batched, int B,
                      float* pred, float* feat);
int ctx_encode_f32(ctx_handle* h, const float* frames, int B, float* feat);
/* ... and with the inputs already in DEVICE memory (d_*: f32 [B,H,W,C]; d_ctx0 [H,W,C] or, ctx_batched != 0, [B,H,W,C]), e.g. the
 * output buffer of ctx_cnn_forward_u8_dev on the same stream: mode 'oursinception' without a host round trip of the feature maps
 * (base.py:121-132, 216-218, 234-235).  pred / feat are HOST buffers (nullable). */
int ctx_translate_dev(ctx_handle* h, const float* d_src, const float* d_ctx0, int ctx_batched, int B, float* pred, float* feat);
int ctx_encode_dev(ctx_handle* h, const float* d_frames, int B, float* feat);

/* The per-path cost of the reward hook computed where the frames already are (base.py:232-249), several paths per call:
 * ctx_reward_set_cache keeps, per viewpoint vp, the demo cache  means [bs, featsize] (= self.means[vp]) and  imgs [bs,H,W,3]
 * (= self.imgs[vp])  on the device;  ctx_reward_costs encodes  frames [npaths*bs,H,W,3] uint8 (npaths rollouts of bs rendered
 * frames) and returns  costs[p*bs + j] = sum((means[j] - input_z[p,j])^2) + scale * sum((imgs[j] - image_trans[0][p,j])^2)
 * (ablation 0 = "None"; 1 = "nofeat": image term only; 2 = "noimage": feature term only).  Only npaths*bs floats come back
 * over PCIe instead of the preprocessed frames (49 MB at 40 paths). */
int ctx_reward_set_cache(ctx_handle* h, int vp, const float* means, const float* imgs, int bs);
int ctx_reward_costs(ctx_handle* h, int vp, const uint8_t* frames, int npaths, float scale, int ablation, float* costs);

/* ---- training --------------------------------------------------------------------------------- */
/* src/ctx/tgt [B,H,W,3] f32 in [-1,1] (tfinput[0], [1], [2]).  scalars = {loss, simloss, recon1,
 * recon2} of the forward pass before the update.  Adam: TF defaults b1 .9, b2 .999, eps 1e-8. */
int ctx_train_step(ctx_handle* h, const float* src, const float* ctx, const float* tgt, int B,
                   float lr, float scalars[4]);
/* CTX_VARIANT_REAL with 0 < keep_prob < 1: seed of the dropout masks of the training entry points (ctx_train_step*,
 * ctx_dev_forward_backward, ctx_dp_train_step).  The factor of element e at dropout site s in the step that follows `t` Adam updates
 * is  (hash32(seed, t, s, e) < keep_prob * 2^32) / keep_prob  -- a counter-based hash (csrc/kernels.hip: drop_hash) that
 * oracle/ctx_oracle_real.py restates, so a step can be checked with the masks it used.  Default seed 0.  (TensorFlow's own random
 * stream is not reproducible from outside; tf.nn.dropout's arithmetic x * mask / keep_prob is.)  After ctx_dp_init rank r hashes with
 * seed ^ (0x9E3779B9 * r), so the shards of a data-parallel batch draw different masks (rank 0 keeps the single-device masks). */
int ctx_set_dropout_seed(ctx_handle* h, uint64_t seed);
/* Same on uint8 frames, preprocessed on device with (x/255 - 0.5)*2. */
int ctx_train_step_u8(ctx_handle* h, const uint8_t* src, const uint8_t* ctx, const uint8_t* tgt,
                      int B, float lr, float scalars[4]);
/* The trainer's input pipeline on device (scripts/train_script.py:144-159).  ctx_demos_upload keeps the demo
 * tensor vdata[T][N][H][W][3] (uint8 frames, T frames of N videos) resident in HBM; ctx_train_step_sampled
 * builds the batch  src[b] = vdata[b % T][choicesrc[b]], tgt[b] = vdata[b % T][choicetgt[b]],
 * ctx[b] = vdata[0][choicetgt[b]]  with the trainer's x/127.5 - 1 scaling and runs one train step.  The two
 * index arrays are what `np.random.choice(ntrain, batch_size)` returns (:154-155). */
int ctx_demos_upload(ctx_handle* h, const uint8_t* vdata, int T, int N);
int ctx_train_step_sampled(ctx_handle* h, const int32_t* choicesrc, const int32_t* choicetgt, int B,
                           float lr, float scalars[4]);
/* The validation batch of the trainer (train_script.py:169-176) from the resident demo tensor: the same gather as
 * ctx_train_step_sampled, forward + losses, no update.  out / out2 (nullable) [B,H,W,3]. */
int ctx_eval_sampled(ctx_handle* h, const int32_t* choicesrc, const int32_t* choicetgt, int B, float scalars[4],
                     float* out, float* out2);
/* Host copies of model.out / model.out2 [B,H,W,3] of the last training-mode forward and of the tgt frames it was fed
 * (tfinput[2]) -- what the trainer's `nn_err` fetch reads next to the optimizer (train_script.py:148,163).  Any may be NULL. */
int ctx_last_outputs(ctx_handle* h, float* out, float* out2, float* tgt);
/* Forward + losses only.  out / out2 (nullable) [B,H,W,3]. */
int ctx_eval(ctx_handle* h, const float* src, const float* ctx, const float* tgt, int B,
             float scalars[4], float* out, float* out2);

/* ---- device-resident phases (benchmarks, data parallel) -------------------------------------- */
/* d_* are DEVICE pointers [B,H,W,3] f32.  Enqueues forward + backward on the handle's stream and
 * returns without synchronising.  sim_batch: batch in the simloss mean's denominator (0 = B); a
 * data-parallel shard passes the GLOBAL batch so a SUM all-reduce of the gradient arena equals the
 * full-batch gradient. */
int ctx_dev_forward_backward(ctx_handle* h, const float* d_src, const float* d_ctx,
                             const float* d_tgt, int B, int sim_batch);
/* The handle's own frame buffer for a batch of B: the device entry points (ctx_dev_*, ctx_dp_train_step) copy the caller's three
 * tensors into it (3 B frames device-to-device per step) -- unless a pointer passed to them IS the one returned here, i.e. the caller
 * (a device-side sampler, a front end) wrote that slot in place.  The pointers depend on B (slots are packed [tgt | src | ctx]). */
int ctx_dev_frames(ctx_handle* h, int B, float** d_src, float** d_ctx, float** d_tgt);
int ctx_dev_forward(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt,
                    int B);
/* One whole training step on device-resident frames: forward + backward + Adam (train_script.py:163's
 * sess.run([..., optim]) without the host copies), enqueued on the handle's stream, no synchronisation.
 * Same result, bit for bit, as ctx_dev_forward_backward(sim_batch = 0) followed by ctx_dev_adam(lr).  With
 * option "early_adam" set, Adam's update of a parameter slice is enqueued beside the remaining
 * backward as soon as that slice's gradients are final and its parameters are no longer read (on by default:
 * -0.06 ms of 13.1 on MI355X; bit-identical either way).  The host-fed steps (ctx_train_step, _u8, _sampled)
 * go through the same code. */
int ctx_dev_train_step(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B,
                       float lr);
/* Fused multi-tensor Adam over the whole arena with the gradients currently in the grad arena. */
/* Data-parallel overlap (new; the reference is single-device).  When set, ctx_dev_forward_backward calls fn(user, 0, first,
 * count) on the calling thread as soon as gradients [first, first + count) of the gradient arena -- translate/ and deconv/,
 * the tail of the arena -- are complete in the handle's stream order; the caller starts their all-reduce there (ordered
 * after the handle's stream) while the encoders' backward is still being enqueued, and reduces [0, first) after the call
 * returns.  fn == NULL clears it. */
typedef void (*ctx_bucket_fn)(void* user, int bucket, int64_t first, int64_t count);
int ctx_set_grad_bucket_callback(ctx_handle* h, ctx_bucket_fn fn, void* user);
int ctx_dev_adam(ctx_handle* h, float lr);
/* Synchronises and copies {loss, simloss, recon1, recon2} of the last forward. */
int ctx_dev_scalars(ctx_handle* h, float scalars[4]);
/* Device pointer to the parameters (flat, ctx_param_info order).  WRITABLE: a caller may update the weights through it (its own
 * optimiser, a torch-side broadcast) on ctx_stream(h) or after ctx_sync(h).  The library cannot see such writes, so from the first call
 * on it stops trusting filters it packed earlier for the direct kernels (ContextAEReal / the front end): they are re-packed in front of
 * every launch, and inference graphs captured before the call are dropped.  Handles that never call it keep the packed-filter cache. */
void* ctx_dev_params(ctx_handle* h);
void* ctx_dev_grads(ctx_handle* h);
void* ctx_dev_scalar_buf(ctx_handle* h); /* device f32[4] written by the last forward */
void* ctx_stream(ctx_handle* h);         /* hipStream_t the kernels are enqueued on */
int ctx_sync(ctx_handle* h);
/* Device outputs of the last forward (valid until the next call): model.out / out2 [B,H,W,3],
 * input_z / translated_z [B,featsize]. */
int ctx_dev_outputs(ctx_handle* h, const float** out, const float** out2, const float** input_z,
                    const float** translated_z);
/* Host copies of model.input_z / model.translated_z [B, featsize] (arm_shaping.py:1298, :1312) of the last training-mode
 * forward (ctx_eval / ctx_train_step* / ctx_dev_forward*), i.e. sess.run([..., input_z, translated_z], feed) next to the
 * losses; rows are de-padded (the device keeps them at a stride of featsize rounded up to 32 for CTX_VARIANT_REAL).
 * Either pointer may be NULL; *B (nullable) receives the batch of that forward. */
int ctx_last_codes(ctx_handle* h, float* input_z, float* translated_z, int* B);

/* ---- data parallel over RCCL (new: the reference is single-device) ------------------------------
 * One process per GPU, each with a full replica + Adam state; per step: local forward/backward on the rank's shard with
 * the simloss mean taken over the GLOBAL batch -> SUM all-reduce of the flat f32 gradient arena over xGMI -> identical
 * local Adam.  librccl is loaded at run time on the first ctx_dp_* call (CTX_RCCL_LIB overrides the search; a process
 * that already holds a librccl.so.1 -- PyTorch's -- shares it).
 *   ctx_dp_unique_id      rank 0 makes the rendezvous blob (an ncclUniqueId); the host ships it to the other ranks
 *   ctx_dp_init           collective: creates the communicator on the handle's device, then broadcasts rank 0's parameters
 *                         and Adam slots so the replicas start identical
 *   ctx_dp_allreduce_grads  the exchange step alone: in-place SUM all-reduce of the gradient arena, stream-ordered between
 *                         ctx_dev_forward_backward(sim_batch = B * world) and ctx_dev_adam (asynchronous)
 *   ctx_dp_train_step     the whole step with the bucketed schedule: the translate/deconv gradients, then each encoder's
 *                         h4_lin / hz_lin slice, are reduced on a second stream while the encoders' backward still runs;
 *                         only the encoders' conv filters (18 % of the arena) go after it; then Adam.  scalars (nullable) =
 *                         GLOBAL {loss, simloss, recon1, recon2} (one more 16-byte all-reduce and a sync).  d_* are DEVICE
 *                         pointers [B,H,W,3] f32; every rank passes the same B.
 *   ctx_dp_scalars        global scalars of the last forward (collective) */
#define CTX_DP_UNIQUE_ID_BYTES 128
int ctx_dp_unique_id(uint8_t id[CTX_DP_UNIQUE_ID_BYTES]);
int ctx_dp_init(ctx_handle* h, const uint8_t id[CTX_DP_UNIQUE_ID_BYTES], int rank, int world);
int ctx_dp_world(const ctx_handle* h, int* rank, int* world);   /* (rank 0, world 0) before ctx_dp_init */
int ctx_dp_allreduce_grads(ctx_handle* h);
int ctx_dp_train_step(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B, float lr,
                      float scalars[4]);
int ctx_dp_scalars(ctx_handle* h, float scalars[4]);
/* The trainer's loop on N GPUs without a host gather (scripts/train_script.py:153-167 + SURVEY.md 8e / 8f-3): every rank holds the
 * whole demo tensor in HBM (ctx_demos_upload) and is handed the SAME global index arrays choicesrc / choicetgt [B_global]
 * (np.random.choice(ntrain, batch_size) twice, :154-155; B_global a multiple of the world size, B_global / world <= max_batch).
 * Rank r gathers rows b in [r B/world, (r+1) B/world) of the global batch on the device -- src[b] = vdata[b % T][choicesrc[b]],
 * tgt[b] = vdata[b % T][choicetgt[b]], ctx[b] = vdata[0][choicetgt[b]], b the GLOBAL row -- and runs ctx_dp_train_step's bucketed
 * schedule on them: N ranks leave the parameters a single handle's ctx_train_step_sampled leaves on the same arrays (up to f32
 * summation order).  scalars (nullable): the GLOBAL {loss, simloss, recon1, recon2}.
 * ctx_dp_eval_sampled: the validation fetch (:169-176) sharded the same way -- forward + losses on this rank's rows, GLOBAL scalars
 * (collective: every rank must call it), out / out2 (nullable) = THIS RANK's rows [B_global / world, H, W, 3]. */
int ctx_dp_train_step_sampled(ctx_handle* h, const int32_t* choicesrc, const int32_t* choicetgt, int B_global, float lr,
                              float scalars[4]);
int ctx_dp_eval_sampled(ctx_handle* h, const int32_t* choicesrc, const int32_t* choicetgt, int B_global, float scalars[4], float* out,
                        float* out2);
/* In-place SUM over the ranks of a HOST buffer of doubles (synchronous; through a device staging buffer and ncclAllReduce
 * on the handle's collective stream).  For the sharded demo cache of the reward hook (sampler/base.py:195-223 builds it on
 * one device; reward.py shards the demo videos rank::world and adds the partial feature / frame sums): the group that
 * ctx_dp_init made serves it, no second communication library in the sampler process. */
int ctx_dp_allreduce_host_f64(ctx_handle* h, double* buf, size_t n);

/* ---- measurement ------------------------------------------------------------------------------ */
/* One entry per launch group of a train step (a layer's forward, input gradient, filter gradient,
 * bias gradient, the losses, Adam): HIP-event time on the handle's stream, averaged over `iters`
 * full steps, plus the group's algorithmic FLOPs (2 per multiply-add, all 25 taps), the share of them that
 * is not a product with SAME padding (useful_frac) and the kernel that executes it.  bench.py derives its
 * `roofline` block from this. */
typedef struct ctx_prof_entry {
    char name[56];
    char kernel[40];
    double flops;       /* 2 per multiply-add, every tap of a SAME-padded layer counted (SURVEY.md 8d's convention) */
    float ms;
    float useful_frac;  /* share of `flops` whose product meets data on both sides: (valid (position, tap) pairs) / (all) of the
                           layer -- (5n-3)^2 / (5n)^2 for a 5x5 stride-2 layer with an n x n small grid; 1 for linear layers.
                           flops * useful_frac is what a kernel that never multiplies padding zeros has to do. */
} ctx_prof_entry;
int ctx_profile_step(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B,
                     float lr, int iters, ctx_prof_entry* entries, int max_entries, int* n_entries);

/* ---- test hook ------------------------------------------------------------------------------- */
/* Copies n floats of a named internal activation / gradient buffer to the host (bring-up and
 * parity tests only; names are listed in csrc/ctx_abi.cpp: ctx_debug_read). */
int ctx_debug_read(ctx_handle* h, const char* name, float* host, size_t n);

/* ---- frozen conv-net front end (mode 'oursinception') --------------------------------------------
 * Replaces the reference's  inception_v3.inception_v3(images, is_training=False)[1]['Mixed_7c']
 * (rllab/sampler/base.py:122-127, scripts/train_script.py:104-111; nets/inception_v3.py:93-416).  The graph is
 * handed over as an op list by the host (imitation_from_observation_amd/inception_frontend.py builds it from the
 * reference's layer table); this library executes it.  Activations are NHWC, channel counts rounded up to 32.
 *   CONV     slim.conv2d: kh x kw (kh*kw <= 25) conv, stride 1|2, TF 'SAME' or 'VALID', with the batch norm folded by the
 *            host into the filter  w' = w / sqrt(var + 0.001)  and a bias  b' = beta - mean / sqrt(var + 0.001),  then ReLU;
 *            filter [kh][kw][src channels (padded)][cout] at w_off, bias [cout] at b_off (floats into the blob);
 *            output written to channels [dst_ch0, dst_ch0 + cout) of buffer dst (tf.concat = adjacent slices).
 *   MAXPOOL  3x3 stride 2 VALID.     AVGPOOL  3x3 stride 1 SAME, mean over the taps inside the image.
 * Buffer 0 is the frame buffer (h, w, 32: channels 0..2 hold the frame); the LAST buffer is the output.
 * Every buffer is written once per pass (by one op, or by several ops into disjoint channel slices). */
enum { CTX_CNN_CONV = 0, CTX_CNN_MAXPOOL = 1, CTX_CNN_AVGPOOL = 2 };
typedef struct ctx_cnn_buf { int32_t h, w, c; } ctx_cnn_buf;
typedef struct ctx_cnn_op {
    int32_t kind, src, dst, dst_ch0;
    int32_t kh, kw, stride, same;     /* CONV only; same: 1 = 'SAME', 0 = 'VALID' */
    int32_t cout;
    int32_t lane;                     /* 0..3: ops of different lanes may run concurrently (the branches of an Inception block);
                                         the library orders every op after the writers of its src buffer */
    int64_t w_off, b_off;
    /* ---- ABI 2 (zero-initialise for the plain forms) ----
     * src_c != 0: the conv reads channels [src_ch0, src_ch0 + src_c) of buffer src (multiples of 32) instead of all of them: the
     *   filter is then [kh][kw][src_c][cout].
     * nsplit != 0 (CONV): a MERGED conv of sibling branches that read the same tensor (the 1x1 heads of an Inception block,
     *   nets/inception_v3.py:140-213, 236-364, 389-416) -- one GEMM with the filters side by side along cout: output columns
     *   [0, nsplit) go to (dst, dst_ch0), columns [nsplit, cout) to (dst2, dst2_ch0). */
    int32_t src_ch0, src_c;
    int32_t nsplit, dst2, dst2_ch0, reserved;
} ctx_cnn_op;
typedef struct ctx_cnn ctx_cnn;
int ctx_cnn_create(const ctx_cnn_buf* bufs, int nbufs, const ctx_cnn_op* ops, int nops, int64_t weight_floats,
                   int max_images, int precision, int device, void* stream, ctx_cnn** out);
void ctx_cnn_destroy(ctx_cnn* h);
const char* ctx_cnn_last_error(const ctx_cnn* h);          /* h == NULL: last creation error of this thread */
int ctx_cnn_set_weights(ctx_cnn* h, const float* blob, size_t n);
/* frames: host uint8 [n,H,W,3], preprocessed like base.py:116-119; out: host f32 [n,h,w,c] of the last buffer.
 * n may exceed max_images (processed in chunks). */
int ctx_cnn_forward_u8(ctx_cnn* h, const uint8_t* frames, int n, float* out);
/* frames: host uint8 [n,H,W,3], n <= max_images; *d_out: DEVICE pointer of the last buffer [n,h,w,c] (valid until the next
 * forward).  Asynchronous on the handle's stream: `frames` must stay untouched until the stream has passed the upload. */
int ctx_cnn_forward_u8_dev(ctx_cnn* h, const uint8_t* frames, int n, const float** d_out);
/* d_frames: DEVICE f32 [n,H,W,3] in [-1,1], n <= max_images; *d_out: device pointer of the last buffer.
 * Asynchronous on the handle's stream (ctx_cnn_stream / ctx_cnn_sync). */
int ctx_cnn_forward_dev(ctx_cnn* h, const float* d_frames, int n, const float** d_out);
int ctx_cnn_read_buffer(ctx_cnn* h, int index, int n, float* out);   /* end-point tests */
/* Per-op HIP-event times (ms, averaged over `iters` passes over the n images currently in buffer 0); measurement only. */
int ctx_cnn_profile(ctx_cnn* h, int n, int iters, float* ms, int max_ops);
void* ctx_cnn_stream(ctx_cnn* h);
int ctx_cnn_sync(ctx_cnn* h);

#ifdef __cplusplus
}
#endif
#endif /* CTXTRANS_H */
