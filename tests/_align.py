"""lrelu' branch alignment between a device pass and the float64 oracle.

lrelu is not differentiable at 0: an activation that the device computes within its rounding error of zero can land
on the other side than in the float64 oracle, and the two (equally valid) subgradients differ by 0.8 * dy at that
element.  In exact f32 that is 1-4 elements per few million; with split-bf16 products (error ~1e-5) a few hundred.
To compare gradients like with like, the oracle's saved activations are given the device's sign at exactly those
elements before its backward pass runs.  Returns the number of aligned elements."""
import numpy as np


def align_skipnew_cache(tr, c, B):
    """tr: Translator that has just run a forward on the same inputs; c: oracle cache from ctx_oracle.forward."""
    cache_of = {}
    for k in range(5):
        cache_of[f"s{k}"] = [(c["e_tgt"][k], slice(0, B)), (c["e_src"][k], slice(B, 2 * B))]
        cache_of[f"c{k}"] = [(c["e_ctx"][k], slice(0, B))]
    cache_of["th0"] = [(c["trans_h0"], slice(0, B))]
    cache_of["dz"] = [(c["d1"][0], slice(0, B)), (c["d2"][0], slice(B, 2 * B))]
    for k in range(1, 4):
        cache_of[f"e{k}"] = [(c["d1"][k], slice(0, B)), (c["d2"][k], slice(B, 2 * B))]
    cache_of["Z"] = [(c["e_tgt"][5], slice(B, 2 * B)), (c["e_src"][5], slice(2 * B, 3 * B))]
    cache_of["cz"] = [(c["e_ctx"][5], slice(0, B))]
    nflip, worst = 0, 0.0
    for name, parts in cache_of.items():
        rows = max(sl.stop for _, sl in parts)
        per_row = int(np.prod(parts[0][0].shape[1:]))
        got = tr.debug_read(name, rows * per_row).reshape((rows,) + parts[0][0].shape[1:])
        for arr, sl in parts:
            m = (got[sl] >= 0) != (arr >= 0)
            if m.any():
                worst = max(worst, float(np.abs(arr[m]).max() / np.abs(arr).max()))
                arr[m] = np.where(got[sl][m] >= 0, 1e-300, -1e-300)
                nflip += int(m.sum())
    return nflip, worst


def align_gen_cache(tr, c, B):
    """The same for the table-driven models (ContextAEInception2 / ContextAEReal without channel padding): device buffers a0..a3 hold
    the conv outputs of the stacked [tgt | src | ctx] images, a4 = h4, Z = [trans_z | tgt_z | src_z | ctx_z], dz / e1..e3 both decoder
    passes (ctx_debug_read's names for these variants).  c: cache of oracle/ctx_oracle_incep.forward (or ctx_oracle_real.forward).
    The code-wide buffers (a4, th0, Z) are kept at a row stride of featsize rounded up to 32 (ContextAEReal: 100 -> 128)."""
    cache_of = {}
    for k in range(5):
        cache_of[f"a{k}"] = [(c["e_tgt"][k], slice(0, B)), (c["e_src"][k], slice(B, 2 * B)), (c["e_ctx"][k], slice(2 * B, 3 * B))]
    cache_of["th0"] = [(c["trans_h0"], slice(0, B))]
    cache_of["dz"] = [(c["d1"][0], slice(0, B)), (c["d2"][0], slice(B, 2 * B))]
    for k in range(1, 4):
        cache_of[f"e{k}"] = [(c["d1"][k], slice(0, B)), (c["d2"][k], slice(B, 2 * B))]
    cache_of["Z"] = [(c["e_tgt"][5], slice(B, 2 * B)), (c["e_src"][5], slice(2 * B, 3 * B)), (c["e_ctx"][5], slice(3 * B, 4 * B))]
    nflip, worst, where = 0, 0.0, {}
    for name, parts in cache_of.items():
        rows = max(sl.stop for _, sl in parts)
        per_row = int(np.prod(parts[0][0].shape[1:]))
        if name in ("a4", "th0", "Z"):
            stride = -(-per_row // 32) * 32
            got = tr.debug_read(name, rows * stride).reshape(rows, stride)[:, :per_row]
        else:
            got = tr.debug_read(name, rows * per_row).reshape((rows,) + parts[0][0].shape[1:])
        for arr, sl in parts:
            m = (got[sl] >= 0) != (arr >= 0)
            if m.any():
                worst = max(worst, float(np.abs(arr[m]).max() / np.abs(arr).max()))
                arr[m] = np.where(got[sl][m] >= 0, 1e-300, -1e-300)
                nflip += int(m.sum())
                where[name] = where.get(name, 0) + int(m.sum())
    return nflip, worst, where
