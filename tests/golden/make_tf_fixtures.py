"""Recipe that would PIN the oracle on the reference itself: import the reference's own model classes
(gym/envs/mujoco/arm_shaping.py: ContextSkipNew :1260-1354, ContextAEReal :1599-1684, ContextAEInception2 :1786-1894; and
nets/inception_v3.py:93-416 to Mixed_7c) under TensorFlow 1.x, feed them the
inputs and parameters of the committed oracle fixtures, and write the reference's outputs in the SAME .npz schema
(tests/golden/make_golden.py: make / make_real) as  tests/golden/tf_<tag>.npz -- or, with --check, compare them with the
committed oracle fixtures and print the deviations.

IT CANNOT RUN IN THE BUILD CONTAINER TODAY: `import tensorflow` fails there and the module imports
tf.contrib.slim, so it needs a TF 1.x environment (1.4 ... 1.15, with matplotlib and scipy as the reference's
environment.yml lists them) plus a checkout of the reference:

    REFERENCE_ROOT=/path/to/imitation_from_observation  python tests/golden/make_tf_fixtures.py [--check]

Until someone runs it, the oracle stays "parity unpinned" (oracle/ctx_oracle.py header).  Once
tests/golden/tf_*.npz exist, tests/test_tf_pin.py (CPU) checks every array in them against the oracle's fixtures (<= 1e-4).  Nothing from
the reference is copied: the script loads the module from REFERENCE_ROOT at run time and only stores arrays.
"""
import argparse
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ctx_oracle as o  # noqa: E402
from oracle import ctx_oracle_real as r  # noqa: E402

N_HEAD = 64


def load_reference_module(root):
    """gym/envs/mujoco/arm_shaping.py as a stand-alone module (importing the `gym` package would pull mujoco_py)."""
    sys.path.insert(0, root)                                        # `from nets import inception_v3`, arm_shaping.py:8
    path = os.path.join(root, "gym", "envs", "mujoco", "arm_shaping.py")
    spec = importlib.util.spec_from_file_location("ref_arm_shaping", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def digest(a):
    a = np.asarray(a, np.float64).reshape(-1)
    return np.array([a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())]), a[:N_HEAD].copy()


def run_reference(mod, cls_name, p, frames_u8, lr, steps):
    """Builds the reference graph exactly as scripts/train_script.py:118-129 does (float placeholder (3,B,H,W,3), Adam on
    model.loss), assigns the fixture's parameters to the TF variables BY NAME, and fetches everything the schema holds."""
    import tensorflow as tf
    tf1 = tf.compat.v1 if hasattr(tf, "compat") and hasattr(tf.compat, "v1") else tf
    tf1.reset_default_graph()
    B, H, W, _ = frames_u8[0].shape
    image = tf1.placeholder(tf.float32, (3, B, H, W, 3))
    model = getattr(mod, cls_name)()
    model.build(image)                                              # arm_shaping.py:1272 / :1611
    tvars = {v.name[:-2]: v for v in tf1.trainable_variables()}
    missing = sorted(set(p) - set(tvars)), sorted(set(tvars) - set(p))
    assert missing == ([], []), f"variable names differ from the oracle's inventory: {missing}"
    lrph = tf1.placeholder(tf.float32, ())
    names = list(p)
    grads = tf1.gradients(model.loss, [tvars[n] for n in names])
    opt = tf1.train.AdamOptimizer(lrph).minimize(model.loss, var_list=[tvars[n] for n in names])   # train_script.py:124-128
    # the sampler's uint8 path (rllab/sampler/base.py:116-119) on a second placeholder
    u8 = tf1.placeholder(tf.uint8, (3, B, H, W, 3))
    pre = tf1.multiply(tf1.subtract(tf.image.convert_image_dtype(u8, dtype=tf.float32), 0.5), 2.0)
    out = {}
    with tf1.Session() as sess:
        sess.run(tf1.global_variables_initializer())
        for n in names:
            sess.run(tvars[n].assign(np.asarray(p[n], np.float32)))
        src, ctx, tgt = (sess.run(pre, {u8: np.stack([f, f, f])})[0] for f in frames_u8)
        feed = {image: np.stack([src, ctx, tgt])}                   # slot 0 src, 1 ctx, 2 tgt (arm_shaping.py:1278-1280)
        fetch = dict(input_z=model.input_z, translated_z=model.translated_z, out=model.out, out2=model.out2, loss=model.loss,
                     simloss=model.simloss, recon1=model.recon1, recon2=model.recon2)
        res = sess.run(fetch, feed)
        out.update({k: np.asarray(res[k], np.float32) for k in ("input_z", "translated_z", "out", "out2")})
        out["scalars"] = np.array([res["loss"], res["simloss"], res["recon1"], res["recon2"]], np.float64)
        g = sess.run(grads, feed)
        out["grad_digest"] = np.stack([digest(x)[0] for x in g])
        out["grad_head"] = np.stack([np.pad(digest(x)[1], (0, N_HEAD - min(N_HEAD, x.size))) for x in g])
        # the reward hook's fetches: [translated_z, out] with image = [src, [ctx0]*B, [ctx0]*B]; input_z of the src slot
        c0 = np.broadcast_to(ctx[0], src.shape)
        tz, pred = sess.run([model.translated_z, model.out], {image: np.stack([src, c0, c0])})
        out["translate_pred"], out["translate_feat"] = np.asarray(pred, np.float32), np.asarray(tz, np.float32)
        out["encode_feat"] = out["input_z"]
        traj = []
        for _ in range(steps):
            _, l, s, r1, r2 = sess.run([opt, model.loss, model.simloss, model.recon1, model.recon2], {**feed, lrph: lr})
            traj.append([l, s, r1, r2])
        out["train_scalars"] = np.array(traj, np.float64)
    return out


def run_reference_incep2(mod, z, p):
    """ContextAEInception2(strides, kernels, filters).build(image) on float feature maps [3,B,h,w,C] (arm_shaping.py:1786-1894;
    train_script.py:110-121 builds it on featreshape): same fetches as run_reference, no uint8 path (image_trans IS the input)."""
    import tensorflow as tf
    tf1 = tf.compat.v1 if hasattr(tf, "compat") and hasattr(tf.compat, "v1") else tf
    tf1.reset_default_graph()
    src, ctx, tgt = (np.asarray(z[k], np.float32) for k in ("src_f32", "ctx_f32", "tgt_f32"))
    image = tf1.placeholder(tf.float32, (3,) + src.shape)
    model = mod.ContextAEInception2([int(v) for v in z["strides"]], [int(v) for v in z["kernels"]], [int(v) for v in z["filters"]])
    model.build(image)
    tvars = {v.name[:-2]: v for v in tf1.trainable_variables()}
    missing = sorted(set(p) - set(tvars)), sorted(set(tvars) - set(p))
    assert missing == ([], []), f"variable names differ from the oracle's inventory: {missing}"
    names = list(p)
    lrph = tf1.placeholder(tf.float32, ())
    grads = tf1.gradients(model.loss, [tvars[n] for n in names])
    opt = tf1.train.AdamOptimizer(lrph).minimize(model.loss, var_list=[tvars[n] for n in names])
    out = {}
    with tf1.Session() as sess:
        sess.run(tf1.global_variables_initializer())
        for n in names:
            sess.run(tvars[n].assign(np.asarray(p[n], np.float32)))
        feed = {image: np.stack([src, ctx, tgt])}
        fetch = dict(input_z=model.input_z, translated_z=model.translated_z, out=model.out, out2=model.out2, loss=model.loss,
                     simloss=model.simloss, recon1=model.recon1, recon2=model.recon2)
        res = sess.run(fetch, feed)
        out.update({k: np.asarray(res[k], np.float32) for k in ("input_z", "translated_z", "out", "out2")})
        out["scalars"] = np.array([res["loss"], res["simloss"], res["recon1"], res["recon2"]], np.float64)
        g = sess.run(grads, feed)
        out["grad_digest"] = np.stack([digest(x)[0] for x in g])
        out["grad_head"] = np.stack([np.pad(digest(x)[1], (0, N_HEAD - min(N_HEAD, x.size))) for x in g])
        c0 = np.broadcast_to(ctx[0], src.shape)
        tz, pred = sess.run([model.translated_z, model.out], {image: np.stack([src, c0, c0])})
        out["translate_pred"], out["translate_feat"] = np.asarray(pred, np.float32), np.asarray(tz, np.float32)
        out["encode_feat"] = out["input_z"]
        traj = []
        for _ in range(int(z["steps"])):
            _, l, s_, r1, r2 = sess.run([opt, model.loss, model.simloss, model.recon1, model.recon2], {**feed, lrph: float(z["lr"])})
            traj.append([l, s_, r1, r2])
        out["train_scalars"] = np.array(traj, np.float64)
    return out


def run_reference_inception_v3(root, z):
    """inception_v3.inception_v3(images, num_classes=1001, is_training=False)[1] (rllab/sampler/base.py:122-127;
    nets/inception_v3.py:93-416) with the oracle's synthetic variables assigned BY NAME; returns the end points the fixture holds."""
    import tensorflow as tf
    tf1 = tf.compat.v1 if hasattr(tf, "compat") and hasattr(tf.compat, "v1") else tf
    from oracle import inception_oracle as io
    sys.path.insert(0, root)
    from nets import inception_v3 as ref_net                      # the reference's own file, loaded from REFERENCE_ROOT
    slim = tf.contrib.slim
    tf1.reset_default_graph()
    frames = z["frames_u8"]
    u8 = tf1.placeholder(tf.uint8, frames.shape)
    images = tf1.multiply(tf1.subtract(tf.image.convert_image_dtype(u8, dtype=tf.float32), 0.5), 2.0)     # base.py:116-119
    with slim.arg_scope(ref_net.inception_v3_arg_scope()):
        _, end_points = ref_net.inception_v3(images, num_classes=1001, is_training=False)
    p = io.init_params(int(z["pseed"]), np.float32)
    allv = {v.name[:-2]: v for v in tf1.global_variables()}
    names = [str(n) for n in z["endpoints"]]
    with tf1.Session() as sess:
        sess.run(tf1.global_variables_initializer())
        for n, a in p.items():
            sess.run(allv["InceptionV3/" + n].assign(a))
        got = sess.run([end_points[n] for n in names], {u8: frames})
    ep = dict(zip(names, got))
    return {"Mixed_7c": np.asarray(ep["Mixed_7c"], np.float32),
            "endpoint_digest": np.stack([digest(v)[0] for v in got]),
            "endpoint_head": np.stack([digest(v)[1] for v in got])}


def fixture_inputs(path, real):
    z = np.load(path)
    if real:
        H, W, C, F = (int(v) for v in z["cfg"])
        cfg = r.RealConfig(H=H, W=W, C=C, featsize=F)
        p = r.init_params(cfg, int(z["pseed"]), np.float64, stddev=float(z["stddev"]))
    else:
        H, W, C, d, F = (int(v) for v in z["cfg"])
        cfg = o.SkipNewConfig(H=H, W=W, C=C, df_dim=d, gf_dim=d, featsize=F)
        p = o.init_params(cfg, int(z["pseed"]), np.float64, stddev=float(z["stddev"]))
    brng = np.random.default_rng(int(z["pseed"]) + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * float(z["stddev"])
    return z, cfg, p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="compare with the committed oracle fixtures instead of writing tf_*.npz")
    args = ap.parse_args()
    root = os.environ.get("REFERENCE_ROOT", "/root/reference")
    mod = load_reference_module(root)
    # ContextSkipNew hard-codes featsize 1024 and df_dim 64 defaults (arm_shaping.py:1261-1277): only the production-size
    # fixture can be reproduced by the unmodified class; ContextAEReal hard-codes featsize 100 (:1616).
    jobs = [("skipnew_d64_f1024_64x64_b2", "ContextSkipNew", False), ("real_f100_36x64_b3", "ContextAEReal", True)]
    worst = 0.0
    for tag, cls_name, real in jobs:
        z, cfg, p = fixture_inputs(os.path.join(HERE, tag + ".npz"), real)
        got = run_reference(mod, cls_name, p, [z["src_u8"], z["ctx_u8"], z["tgt_u8"]], float(z["lr"]), int(z["steps"]))
        if args.check:
            for k, v in got.items():
                ref = np.asarray(z[k], np.float64)
                dev = float(np.abs(np.asarray(v, np.float64) - ref).max() / (np.abs(ref).max() + 1e-30))
                worst = max(worst, dev)
                print(f"{tag:32s} {k:16s} max deviation / max |oracle| = {dev:.2e}")
        else:
            keep = {k: z[k] for k in ("cfg", "B", "pseed", "stddev", "src_u8", "ctx_u8", "tgt_u8", "lr", "steps", "param_digest")}
            np.savez_compressed(os.path.join(HERE, f"tf_{tag}.npz"), **keep, **got)
            print("wrote", f"tf_{tag}.npz")
    # ContextAEInception2 (strides / kernels / filters are constructor arguments) and the Inception-v3 front end
    from oracle import ctx_oracle_incep as ci
    extra = []
    for tag in ("incep2_4x4x64_f32_b2", "incep2_8x4x32_k5331_s2121_b2"):
        z = np.load(os.path.join(HERE, tag + ".npz"))
        H, W, C, F = (int(v) for v in z["cfg"])
        cfg = ci.Incep2Config(H=H, W=W, C=C, featsize=F, strides=tuple(int(v) for v in z["strides"]),
                              kernels=tuple(int(v) for v in z["kernels"]), filters=tuple(int(v) for v in z["filters"]))
        p = ci.init_params(cfg, int(z["pseed"]), np.float64, stddev=float(z["stddev"]))
        brng = np.random.default_rng(int(z["pseed"]) + 1)
        for n in p:
            if n.endswith("bias") or n.endswith("biases"):
                p[n] = brng.standard_normal(p[n].shape) * float(z["stddev"])
        keep = ("cfg", "strides", "kernels", "filters", "B", "pseed", "stddev", "fseed", "lr", "steps", "param_digest")
        extra.append((tag, z, run_reference_incep2(mod, z, p), keep))
    z = np.load(os.path.join(HERE, "inception_v3_125x125_b2.npz"))
    extra.append(("inception_v3_125x125_b2", z, run_reference_inception_v3(root, z), ("pseed", "fseed", "B", "S", "endpoints", "param_digest")))
    for tag, z, got, keep in extra:
        if args.check:
            for k, v in got.items():
                ref = np.asarray(z[k], np.float64)
                dev = float(np.abs(np.asarray(v, np.float64) - ref).max() / (np.abs(ref).max() + 1e-30))
                worst = max(worst, dev)
                print(f"{tag:32s} {k:16s} max deviation / max |oracle| = {dev:.2e}")
        else:
            np.savez_compressed(os.path.join(HERE, f"tf_{tag}.npz"), **{k: z[k] for k in keep}, **got)
            print("wrote", f"tf_{tag}.npz")
    if args.check:
        print("worst deviation:", worst, "-> oracle", "PINNED (<= 1e-4)" if worst <= 1e-4 else "DIFFERS from the reference")


if __name__ == "__main__":
    main()
