"""Independent torch-CPU autograd statement of ContextSkipNew (arm_shaping.py:1272-1354), used
only to cross-check the numpy oracle.  It shares NO code with
oracle/ctx_oracle.py: convs go through torch's own conv2d / conv_transpose2d with TF's SAME
semantics emulated by explicit pad / crop, and gradients come from autograd."""
import torch
import torch.nn.functional as F


def lrelu(x):
    return torch.maximum(x, 0.2 * x)


def tf_conv(x_nhwc, w_hwio, b):
    x = x_nhwc.permute(0, 3, 1, 2)
    H, W = x.shape[2:]
    assert H % 2 == 0 and W % 2 == 0
    x = F.pad(x, (1, 2, 1, 2))                       # SAME, k=5, s=2, even input: (1, 2)
    y = F.conv2d(x, w_hwio.permute(3, 2, 0, 1), b, stride=2)
    return y.permute(0, 2, 3, 1)


def tf_deconv(x_nhwc, w_hwoi, b):
    x = x_nhwc.permute(0, 3, 1, 2)
    h, w = x.shape[2:]
    # torch conv_transpose2d weight is [in, out, kh, kw]; TF's is [kh, kw, out, in]
    full = F.conv_transpose2d(x, w_hwoi.permute(3, 2, 0, 1), None, stride=2, padding=0)
    y = full[:, :, 1:1 + 2 * h, 1:1 + 2 * w] + b.view(1, -1, 1, 1)
    return y.permute(0, 2, 3, 1)


def forward(p, src, ctx, tgt, H, W, gf):
    def enc(scope, img, z_lrelu):
        acts, h = [], img
        for k in range(4):
            h = lrelu(tf_conv(h, p[f"{scope}/h{k}_conv/w"], p[f"{scope}/h{k}_conv/biases"]))
            acts.append(h)
        h4 = lrelu(h.reshape(h.shape[0], -1) @ p[f"{scope}/h4_lin/Matrix"] + p[f"{scope}/h4_lin/bias"])
        z = h4 @ p[f"{scope}/hz_lin/Matrix"] + p[f"{scope}/hz_lin/bias"]
        return acts, (lrelu(z) if z_lrelu else z)

    def dec(z, skips):
        h = lrelu(z @ p["deconv/d_h0_lin/Matrix"] + p["deconv/d_h0_lin/bias"]).reshape(-1, H // 16, W // 16, 8 * gf)
        for k in range(1, 5):
            h = tf_deconv(torch.cat([h, skips[4 - k]], 3), p[f"deconv/d_h{k}/w"], p[f"deconv/d_h{k}/biases"])
            if k < 4:
                h = lrelu(h)
        return h

    skips, ctx_z = enc("conv_context", ctx, False)
    _, src_z = enc("conv", src, True)
    _, tgt_z = enc("conv", tgt, True)
    th0 = lrelu(torch.cat([src_z, ctx_z], 1) @ p["translate/trans_h0/Matrix"] + p["translate/trans_h0/bias"])
    trans_z = th0 @ p["translate/trans_z/Matrix"] + p["translate/trans_z/bias"]
    out, out2 = dec(trans_z, skips), dec(tgt_z, skips)
    sim = ((trans_z - tgt_z) ** 2).mean() * 1e3
    r1 = 0.5 * ((tgt - out) ** 2).sum()
    r2 = 0.5 * ((tgt - out2) ** 2).sum()
    return dict(input_z=src_z, translated_z=trans_z, out=out, out2=out2, simloss=sim, recon1=r1,
                recon2=r2, loss=r1 + r2 + sim)


# ------------------------------------------------------------------------------------------------ ContextAEReal
def tf_conv_s(x_nhwc, w_hwio, b, s):
    """SAME conv for stride 1 (pad 2/2) or 2 on an even input (pad 1/2)."""
    x = x_nhwc.permute(0, 3, 1, 2)
    x = F.pad(x, (2, 2, 2, 2)) if s == 1 else F.pad(x, (1, 2, 1, 2))
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1), b, stride=s).permute(0, 2, 3, 1)


def tf_deconv_s(x_nhwc, w_hwoi, b, s):
    x = x_nhwc.permute(0, 3, 1, 2)
    h, w = x.shape[2:]
    full = F.conv_transpose2d(x, w_hwoi.permute(3, 2, 0, 1), None, stride=s, padding=0)
    pb = 2 if s == 1 else 1                              # SAME pad_before of the conv this is the gradient of
    y = full[:, :, pb:pb + s * h, pb:pb + s * w] + b.view(1, -1, 1, 1)
    return y.permute(0, 2, 3, 1)


def forward_real(p, src, ctx, tgt, H, W):
    """ContextAEReal (arm_shaping.py:1599-1684), keep_prob = 1."""
    NS = (1, 2, 1, 2)

    def enc(img):
        acts, h = [], img
        for k in range(4):
            h = lrelu(tf_conv_s(h, p[f"conv/h{k}_conv/w"], p[f"conv/h{k}_conv/biases"], NS[k]))
            acts.append(h)
        h4 = lrelu(h.reshape(h.shape[0], -1) @ p["conv/h4_lin/Matrix"] + p["conv/h4_lin/bias"])
        return acts, lrelu(h4 @ p["conv/hz_lin/Matrix"] + p["conv/hz_lin/bias"])

    def dec(z, skips):
        h = lrelu(z @ p["deconv/d_h0_lin/Matrix"] + p["deconv/d_h0_lin/bias"]).reshape(-1, H // 4, W // 4, 8)
        for k in range(1, 5):
            h = tf_deconv_s(torch.cat([h, skips[4 - k]], 3), p[f"deconv/d_h{k}/w"], p[f"deconv/d_h{k}/biases"], NS[4 - k])
            if k < 4:
                h = lrelu(h)
        return h

    _, src_z = enc(src)
    _, tgt_z = enc(tgt)
    skips, ctx_z = enc(ctx)
    th0 = lrelu(torch.cat([src_z, ctx_z], 1) @ p["translate/trans_h0/Matrix"] + p["translate/trans_h0/bias"])
    trans_z = th0 @ p["translate/trans_z/Matrix"] + p["translate/trans_z/bias"]
    out, out2 = dec(trans_z, skips), dec(tgt_z, skips)
    sim = ((trans_z - tgt_z) ** 2).mean() * 1e3
    r1 = 0.5 * ((tgt - out) ** 2).sum()
    r2 = 0.5 * ((tgt - out2) ** 2).sum()
    return dict(input_z=src_z, translated_z=trans_z, out=out, out2=out2, simloss=sim, recon1=r1, recon2=r2, loss=r1 + r2 + sim)


def tf_same(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def tf_conv_ks(x_nhwc, w_hwio, b, s):
    """tf.nn.conv2d, SAME, any kernel size / stride / grid (incl. 1x1 grids under stride 2)."""
    k = w_hwio.shape[0]
    x = x_nhwc.permute(0, 3, 1, 2)
    _, pt, pb = tf_same(x.shape[2], k, s)
    _, pl, pr = tf_same(x.shape[3], k, s)
    y = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w_hwio.permute(3, 2, 0, 1), b, stride=s)
    return y.permute(0, 2, 3, 1)


def tf_deconv_ks(x_nhwc, w_hwoi, b, out_hw, s):
    """tf.nn.conv2d_transpose, SAME, output_shape given: full transposed conv, cropped at the SAME pad_before of the
    forward conv over the OUTPUT grid."""
    k = w_hwoi.shape[0]
    x = x_nhwc.permute(0, 3, 1, 2)
    _, pt, _ = tf_same(out_hw[0], k, s)
    _, pl, _ = tf_same(out_hw[1], k, s)
    full = F.conv_transpose2d(x, w_hwoi.permute(3, 2, 0, 1), None, stride=s, padding=0)
    need_h, need_w = pt + out_hw[0], pl + out_hw[1]
    full = F.pad(full, (0, max(0, need_w - full.shape[3]), 0, max(0, need_h - full.shape[2])))
    y = full[:, :, pt:need_h, pl:need_w] + b.view(1, -1, 1, 1)
    return y.permute(0, 2, 3, 1)


def forward_incep2(p, src, ctx, tgt, H, W, strides, filters):
    """arm_shaping.py:1792-1894 through torch ops + autograd."""
    sizes, h, w = [], H, W
    for s in strides:
        h, w = -(-h // s), -(-w // s)
        sizes.append((h, w))

    def enc(scope, img):
        acts, hh = [], img
        for k in range(4):
            hh = lrelu(tf_conv_ks(hh, p[f"{scope}/h{k}_conv/w"], p[f"{scope}/h{k}_conv/biases"], strides[k]))
            acts.append(hh)
        h4 = lrelu(hh.reshape(hh.shape[0], -1) @ p[f"{scope}/h4_lin/Matrix"] + p[f"{scope}/h4_lin/bias"])
        return acts, lrelu(h4 @ p[f"{scope}/hz_lin/Matrix"] + p[f"{scope}/hz_lin/bias"])

    def dec(z, skips):
        hh = lrelu(z @ p["deconv/d_h0_lin/Matrix"] + p["deconv/d_h0_lin/bias"]).reshape(-1, sizes[3][0], sizes[3][1], filters[3])
        outs = [sizes[2], sizes[1], sizes[0], (H, W)]
        for k in range(1, 5):
            hh = tf_deconv_ks(torch.cat([hh, skips[4 - k]], 3), p[f"deconv/d_h{k}/w"], p[f"deconv/d_h{k}/biases"], outs[k - 1], strides[4 - k])
            if k < 4:
                hh = lrelu(hh)
        return hh

    skips, ctx_z = enc("conv_context", ctx)
    _, src_z = enc("conv", src)
    _, tgt_z = enc("conv", tgt)
    th0 = lrelu(torch.cat([src_z, ctx_z], 1) @ p["translate/trans_h0/Matrix"] + p["translate/trans_h0/bias"])
    trans_z = th0 @ p["translate/trans_z/Matrix"] + p["translate/trans_z/bias"]
    out, out2 = dec(trans_z, skips) + ctx, dec(tgt_z, skips) + ctx
    sim = ((trans_z - tgt_z) ** 2).mean() * 1e3
    r1, r2 = 0.5 * ((tgt - out) ** 2).sum(), 0.5 * ((tgt - out2) ** 2).sum()
    return dict(input_z=src_z, translated_z=trans_z, out=out, out2=out2, simloss=sim, recon1=r1, recon2=r2, loss=r1 + r2 + sim)


def inception_v3_mixed7c(p, x_nhwc):
    """nets/inception_v3.py:93-416 through torch ops (conv2d with explicit TF padding, batch_norm in eval mode without
    scale, relu, max/avg pooling with TF's SAME semantics); shares no code with oracle/inception_oracle.py."""
    x = x_nhwc.permute(0, 3, 1, 2)
    S = "InceptionV3/"

    def conv(x, scope, stride=1, padding="SAME"):
        w = p[scope + "/weights"]
        kh, kw = w.shape[0], w.shape[1]
        if padding == "SAME":
            _, pt, pb = tf_same(x.shape[2], kh, stride)
            _, pl, pr = tf_same(x.shape[3], kw, stride)
            x = F.pad(x, (pl, pr, pt, pb))
        y = F.conv2d(x, w.permute(3, 2, 0, 1), None, stride=stride)
        y = F.batch_norm(y, p[scope + "/BatchNorm/moving_mean"], p[scope + "/BatchNorm/moving_variance"], None,
                         p[scope + "/BatchNorm/beta"], training=False, eps=0.001)
        return F.relu(y)

    mp = lambda t: F.max_pool2d(t, 3, 2)
    ap = lambda t: F.avg_pool2d(t, 3, 1, padding=1, count_include_pad=False)
    x = conv(x, S + "Conv2d_1a_3x3", 2, "VALID")
    x = conv(x, S + "Conv2d_2a_3x3", 1, "VALID")
    x = conv(x, S + "Conv2d_2b_3x3")
    x = mp(x)
    x = conv(x, S + "Conv2d_3b_1x1", 1, "VALID")
    x = conv(x, S + "Conv2d_4a_3x3", 1, "VALID")
    x = mp(x)
    for name, b1 in (("Mixed_5b", ("Conv2d_0a_1x1", "Conv2d_0b_5x5")), ("Mixed_5c", ("Conv2d_0b_1x1", "Conv_1_0c_5x5")),
                     ("Mixed_5d", ("Conv2d_0a_1x1", "Conv2d_0b_5x5"))):
        P = S + name + "/"
        x = torch.cat([conv(x, P + "Branch_0/Conv2d_0a_1x1"),
                       conv(conv(x, P + "Branch_1/" + b1[0]), P + "Branch_1/" + b1[1]),
                       conv(conv(conv(x, P + "Branch_2/Conv2d_0a_1x1"), P + "Branch_2/Conv2d_0b_3x3"), P + "Branch_2/Conv2d_0c_3x3"),
                       conv(ap(x), P + "Branch_3/Conv2d_0b_1x1")], 1)
    P = S + "Mixed_6a/"
    x = torch.cat([conv(x, P + "Branch_0/Conv2d_1a_1x1", 2, "VALID"),
                   conv(conv(conv(x, P + "Branch_1/Conv2d_0a_1x1"), P + "Branch_1/Conv2d_0b_3x3"), P + "Branch_1/Conv2d_1a_1x1", 2, "VALID"),
                   mp(x)], 1)
    for name in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
        P = S + name + "/"
        t = conv(conv(conv(x, P + "Branch_1/Conv2d_0a_1x1"), P + "Branch_1/Conv2d_0b_1x7"), P + "Branch_1/Conv2d_0c_7x1")
        u = x
        for sc in ("Conv2d_0a_1x1", "Conv2d_0b_7x1", "Conv2d_0c_1x7", "Conv2d_0d_7x1", "Conv2d_0e_1x7"):
            u = conv(u, P + "Branch_2/" + sc)
        x = torch.cat([conv(x, P + "Branch_0/Conv2d_0a_1x1"), t, u, conv(ap(x), P + "Branch_3/Conv2d_0b_1x1")], 1)
    P = S + "Mixed_7a/"
    t = x
    for sc in ("Conv2d_0a_1x1", "Conv2d_0b_1x7", "Conv2d_0c_7x1"):
        t = conv(t, P + "Branch_1/" + sc)
    x = torch.cat([conv(conv(x, P + "Branch_0/Conv2d_0a_1x1"), P + "Branch_0/Conv2d_1a_3x3", 2, "VALID"),
                   conv(t, P + "Branch_1/Conv2d_1a_3x3", 2, "VALID"), mp(x)], 1)
    for name, b1b in (("Mixed_7b", "Conv2d_0b_3x1"), ("Mixed_7c", "Conv2d_0c_3x1")):
        P = S + name + "/"
        t = conv(x, P + "Branch_1/Conv2d_0a_1x1")
        u = conv(conv(x, P + "Branch_2/Conv2d_0a_1x1"), P + "Branch_2/Conv2d_0b_3x3")
        x = torch.cat([conv(x, P + "Branch_0/Conv2d_0a_1x1"),
                       conv(t, P + "Branch_1/Conv2d_0b_1x3"), conv(t, P + "Branch_1/" + b1b),
                       conv(u, P + "Branch_2/Conv2d_0c_1x3"), conv(u, P + "Branch_2/Conv2d_0d_3x1"),
                       conv(ap(x), P + "Branch_3/Conv2d_0b_1x1")], 1)
    return x.permute(0, 2, 3, 1)
