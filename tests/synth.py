"""Seeded synthetic layers/inputs shared by the tests (no reference data exists:
the reference's checkpoints and datasets are not in its repo)."""
import numpy as np


def he_weights(rng, shape):
    fan_in = int(np.prod(shape[1:]))
    w = rng.uniform(-1, 1, shape).astype(np.float32) * np.float32(np.sqrt(6.0 / fan_in))
    b = rng.uniform(-1, 1, shape[0]).astype(np.float32) * np.float32(1.0 / np.sqrt(fan_in))
    return w, b


def qparams_from_acc(acc, s_in, s_w):
    """A deterministic stand-in for calibration: cover the real-valued range of the layer output."""
    real = acc.astype(np.float64) * float(s_in) * float(s_w)
    lo, hi = min(real.min(), 0.0), max(real.max(), 0.0)
    if hi - lo < 1e-12:
        return np.float32(1.0), 0
    zp = int(255 * (0 - lo) / (hi - lo))
    scale = (hi - lo) / 255 if zp == 0 else (0 - lo) / zp
    return np.float32(scale), zp


def conv_case(orc, seed, n, c, h, w, kc, k, stride, pad, s_in=0.025, zp_in=127):
    rng = np.random.default_rng(seed)
    wf, bf = he_weights(rng, (kc, c, k, k))
    qw, qb, s_w = orc.quantize_weight(wf, bf)
    q_in = rng.integers(0, 256, (n, c, h, w), dtype=np.uint8)
    s_in = np.float32(s_in)
    _, acc = orc.conv2d(q_in, qw, qb, stride, pad, s_in, zp_in, s_w, np.float32(1), 0, want_acc=True)
    s_out, zp_out = qparams_from_acc(acc, s_in, s_w)
    out, acc = orc.conv2d(q_in, qw, qb, stride, pad, s_in, zp_in, s_w, s_out, zp_out, want_acc=True)
    return dict(q_in=q_in, qw=qw, qb=qb, s_in=s_in, zp_in=zp_in, s_w=s_w, s_out=s_out, zp_out=zp_out,
                stride=stride, pad=pad, out=out, acc=acc, wf=wf, bf=bf)


def linear_case(orc, seed, m, k, n, s_in=0.031, zp_in=64):
    rng = np.random.default_rng(seed)
    wf, bf = he_weights(rng, (n, k))
    qw, qb, s_w = orc.quantize_weight(wf, bf)
    q_in = rng.integers(0, 256, (m, k), dtype=np.uint8)
    s_in = np.float32(s_in)
    _, pre, _ = orc.linear(q_in, qw, qb, s_in, zp_in, s_w, np.float32(1), 0, want_acc=True)
    s_out, zp_out = qparams_from_acc(pre, s_in, s_w)
    out, pre, post = orc.linear(q_in, qw, qb, s_in, zp_in, s_w, s_out, zp_out, want_acc=True)
    return dict(q_in=q_in, qw=qw, qb=qb, s_in=s_in, zp_in=zp_in, s_w=s_w, s_out=s_out, zp_out=zp_out,
                out=out, acc=pre, acc_post=post, wf=wf, bf=bf)
