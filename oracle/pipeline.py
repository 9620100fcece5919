"""Whole-network INT8 forward composed from the oracle's ops (TEST INFRASTRUCTURE
ONLY; see oracle/i8ie_oracle.c).  Mirrors the reference call stack of SURVEY.md
section 3(A): Module.__call__ quantises with 0.025/127 (i8ie/module.py:20), the
user forward chains Conv2d / relu / max_pool2d / reshape / Linear, then
dequantises (i8ie/module.py:23)."""
import numpy as np

import orc

INPUT_SCALE = np.float32(0.025)
INPUT_ZP = 127


def quantize_layers(networks_entry, state_dict):
    """{attr: (qw, qb, s_w)} via the oracle's quantize_weight (src/layer.cc:6-26)."""
    layers = networks_entry[0]
    out = {}
    for attr in layers:
        qw, qb, s_w = orc.quantize_weight(state_dict[attr + ".weight"], state_dict[attr + ".bias"])
        out[attr] = (qw, qb, s_w)
    return out


def forward(networks_entry, x, qlayers, out_qparams, capture=None):
    """x: float32 NCHW.  out_qparams: {attr: (scale, zp)}.  Returns float32 logits.
    capture: optional dict filled with {attr: u8 output of that layer} and '_logits_u8'."""
    layers, spec, _ = networks_entry
    q = orc.quantize(x, INPUT_SCALE, INPUT_ZP)
    s, zp = INPUT_SCALE, INPUT_ZP
    for op in spec:
        if op[0] == "layer":
            L = layers[op[1]]
            qw, qb, s_w = qlayers[op[1]]
            s_out, zp_out = out_qparams[op[1]]
            s_out = np.float32(s_out)
            if L[0] == "conv":
                q, _ = orc.conv2d(q, qw, qb, L[4], L[5], s, zp, s_w, s_out, zp_out)
            else:
                q, _, _ = orc.linear(q.reshape(q.shape[0], -1), qw, qb, s, zp, s_w, s_out, zp_out)
            s, zp = s_out, int(zp_out)
            if capture is not None:
                capture[op[1]] = q.copy()
        elif op[0] == "relu":
            q = orc.relu(q, zp)
        elif op[0] == "pool":
            q = orc.max_pool2d(q, op[1], op[2])
        else:
            q = q.reshape(-1, op[1])
    if capture is not None:
        capture["_logits_u8"] = q.copy()
        capture["_logits_qparams"] = (s, zp)
    return orc.dequantize(q, s, zp)
