"""The `i8ie` Python surface on the GPU: the reference's own unit tests restated
(unittest/test_layers.py, test_quantization.py, test_refcount.py,
test_tensor_ops.py, test_quantized_layer.py) plus whole-network parity against
the oracle pipeline (bit-exact logits)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def i8ie():
    import int8inferenceengine_amd  # noqa: F401
    import i8ie as mod

    return mod


def _u(shape, lo=-1, hi=1, seed=0):
    return np.random.default_rng(seed).uniform(lo, hi, shape).astype(np.float32)


# ---- unittest/test_layers.py:13-71 (FP32 path vs torch, atol 0.1) ---------------
def test_fp32_linear_vs_torch(i8ie):
    import torch

    w, b, x = _u((500, 800), seed=1), _u(500, seed=2), _u((200, 800), seed=3)
    fc = i8ie.Linear(800, 500)
    fc.load_weight(w)
    fc.load_bias(b)
    want = torch.nn.functional.linear(torch.tensor(x), torch.tensor(w), torch.tensor(b)).numpy()
    assert np.allclose(fc(i8ie.tensor(x)).numpy(), want, atol=0.1)


@pytest.mark.parametrize("cfg", [((30, 10, 22, 22), 1, 0), ((30, 10, 22, 22), 1, 1), ((30, 10, 50, 50), 7, 3)])
def test_fp32_conv2d_vs_torch(i8ie, cfg):
    import torch

    shape, stride, pad = cfg
    w, b, x = _u((20, 10, 3, 3), seed=4), _u(20, seed=5), _u(shape, seed=6)
    conv = i8ie.Conv2d(10, 20, 3, stride=stride, padding=pad)
    conv.load_weight(w)
    conv.load_bias(b)
    want = torch.nn.functional.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(b), stride=stride,
                                      padding=pad).numpy()
    got = conv(i8ie.tensor(x)).numpy()
    assert got.shape == want.shape and np.allclose(got, want, atol=0.1)


# ---- unittest/test_quantization.py:13-23 -------------------------------------------
def test_quantize_dequantize_roundtrip(i8ie):
    a = _u((4, 4), seed=7)
    q = i8ie.quantize(i8ie.tensor(a), 0.025, 100)
    assert q.scale == np.float32(0.025) and q.zero_point == 100 and q.numpy().dtype == np.uint8
    assert np.allclose(a, (q.numpy().astype(np.float32) - 100) * 0.025, atol=0.1)
    assert np.allclose(a, i8ie.dequantize(q).numpy(), atol=0.1)


# ---- unittest/test_refcount.py:36-45 ----------------------------------------------
def test_refcount_through_layers(i8ie):
    fc = i8ie.Linear(4, 4)
    t = i8ie.tensor(_u((4, 4), -100, 100, seed=8))
    fc(t)
    t = fc(t)
    u = fc(t)
    t = fc(u)
    assert t.data.ref_count() == 1 and u.data.ref_count() == 1


# ---- unittest/test_tensor_ops.py:40-46 --------------------------------------------
def test_fp32_max_pool_and_relu_vs_torch(i8ie):
    import torch

    a = _u((1, 1, 4, 4), -100, 100, seed=9)
    for k, s in ((2, 2), (2, 1), (1, 2)):
        want = torch.nn.functional.max_pool2d(torch.tensor(a), k, s).numpy()
        assert np.array_equal(i8ie.max_pool2d(i8ie.tensor(a), k, s).numpy(), want)
    assert np.array_equal(i8ie.relu(i8ie.tensor(a)).numpy(), np.maximum(a, 0))


# ---- wrong-state calls are errors, not segfaults (src/conv2d.cc:105) -----------------
def test_wrong_state_calls_raise(i8ie):
    fc = i8ie.Linear(8, 4)
    q = i8ie.quantize(i8ie.tensor(_u((2, 8))), 0.025, 127)
    with pytest.raises(RuntimeError):
        fc(q)  # INT8 input before convert()
    fc.convert()
    with pytest.raises(RuntimeError):
        fc(i8ie.tensor(_u((2, 8))))  # FP32 input after convert(): FP32 weights were released
    assert fc(q).numpy().shape == (2, 4)
    fc.convert()  # "already quantized": a no-op (src/layer.cc:37-40)


# ---- whole networks: calibrate with the product, check against the oracle -------------
@pytest.mark.parametrize("name,batch", [("mnist_fc", 100), ("two_conv", 100), ("simple_conv", 100), ("alexnet", 4)])
def test_network_logits_bit_exact(i8ie, orc, name, batch):
    import pipeline
    from int8inferenceengine_amd import workloads as wl

    sd = wl.synthetic_state_dict(name)
    net = wl.calibrated(name, sd)
    assert net.is_quant
    x = wl.synthetic_input(name, batch, seed=5)
    y = net(i8ie.tensor(x))
    got = y.numpy()
    entry = wl.NETWORKS[name]
    qlayers = pipeline.quantize_layers(entry, sd)
    for attr in wl.layer_names(name):  # a10: product-side weight quantisation == oracle's
        L = getattr(net, attr).layer
        assert np.array_equal(L.q_weight(), qlayers[attr][0]) and np.array_equal(L.q_bias(), qlayers[attr][1])
        assert np.float32(L.weight_scale()) == qlayers[attr][2]
    qparams = {a: getattr(net, a).output_qparams() for a in wl.layer_names(name)}
    cap = {}
    want = pipeline.forward(entry, x, qlayers, qparams, capture=cap)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # calibration gave usable ranges: the logits are not all one level
    assert len(np.unique(cap["_logits_u8"])) > 4


def test_per_layer_accumulators_through_python_api(i8ie, orc):
    """forward_debug exposes the INT32 pre-requant accumulators of each layer."""
    from int8inferenceengine_amd import workloads as wl

    name = "two_conv"
    sd = wl.synthetic_state_dict(name)
    net = wl.calibrated(name, sd)
    x = wl.synthetic_input(name, 10, seed=6)
    q = i8ie.quantize(i8ie.tensor(x), 0.025, 127)
    out, acc = net.conv1.forward_debug(q)
    qw, qb, s_w = orc.quantize_weight(sd["conv1.weight"], sd["conv1.bias"])
    s_out, zp_out = net.conv1.output_qparams()
    want, wacc = orc.conv2d(q.numpy(), qw, qb, 1, 0, np.float32(0.025), 127, s_w, np.float32(s_out), zp_out,
                            want_acc=True)
    assert np.array_equal(acc, wacc) and np.array_equal(out.numpy(), want)
    assert out.scale == np.float32(s_out) and out.zero_point == zp_out  # src/conv2d.cc:112-113


def test_int8_tracks_fp32_like_reference_test(i8ie):
    """unittest/test_quantized_layer.py:59-95: >= 80 % of de-quantised INT8 outputs within rtol 0.3
    of the FP32 outputs (there vs torch with a trained checkpoint; here vs our FP32 path, synthetic
    weights), checked at the first conv and after the first pool."""
    from int8inferenceengine_amd import workloads as wl

    name = "two_conv"
    sd = wl.synthetic_state_dict(name)
    fp = wl.build(name)
    fp.load(sd)
    qn = wl.calibrated(name, sd, calib_batch=_u((100, 1, 28, 28), -2, 2, seed=10))
    x = _u((10, 1, 28, 28), -2, 2, seed=11)
    q = qn.conv1(i8ie.quantize(i8ie.tensor(x), 0.025, 127))
    f = fp.conv1(i8ie.tensor(x))

    def close_frac(a, b):
        return np.isclose(a, b, rtol=0.3).sum() / a.size

    assert close_frac(f.numpy(), i8ie.dequantize(q).numpy()) > 0.8
    q, f = i8ie.max_pool2d(q, 2, 2), i8ie.max_pool2d(f, 2, 2)
    assert close_frac(f.numpy(), i8ie.dequantize(q).numpy()) > 0.8


def test_alexnet_batch_invariance(i8ie):
    """Full path at a larger batch: a slice of a 96-image batch equals the same images alone."""
    from int8inferenceengine_amd import workloads as wl

    net = wl.calibrated("alexnet")
    x = wl.synthetic_input("alexnet", 96, seed=12)
    full = net(i8ie.tensor(x)).numpy()
    part = net(i8ie.tensor(x[40:44])).numpy()
    assert np.array_equal(full[40:44].view(np.uint32), part.view(np.uint32))
    assert len(np.unique(full.argmax(1))) >= 1


def test_views_of_nhwc_activations_keep_reference_element_order(i8ie):
    """Conv outputs live NHWC internally.  reshape() views are defined on the reference's NCHW element order:
    the flatten feeding a Linear layer stays lazy, every other view (and anything observing the bytes) must
    see NCHW order."""
    from int8inferenceengine_amd import workloads as wl

    net = wl.calibrated("alexnet", wl.synthetic_state_dict("alexnet", seed=3))  # conv1: 96 features -> NHWC
    x = wl.synthetic_input("alexnet", 2, seed=9)
    conv = getattr(net, wl.layer_names("alexnet")[0])
    q = i8ie.quantize(i8ie.tensor(x), 0.02, 120)
    y = i8ie.relu(conv(q))                      # NHWC inside
    assert i8ie.relu(conv(q)).data.layout() == 1
    ref = y.numpy()                             # NCHW bytes, as the reference defines them
    n, c, h, w = ref.shape
    assert np.array_equal(y.reshape(n, -1).numpy(), ref.reshape(n, -1))           # flatten view
    assert np.array_equal(y.reshape(n, c, h * w, 1).numpy(), ref.reshape(n, c, h * w, 1))  # 4-D -> 4-D view
    assert np.array_equal(y.reshape(-1, c * h * w).numpy(), ref.reshape(-1, c * h * w))
    y2 = i8ie.relu(conv(q))
    v = y2.reshape(n, c, h * w, 1)              # view taken BEFORE anything observed the buffer
    assert np.array_equal(i8ie.relu(v).numpy(), ref.reshape(n, c, h * w, 1))
    assert np.array_equal(i8ie.max_pool2d(y2.reshape(n, c, h, w), 3, 2).numpy(),
                          i8ie.max_pool2d(y, 3, 2).numpy())


def test_deferred_launches_can_be_observed_more_than_once(i8ie):
    """y = layer(x); z = relu(y) record ONE deferred launch shared by y and z.  Observing both, in either order,
    must work (the reference computes both eagerly): the second observation launches the layer again instead
    of finding its input gone.  Same for dequantize(head(x)) followed by head(x) itself."""
    from int8inferenceengine_amd import workloads as wl

    net = wl.calibrated("alexnet", wl.synthetic_state_dict("alexnet", seed=3))
    conv = getattr(net, wl.layer_names("alexnet")[0])
    fc = getattr(net, wl.layer_names("alexnet")[-1])
    q = i8ie.quantize(i8ie.tensor(wl.synthetic_input("alexnet", 2, seed=9)), 0.02, 120)
    for first in ("relu", "plain"):
        y = conv(q)
        z = i8ie.relu(y)
        a, b = (z.numpy(), y.numpy()) if first == "relu" else tuple(reversed((y.numpy(), z.numpy())))
        assert np.array_equal(a, np.maximum(b, y.zero_point))
        assert (b < y.zero_point).any()  # the un-fused result really is the pre-relu tensor
    h = i8ie.quantize(i8ie.tensor(_u((3, 4096), seed=11)), 0.01, 128)
    u = fc(h)
    f = i8ie.dequantize(u)
    fv, uv = f.numpy(), u.numpy()
    want = ((uv.astype(np.int32) - int(u.zero_point)).astype(np.float32) * np.float32(u.scale)).astype(np.float32)
    assert np.array_equal(fv.view(np.uint32), want.view(np.uint32))
    # the same for a recorded max_pool2d and relu(max_pool2d(.))
    y = i8ie.relu(conv(q))
    pz = i8ie.max_pool2d(y, 3, 2)
    rz = i8ie.relu(pz)
    assert np.array_equal(rz.numpy(), np.maximum(pz.numpy(), pz.zero_point))
    # a subsampling 1 x 1 pool behind a still-pending conv: no kernel folds it, the result has the pooled shape and bytes
    # (round 3 returned the unpooled storage under the pooled shape)
    full = i8ie.relu(conv(q)).numpy()
    sub = i8ie.max_pool2d(i8ie.relu(conv(q)), 1, 2).numpy()
    assert sub.shape == full[:, :, ::2, ::2].shape and np.array_equal(sub, full[:, :, ::2, ::2])
    # and the whole network still gives the same logits when an intermediate was peeked at
    x = i8ie.tensor(wl.synthetic_input("alexnet", 2, seed=5))
    assert np.array_equal(net(x).numpy(), net(x).numpy())


def test_recorded_launches_are_shared_between_holders(i8ie):
    """A recorded op is one node shared by every holder and remembers what it produced: a tensor
    consumed by two ops, or observed after its consumer ran, must not launch its producers again.  Counted with
    the per-kernel profile: conv1 runs once although its output feeds two pools and is then read back."""
    import _CXX_i8ie as cx
    from int8inferenceengine_amd import workloads as wl

    net = wl.calibrated("alexnet", wl.synthetic_state_dict("alexnet", seed=3))
    q = i8ie.quantize(i8ie.tensor(wl.synthetic_input("alexnet", 2, seed=9)), 0.025, 127)
    cx.synchronize()
    cx.profile_start()
    try:
        y = i8ie.relu(net.conv1(q))
        p1 = i8ie.max_pool2d(y, 3, 2)
        p2 = i8ie.max_pool2d(y, 2, 2)  # a second consumer of the same recorded tensor
        a, b = p1.numpy(), p2.numpy()
        yv = y.numpy()                 # ... which is then observed itself
        again = i8ie.max_pool2d(y, 3, 2).numpy()
    finally:
        prof = cx.profile_stop()
    convs = sum(v[0] for k, v in prof.items() if k.startswith(("stem_conv", "conv_smallc", "igemm_conv")))
    assert convs == 1, prof
    assert np.array_equal(a, again) and a.shape == (2, 96, 27, 27) and b.shape == (2, 96, 27, 27) and yv.shape == (2, 96, 55, 55)
    # ... while the same expression with no other holder of the conv output is ONE contraction launch with the pool
    # folded into it (csrc/i8ie_stem.hip), and the same bytes
    cx.profile_start()
    try:
        fused = i8ie.max_pool2d(i8ie.relu(net.conv1(q)), 3, 2).numpy()
    finally:
        prof2 = cx.profile_stop()
    assert prof2.get("stem_conv_pool", (0,))[0] == 1 and "maxpool_u8_nhwc" not in prof2, prof2
    assert np.array_equal(fused, a)
    # the values are the reference's: pools of the observed tensor
    want = np.zeros_like(a)
    for m in range(3):
        for l in range(3):
            want = np.maximum(want, yv[:, :, m:m + 53:2, l:l + 53:2])
    assert np.array_equal(a, want)


def test_s8_overloads_of_relu_and_max_pool(i8ie):
    """src/functional.cc:78-82 registers relu / max_pool2d for Tensor<s8> too: the generic templates
    (relu: x > 0 ? x : 0 without carrying scale / zero point, :5-13; max-pool: running maximum from -127,
    scale and zero point copied, :28-31, 36-64)."""
    import _CXX_i8ie as cx

    rng = np.random.default_rng(3)
    a = rng.integers(-128, 128, (2, 5, 9, 7)).astype(np.int8)
    t = cx.tensor_s8(a, 0.5, 9)
    r = cx.relu(t)
    assert type(r).__name__ == "6TensorIcE" and np.array_equal(r.numpy(), np.maximum(a, 0))
    assert r.scale() == 1.0 and r.zero_point() == 0
    for k, s in ((3, 2), (2, 2), (2, 1), (1, 2)):
        p = cx.max_pool2d(t, k, s)
        oh, ow = (9 - k) // s + 1, (7 - k) // s + 1
        want = np.full((2, 5, oh, ow), -127, np.int64)
        for m in range(k):
            for l in range(k):
                want = np.maximum(want, a[:, :, m:m + s * (oh - 1) + 1:s, l:l + s * (ow - 1) + 1:s])
        assert np.array_equal(p.numpy(), want.astype(np.int8))
        assert p.scale() == 0.5 and p.zero_point() == 9
    cx.relu(type(t)())  # the default-constructed tensor (all the reference can build from Python) is accepted


def test_shape_needs_no_launch_and_no_copy(i8ie):
    """Tensor.shape comes from metadata: reading it must not realise a pending launch."""
    from int8inferenceengine_amd import workloads as wl

    net = wl.calibrated("two_conv")
    y = net.conv1(i8ie.quantize(i8ie.tensor(wl.synthetic_input("two_conv", 3, seed=1)), 0.025, 127))
    import _CXX_i8ie as cx

    cx.profile_start()
    try:
        assert y.shape == (3, 20, 24, 24)
    finally:
        prof = cx.profile_stop()
    assert len(prof) == 0, prof
