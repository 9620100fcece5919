"""Whole-forward replay as one HIP graph (include/i8ie_hip.h i8ie_graph_*, int8inferenceengine_amd/graph.py):
the replayed forward must produce the eager forward's bytes, for the captured input values and for new ones
loaded into the captured input buffer, and the oracle's logits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import int8inferenceengine_amd  # noqa: F401
    import _CXX_i8ie as cx
    import i8ie
    from int8inferenceengine_amd import workloads as wl
    from int8inferenceengine_amd.graph import GraphedForward
    return cx, i8ie, wl, GraphedForward


@pytest.mark.parametrize("name,batch", [("two_conv", 16), ("alexnet", 8), ("alexnet", 125)])
def test_graph_replay_matches_eager_forward(env, name, batch):
    cx, i8ie, wl, GraphedForward = env
    net = wl.calibrated(name, wl.synthetic_state_dict(name, seed=5))
    xa = wl.synthetic_input(name, batch, seed=11)
    xb = wl.synthetic_input(name, batch, seed=12)
    want_a = net(i8ie.tensor(xa)).numpy()
    want_b = net(i8ie.tensor(xb)).numpy()
    assert not np.array_equal(want_a, want_b)
    x = i8ie.tensor(xa).prefetch()
    g = GraphedForward(net, x)
    assert g.kernel_nodes >= 3 and g.nodes >= g.kernel_nodes
    for _ in range(3):
        assert np.array_equal(g().numpy(), want_a)
    g.load(xb)
    assert np.array_equal(g().numpy(), want_b)
    g.load(xa)
    assert np.array_equal(g().numpy(), want_a)
    # eager calls between replays do not disturb the captured buffers
    assert np.array_equal(net(i8ie.tensor(xb)).numpy(), want_b)
    assert np.array_equal(g().numpy(), want_a)


def test_replay_survives_an_eager_forward_at_another_batch_size(env):
    """A graph captured at 125 images (the 8-GPU shard: conv5 as two passes of 128, conv3/4 two of 192) keeps
    replaying correctly after an eager forward at 1000 images, which packs the weights of the same layers for wider
    passes (256 / 384), needs a larger workspace (the grouped first-layer image) and recycles bordered blocks: the
    packed weights live per packing key in the layer handle, the workspace the graph was captured with stays
    allocated, and a GraphedForward keeps its network alive."""
    import gc
    cx, i8ie, wl, GraphedForward = env
    xs = wl.synthetic_input("alexnet", 125, seed=21)
    xl = wl.synthetic_input("alexnet", 1000, seed=22)

    def make():
        net = wl.calibrated("alexnet", wl.synthetic_state_dict("alexnet", seed=5))
        want = net(i8ie.tensor(xs)).numpy()
        return GraphedForward(net, i8ie.tensor(xs).prefetch()), want  # the only reference to `net` is the graph's

    g, want_s = make()
    gc.collect()
    assert np.array_equal(g().numpy(), want_s)
    want_l = g.forward(i8ie.tensor(xl)).numpy()          # eager, 8 x the batch: new packing keys, workspace growth
    assert np.array_equal(g().numpy(), want_s)           # replay: its own weights, its own workspace
    assert np.array_equal(g.forward(i8ie.tensor(xl)).numpy(), want_l)
    assert np.array_equal(want_l[:5], g.forward(i8ie.tensor(xl[:5])).numpy())  # (batch invariance of the rows)
    g.load(xs[::-1].copy())
    assert np.array_equal(g().numpy(), want_s[::-1])
    assert np.array_equal(g.x.numpy(), xs[::-1])         # load() leaves no stale host mirror behind


def test_graph_logits_match_the_oracle(env, orc):
    import pipeline
    cx, i8ie, wl, GraphedForward = env
    name, batch = "alexnet", 16
    sd = wl.synthetic_state_dict(name, seed=42)
    net = wl.calibrated(name, sd)
    x_np = wl.synthetic_input(name, batch, seed=3)
    g = GraphedForward(net, i8ie.tensor(x_np).prefetch())
    got = g().numpy()
    qp = {a: getattr(net, a).output_qparams() for a in wl.layer_names(name)}
    entry = wl.NETWORKS[name]
    want = pipeline.forward(entry, x_np, pipeline.quantize_layers(entry, sd), qp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_graph_holds_its_blocks_and_returns_them(env):
    cx, i8ie, wl, GraphedForward = env
    net = wl.calibrated("two_conv", wl.synthetic_state_dict("two_conv", seed=1))
    x = i8ie.tensor(wl.synthetic_input("two_conv", 8, seed=2)).prefetch()
    cx.synchronize()
    live0 = cx.memory_stats()[0]
    g = GraphedForward(net, x)
    live1 = cx.memory_stats()[0]
    assert live1 > live0  # intermediates freed during the capture stay owned by the graph
    y = g().numpy()
    del g
    import gc
    gc.collect()
    assert cx.memory_stats()[0] <= live0 + y.nbytes + 4096


def test_capture_refuses_profiling_and_nested_capture(env):
    cx, i8ie, wl, GraphedForward = env
    cx.profile_start()
    try:
        with pytest.raises(Exception):
            cx.graph_begin()
    finally:
        cx.profile_stop()
    cx.graph_begin()
    try:
        with pytest.raises(Exception):
            cx.graph_begin()
    finally:
        cx.graph_end()
