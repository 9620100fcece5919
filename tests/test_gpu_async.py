"""Asynchronous transfers: pinned host memory, the transfer
stream, events -- first through the C-ABI with ctypes, then through the `i8ie` surface
(`numpy_async`, `pinned_empty`).  The reference has no counterpart (its tensors are host
arrays), so the checks are the obvious ones: the bytes that arrive are the bytes sent, the
ordering primitives order, and results equal the synchronous path / the oracle."""
import ctypes as C

import numpy as np
import pytest

import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import int8inferenceengine_amd  # noqa: F401  (torch's HIP runtime first)

    c = abi.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def i8ie():
    import int8inferenceengine_amd  # noqa: F401
    import i8ie as mod

    return mod


def _pinned(ctx, nbytes):
    p = C.c_void_p()
    abi.ck(abi.lib().i8ie_host_malloc(ctx.h, C.c_size_t(nbytes), C.byref(p)))
    return p


def _view(p, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.frombuffer((C.c_char * n).from_address(p.value), dtype=dtype).reshape(shape)


def test_abi_upload_on_transfer_stream_then_quantize_then_async_readback(ctx):
    import orc

    L = abi.lib()
    n = 1 << 20
    x = np.random.default_rng(0).uniform(-3, 3, n).astype(np.float32)
    src = _pinned(ctx, n * 4)
    dst = _pinned(ctx, n)
    _view(src, (n,), np.float32)[:] = x
    dx, dq = ctx.empty((n,), np.float32), ctx.empty((n,), np.uint8)
    up, done = C.c_void_p(), C.c_void_p()
    abi.ck(L.i8ie_event_create(ctx.h, C.byref(up)))
    abi.ck(L.i8ie_event_create(ctx.h, C.byref(done)))
    # upload on the transfer stream; the compute stream waits for it on the device
    abi.ck(L.i8ie_memcpy_h2d_async(ctx.h, dx.ptr, src, C.c_size_t(n * 4), 1))
    abi.ck(L.i8ie_event_record(ctx.h, up, 1))
    abi.ck(L.i8ie_stream_wait_event(ctx.h, up, 0))
    abi.ck(L.i8ie_quantize_f32_u8(ctx.h, dx.ptr, dq.ptr, C.c_int64(n), C.c_float(0.05), C.c_uint8(117)))
    abi.ck(L.i8ie_memcpy_d2h_async(ctx.h, dst, dq.ptr, C.c_size_t(n), 0))
    abi.ck(L.i8ie_event_record(ctx.h, done, 0))
    abi.ck(L.i8ie_event_synchronize(done))
    flag = C.c_int(0)
    abi.ck(L.i8ie_event_query(done, C.byref(flag)))
    assert flag.value == 1
    got = _view(dst, (n,), np.uint8).copy()
    assert np.array_equal(got, orc.quantize(x, 0.05, 117))
    for e in (up, done):
        abi.ck(L.i8ie_event_destroy(e))
    for p in (src, dst):
        abi.ck(L.i8ie_host_free(ctx.h, p))
    dx.free()
    dq.free()


def test_abi_async_copy_refuses_pageable_memory(ctx):
    L = abi.lib()
    d = ctx.empty((1024,), np.uint8)
    pageable = np.zeros(1024, np.uint8)
    rc = L.i8ie_memcpy_h2d_async(ctx.h, d.ptr, pageable.ctypes.data_as(C.c_void_p), C.c_size_t(1024), 0)
    assert rc == -1 and b"i8ie_host_malloc" in L.i8ie_last_error()
    rc = L.i8ie_memcpy_d2h_async(ctx.h, pageable.ctypes.data_as(C.c_void_p), d.ptr, C.c_size_t(1024), 0)
    assert rc == -1
    yes = C.c_int(7)
    abi.ck(L.i8ie_host_is_pinned(ctx.h, pageable.ctypes.data_as(C.c_void_p), C.c_size_t(1024), C.byref(yes)))
    assert yes.value == 0
    p = _pinned(ctx, 4096)
    abi.ck(L.i8ie_host_is_pinned(ctx.h, C.c_void_p(p.value + 100), C.c_size_t(3996), C.byref(yes)))
    assert yes.value == 1
    abi.ck(L.i8ie_host_is_pinned(ctx.h, C.c_void_p(p.value + 100), C.c_size_t(3997), C.byref(yes)))
    assert yes.value == 0  # runs past the end of the block
    assert L.i8ie_host_free(ctx.h, C.c_void_p(p.value + 8)) == -1  # not a block start
    abi.ck(L.i8ie_host_free(ctx.h, p))
    d.free()


def test_numpy_async_equals_numpy(i8ie):
    x = np.random.default_rng(1).uniform(-2, 2, (64, 3, 17, 19)).astype(np.float32)
    q = i8ie.quantize(i8ie.tensor(x), 0.02, 100)
    r = i8ie.relu(q)
    fut = r.numpy_async()
    got = fut.result()
    assert fut.done()
    assert got.dtype == np.uint8 and got.shape == x.shape
    assert np.array_equal(got, r.numpy())
    assert fut.result() is got  # idempotent
    # a host-side tensor resolves at once
    small = i8ie.tensor(np.arange(6, dtype=np.float32).reshape(2, 3))
    assert np.array_equal(small.numpy_async().result(), np.arange(6, dtype=np.float32).reshape(2, 3))


def test_pinned_upload_matches_pageable_and_survives_refill(i8ie):
    import orc

    rng = np.random.default_rng(2)
    a = i8ie.pinned_empty((8, 3, 32, 32))
    assert a.dtype == np.float32 and a.shape == (8, 3, 32, 32)
    x0 = rng.uniform(-2, 2, a.shape).astype(np.float32)
    x1 = rng.uniform(-2, 2, a.shape).astype(np.float32)
    a[...] = x0
    t0 = i8ie.tensor(a).wait_upload()
    a[...] = x1  # refill after wait_upload: t0 keeps batch 0
    t1 = i8ie.tensor(a[2:6])  # a contiguous slice of a pinned array uploads asynchronously too
    assert np.array_equal(i8ie.quantize(t0, 0.02, 90).numpy(), orc.quantize(x0, 0.02, 90))
    assert np.array_equal(i8ie.quantize(t1, 0.02, 90).numpy(), orc.quantize(x1[2:6], 0.02, 90))
    assert np.array_equal(t0.numpy(), x0)
    # blocks cycle through the upload cache: many generations, same answers
    for g in range(6):
        a[...] = x0 + g
        t = i8ie.tensor(a)
        got = i8ie.quantize(t, 0.05, 64).numpy()
        assert np.array_equal(got, orc.quantize(x0 + np.float32(g), 0.05, 64))
        t.wait_upload()


def test_pipelined_batches_equal_synchronous_batches(i8ie):
    """The bench's depth-2 pipeline: launch batch i+1 (upload + kernels) before reading batch i."""
    from int8inferenceengine_amd import workloads as wl

    net = wl.calibrated("simple_conv", wl.synthetic_state_dict("simple_conv", seed=5))
    xs = [wl.synthetic_input("simple_conv", 16, seed=100 + i) for i in range(5)]
    want = [net(i8ie.tensor(x)).numpy() for x in xs]
    pin = [i8ie.pinned_empty(xs[0].shape) for _ in range(2)]
    got, pending = [], None
    held = [None, None]
    for i, x in enumerate(xs):
        if held[i & 1] is not None:
            held[i & 1].wait_upload()  # the buffer is being refilled: its last upload must be over
        pin[i & 1][...] = x
        t = i8ie.tensor(pin[i & 1])
        held[i & 1] = t
        fut = net(t).numpy_async()
        if pending is not None:
            got.append(pending.result())
        pending = fut
    got.append(pending.result())
    for w, g in zip(want, got):
        assert np.array_equal(w.view(np.uint32), g.view(np.uint32))


def test_zero_copy_torch_interop(i8ie):
    """a torch CUDA tensor enters the engine without a copy (`from_torch`), an engine
    tensor leaves it without one (`__cuda_array_interface__`)."""
    import orc
    import torch

    x = torch.empty((6, 3, 20, 20), dtype=torch.float32, device="cuda").uniform_(-2, 2)
    t = i8ie.from_torch(x)
    assert t.data.data_ptr() == x.data_ptr()                       # same memory
    q = i8ie.quantize(t, 0.02, 99)
    assert np.array_equal(q.numpy(), orc.quantize(x.cpu().numpy(), 0.02, 99))
    v = t.reshape(6, -1)                                           # views keep the owner alive
    del t, x
    torch.cuda.empty_cache()
    assert v.numpy().shape == (6, 1200)
    # export: u8 and f32 results as torch tensors on the same memory
    qt = torch.as_tensor(q, device="cuda")
    assert qt.dtype == torch.uint8 and tuple(qt.shape) == (6, 3, 20, 20)
    assert qt.data_ptr() == q.data.data_ptr()
    assert np.array_equal(qt.cpu().numpy(), q.numpy())
    d = i8ie.dequantize(q)
    dt = torch.as_tensor(d, device="cuda")
    assert dt.dtype == torch.float32 and np.array_equal(dt.cpu().numpy(), d.numpy())
    with pytest.raises(TypeError):
        i8ie.from_torch(torch.zeros(4, dtype=torch.float64, device="cuda"))
    with pytest.raises(TypeError):
        i8ie.from_torch(torch.zeros(4))
