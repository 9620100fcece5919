// pybind_module.cc -- the rebuilt extension module `_CXX_i8ie`.
//
// Same Python-visible surface as the reference's module (src/pybind11.cc:37-55,
// declare_linear src/fully_connected.cc:54-72, declare_conv2d
// src/conv2d.cc:144-165, declare_tensor_funcs src/functional.cc:66-82), but
// every tensor lives in MI355X HBM and every op is a HIP kernel reached through
// the C-ABI of libi8ie_hip.so (include/i8ie_hip.h).  This file contains no HIP
// code and no arithmetic of the hot path: only ownership, shapes, the
// prepare/convert state machine (src/layer.cc:28-54) and the calibrator
// (src/calibrator.cc).  There is no CPU fallback: if the library or a GPU is
// missing, calls raise RuntimeError.
//
// Additive API (not in the reference; needed for parity tests and benchmarks):
//   layer.set_output_qparams(scale, zp) / output_qparams() / q_weight() /
//   q_bias() / weight_scale() / forward_debug(u8 tensor) -> (out, int32 acc);
//   tensor.prefetch();  module: set_device, device, synchronize, memory_stats,
//   set_calibration_seed, abi_version.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "i8ie_hip.h"
#include "pybind11/numpy.h"
#include "pybind11/pybind11.h"
#include "pybind11/stl.h"

namespace py = pybind11;
using u8_t = unsigned char;
using s8_t = signed char;

namespace {

// ---------------------------------------------------------------- runtime ----
struct Runtime {
  i8ie_ctx* ctx = nullptr;
  int device = -1;
  long calib_seed = -1;  // < 0: std::random_device, as the reference (src/calibrator.cc:9-10)
  int calib_mode = 0;    // 0 auto: seeded -> the reference's mt19937 stream replayed on the host (golden-pinned),
                         // unseeded -> sampled on the device (no D2H of the layer outputs); 1 host; 2 device
};
Runtime& rt() {
  static Runtime* r = new Runtime();  // intentionally leaked: outlives every tensor at exit
  return *r;
}
void check(int rc) {
  if (rc != I8IE_OK) throw std::runtime_error(std::string("i8ie: ") + i8ie_last_error());
}
int default_device() {
  if (const char* e = std::getenv("I8IE_DEVICE")) return std::atoi(e);
  if (const char* e = std::getenv("LOCAL_RANK")) return std::atoi(e);  // one process per GPU
  return 0;
}
i8ie_ctx* ctx() {
  Runtime& r = rt();
  if (!r.ctx) {
    if (r.device < 0) r.device = default_device();
    check(i8ie_ctx_create(r.device, &r.ctx));
  }
  return r.ctx;
}

// ---------------------------------------------------------------- storage ----
// One buffer shared by a tensor and its reshape views (the role of the
// py::capsule in include/tensor.h:28,94-104).  A host mirror is kept for
// tensors that came from numpy so that small host-side round trips
// (argmax / == / sum in i8ie/__init__.py) do not touch the device.
struct Storage;
// Bordered NHWC buffers whose border bytes already hold the zero point, keyed by geometry: a network that
// runs the same shapes again gets them back and skips the border fill (kernels only write interiors).
struct BorderKey {
  size_t bytes;
  int n, c, h, w, b, zp;
  bool operator<(const BorderKey& o) const {
    return std::tie(bytes, n, c, h, w, b, zp) < std::tie(o.bytes, o.n, o.c, o.h, o.w, o.b, o.zp);
  }
};
std::map<BorderKey, std::vector<void*>>& border_cache() {
  static auto* m = new std::map<BorderKey, std::vector<void*>>();
  return *m;
}

// ---- pinned host staging, events and upload blocks (asynchronous transfers) ----
struct PinnedPool {
  std::map<size_t, std::vector<void*>> free_blocks;  // by capacity (multiples of 4 KiB)
  std::vector<i8ie_event*> events;
  size_t user_blocks = 0;  // live pinned_empty() arrays
};
PinnedPool& ppool() {
  static auto* p = new PinnedPool();
  return *p;
}
size_t pinned_capacity(size_t bytes) { return (bytes + 4095) & ~(size_t)4095; }
void* pinned_take(size_t cap) {
  auto& v = ppool().free_blocks[cap];
  if (!v.empty()) {
    void* p = v.back();
    v.pop_back();
    return p;
  }
  void* p = nullptr;
  check(i8ie_host_malloc(ctx(), cap, &p));
  return p;
}
void pinned_put(void* p, size_t cap) {
  auto& v = ppool().free_blocks[cap];
  if (v.size() < 8) v.push_back(p);
  else i8ie_host_free(ctx(), p);
}
i8ie_event* event_take() {
  auto& v = ppool().events;
  if (!v.empty()) {
    i8ie_event* e = v.back();
    v.pop_back();
    return e;
  }
  i8ie_event* e = nullptr;
  check(i8ie_event_create(ctx(), &e));
  return e;
}
void event_put(i8ie_event* e) { ppool().events.push_back(e); }
// Device blocks written by the transfer stream.  They bypass the stream-ordered allocator (whose reuse
// rule only covers the compute stream): a block comes back with an event recorded behind its last
// compute-stream reader, and the next upload into it makes the transfer stream wait for that event.
struct UploadSlot {
  void* dev;
  i8ie_event* free_ev;
};
std::map<size_t, std::vector<UploadSlot>>& upload_cache() {
  static auto* m = new std::map<size_t, std::vector<UploadSlot>>();
  return *m;
}

struct Storage {
  size_t bytes = 0;
  void* dev = nullptr;
  i8ie_event* ready_ev = nullptr;  // recorded on the transfer stream behind the upload into `dev`
  bool upload_block = false;       // `dev` belongs to upload_cache()
  bool external = false;           // `dev` is foreign HIP memory (e.g. a torch tensor's): never freed here
  py::object keep;                 // the pinned ndarray an asynchronous upload reads from / the foreign owner
  int bzp = -1;  // >= 0: bordered NHWC buffer with border bytes == bzp, the physical byte (returns to border_cache)
  std::vector<unsigned char> host;
  bool host_valid = false;
  // physical layout of a 4-D u8 activation: the engine keeps NHWC between layers and
  // converts back to the reference's NCHW only when the bytes are observed
  int layout = I8IE_LAYOUT_NCHW;
  int border = 0;                      // NHWC only: physical border (pixels) holding the zero point
  bool s8 = false;                     // NHWC only: bytes stored re-biased (^0x80, I8IE_LAYOUT_NHWC_S8; border: zp ^ 0x80)
  int dn = 0, dc = 0, dh = 0, dw = 0;  // logical NCHW dims, valid when layout == NHWC
  void set_nhwc(const std::vector<ssize_t>& shp, int b) {
    layout = I8IE_LAYOUT_NHWC;
    border = b;
    dn = (int)shp[0]; dc = (int)shp[1]; dh = (int)shp[2]; dw = (int)shp[3];
  }
  void unbias() {  // re-biased bytes -> plain u8, in place (border bytes become the zero point again)
    if (!s8 || !dev) return;
    check(i8ie_rebias_u8(ctx(), (const uint8_t*)dev, (uint8_t*)dev, (int64_t)bytes));
    s8 = false;
    if (bzp >= 0) bzp ^= 0x80;
  }
  void to_nchw() {
    if (layout == I8IE_LAYOUT_NCHW || external) return;
    unbias();
    void* fresh = nullptr;
    const size_t logical = (size_t)dn * dc * dh * dw;
    check(i8ie_malloc(ctx(), logical, &fresh));
    check(i8ie_layout_convert_u8(ctx(), (const uint8_t*)dev, (uint8_t*)fresh, dn, dc, dh, dw, 0, border, 0));
    i8ie_free(ctx(), dev);  // stream-ordered: the conversion above still reads it, reuse is queued behind
    dev = fresh;
    bytes = logical;
    layout = I8IE_LAYOUT_NCHW;
    border = 0;
    bzp = -1;
  }
  void wait_upload_on_stream() {  // the compute stream waits (device side) for the upload
    if (!ready_ev) return;
    check(i8ie_stream_wait_event(ctx(), ready_ev, 0));
    event_put(ready_ev);
    ready_ev = nullptr;
    keep = py::object();
  }
  void wait_upload_on_host() {
    if (!ready_ev) return;
    {
      py::gil_scoped_release nogil;
      check(i8ie_event_synchronize(ready_ev));
    }
    keep = py::object();  // the source may be refilled; the compute stream still has to wait on the event
  }
  ~Storage() {
    if (external) return;  // the owner object in `keep` frees it
    if (!dev || !rt().ctx) return;
    if (ready_ev) {  // never consumed: let the copy finish before the block is reused
      i8ie_stream_wait_event(rt().ctx, ready_ev, 0);
      event_put(ready_ev);
      ready_ev = nullptr;
    }
    int capturing = 0;  // a graph being captured has this address baked in: the block goes to the graph
    i8ie_ctx_is_capturing(rt().ctx, &capturing);  // (i8ie_free hands it over), never to a cache of ours
    if (upload_block && !capturing) {
      auto& slot = upload_cache()[bytes];
      if (slot.size() < 4) {
        i8ie_event* e = event_take();
        i8ie_event_record(rt().ctx, e, 0);
        slot.push_back(UploadSlot{dev, e});
        return;
      }
    }
    if (bzp >= 0 && layout == I8IE_LAYOUT_NHWC && border > 0 && !capturing) {
      auto& slot = border_cache()[BorderKey{bytes, dn, dc, dh, dw, border, bzp}];
      if (slot.size() < 4) {
        slot.push_back(dev);
        return;
      }
    }
    i8ie_free(rt().ctx, dev);
  }
  void* device_ptr() {
    if (ready_ev) wait_upload_on_stream();
    if (!dev) {
      check(i8ie_malloc(ctx(), bytes, &dev));
      if (host_valid) {
        py::gil_scoped_release nogil;
        check(i8ie_memcpy_h2d(ctx(), dev, host.data(), bytes));
      }
      if (bytes > (1u << 20)) {  // keep the mirror only for small tensors
        std::vector<unsigned char>().swap(host);
        host_valid = false;
      }
    }
    return dev;
  }
  void read_back(void* dst) {
    if (ready_ev) wait_upload_on_stream();
    if (host_valid) {
      std::memcpy(dst, host.data(), bytes);
      return;
    }
    py::gil_scoped_release nogil;
    check(i8ie_memcpy_d2h(ctx(), dst, dev, bytes));
  }
};

std::shared_ptr<Storage> device_storage(size_t bytes) {
  auto s = std::make_shared<Storage>();
  s->bytes = bytes;
  check(i8ie_malloc(ctx(), bytes, &s->dev));
  return s;
}

// NHWC u8 storage for logical shape `shp` (NCHW order) with a border of b pixels holding zp (s8: the bytes of this
// tensor are stored re-biased, I8IE_LAYOUT_NHWC_S8, so its border holds zp ^ 0x80)
std::shared_ptr<Storage> nhwc_storage(const std::vector<ssize_t>& shp, int b, int zp, bool s8 = false) {
  const size_t bytes = (size_t)shp[0] * shp[1] * (shp[2] + 2 * b) * (shp[3] + 2 * b);
  if (s8) zp ^= 0x80;
  if (b <= 0) {
    auto s = device_storage(bytes);
    s->set_nhwc(shp, 0);
    s->s8 = s8;
    return s;
  }
  auto s = std::make_shared<Storage>();
  s->bytes = bytes;
  s->set_nhwc(shp, b);
  s->s8 = s8;
  s->bzp = zp;
  auto it = border_cache().find(BorderKey{bytes, s->dn, s->dc, s->dh, s->dw, b, zp});
  if (it != border_cache().end() && !it->second.empty()) {
    s->dev = it->second.back();  // border still valid from its previous life
    it->second.pop_back();
    return s;
  }
  check(i8ie_malloc(ctx(), bytes, &s->dev));
  check(i8ie_fill_border_u8(ctx(), (uint8_t*)s->dev, s->dn, s->dc, s->dh, s->dw, b, (uint8_t)zp));
  return s;
}

// A recorded (not yet launched) operation.  Every holder -- python-side copies of the tensor, relu(y) made from y,
// the closures of consumers that captured their input -- shares ONE node, and the node remembers what it has
// produced: a producer observed twice, or consumed by two ops, launches once per distinct (relu, border) request,
// and its upstream chain (whose nodes cache in the same way) is never launched again.
struct PendNode {
  // arguments: fuse relu; physical border wanted by the consumer; consumer reads re-biased bytes (both honoured for NHWC
  // results of layers whose kernel can; what came out is in the Storage)
  std::function<std::shared_ptr<Storage>(bool, int, bool)> fn;
  // a conv layer whose kernel can fold a following max_pool2d into its epilogue (i8ie_layer_fuses_pool): arguments
  // relu, border of the POOLED result, pool kernel_size, stride; returns null when this launch cannot (the caller
  // then pools the unfused result)
  std::function<std::shared_ptr<Storage>(bool, int, bool, int, int)> fn_pool;
  struct Made {
    bool relu;
    int border;
    std::shared_ptr<Storage> st;
  };
  std::vector<Made> made;
  std::shared_ptr<Storage> get(bool relu, int border, bool s8 = false) {
    for (const Made& m : made)
      if (m.relu == relu && m.border == border && m.st) {
        if (m.st->s8 && !s8) continue;  // (a plain reader does not take the re-biased copy)
        // a bordered NHWC result that was since converted in place (to_nchw) no longer has the border asked for
        if (border > 0 && (m.st->layout != I8IE_LAYOUT_NHWC || m.st->border != border)) continue;
        return m.st;
      }
    std::shared_ptr<Storage> st = fn(relu, border, s8);
    made.push_back(Made{relu, border, st});
    return st;
  }
};
template <typename F>
std::shared_ptr<PendNode> make_pend(F&& f) {
  auto n = std::make_shared<PendNode>();
  n->fn = std::forward<F>(f);
  return n;
}

// ----------------------------------------------------------------- tensor ----
template <typename T>
struct Tensor {
  std::shared_ptr<Storage> st;
  std::vector<ssize_t> shape;
  ssize_t size = 0;
  float scale = 1;      // include/tensor.h:153
  u8_t zero_point = 0;  // include/tensor.h:154
  // A layer forward that has been recorded but not launched yet: lets a following relu
  // fold into the epilogue.  Launching is observationally identical to the eager call.
  // arguments: fuse relu; physical border wanted by the consumer (honoured for NHWC results)
  std::shared_ptr<PendNode> pend;
  bool pend_relu = false;
  // a pending small Linear layer can also produce dequantize(layer(x)) directly (argument: fuse relu)
  std::shared_ptr<std::function<std::shared_ptr<Storage>(bool)>> pend_f32;
  // set while this tensor is a not-yet-launched quantize(x, scale, zp): lets the first conv read x directly
  std::shared_ptr<Storage> qsrc;
  float qscale = 0;
  u8_t qzp = 0;

  void realize(int border = 0, bool s8 = false) {
    if (!pend) return;
    st = pend->get(pend_relu, border, s8);
    pend.reset();
    pend_f32.reset();
    qsrc.reset();
  }

  Tensor() = default;
  explicit Tensor(std::vector<ssize_t> shp) : shape(std::move(shp)) {
    size = 1;
    for (ssize_t d : shape) size *= d;
    st = device_storage((size_t)size * sizeof(T));
  }
  T* dptr() {  // NCHW / row-major bytes, as the reference lays them out
    realize();
    if (!st) throw std::runtime_error("i8ie: empty tensor");
    st->to_nchw();
    return static_cast<T*>(st->device_ptr());
  }
  T* dptr_any() {  // whatever layout the storage is in (see st->layout), plain bytes
    realize();
    if (!st) throw std::runtime_error("i8ie: empty tensor");
    T* p = static_cast<T*>(st->device_ptr());
    st->unbias();
    return p;
  }
  T* dptr_raw() {  // ... as it is, re-biased or not (st->s8): for consumers that take I8IE_LAYOUT_NHWC_S8
    if (!st) throw std::runtime_error("i8ie: empty tensor");
    return static_cast<T*>(st->device_ptr());
  }
  py::array_t<T> numpy() {
    realize();
    py::array_t<T> out(shape);
    if (st && size > 0) {
      st->to_nchw();
      st->read_back(out.mutable_data());
    }
    return out;
  }
  struct HostFuture;
  std::shared_ptr<HostFuture> numpy_async();
  // include/tensor.h:106-133: at most one -1, no zeros, sizes must match
  Tensor<T> reshape(std::vector<ssize_t> shp) {
    realize();
    // Views are defined on the reference's element order, so the buffer goes back to NCHW here -- except for the
    // flatten x.reshape(n, -1) of a border-free NHWC activation: that view is converted only if its bytes are
    // observed (dptr() / numpy()); a Linear layer reads it as it lies (K walked in (h, w, c) order).  Any other
    // new shape would disagree with the storage's logical dims for the layout-aware consumers.
    const bool flatten = st && st->layout == I8IE_LAYOUT_NHWC && st->border == 0 && shp.size() == 2 &&
                         (shp[0] == st->dn || (shp[0] < 0 && shp[1] == (ssize_t)st->dc * st->dh * st->dw) ||
                          (shp[1] < 0 && shp[0] == st->dn));
    if (st && !flatten) st->to_nchw();
    ssize_t midx = -1, sz = 1;
    for (size_t i = 0; i < shp.size(); ++i) {
      if (shp[i] < 0) {
        if (midx != -1) throw std::runtime_error("i8ie: reshape: more than one negative dimension");
        midx = (ssize_t)i;
      } else if (shp[i] == 0) {
        throw std::runtime_error("i8ie: reshape: zero dimension");
      } else {
        sz *= shp[i];
      }
    }
    if (midx >= 0) {
      if (size % sz != 0) throw std::runtime_error("i8ie: reshape: size is not divisible");
      shp[midx] = size / sz;
      sz *= shp[midx];
    }
    if (sz != size) throw std::runtime_error("i8ie: reshape: element count differs");
    Tensor<T> v;
    v.st = st;
    v.shape = std::move(shp);
    v.size = size;
    v.scale = scale;
    v.zero_point = zero_point;
    return v;
  }
};

// numpy() split in two: the device->host copy is queued behind the tensor's kernels into pinned staging
// memory and result() waits for it, so the caller can launch the next batch in between.
template <typename T>
struct Tensor<T>::HostFuture {
  std::vector<ssize_t> shape;
  size_t bytes = 0, cap = 0;
  void* buf = nullptr;
  i8ie_event* ev = nullptr;
  py::object ready;  // already on the host
  HostFuture() = default;
  HostFuture(const HostFuture&) = delete;
  HostFuture& operator=(const HostFuture&) = delete;
  bool done() {
    if (!ev) return true;
    int d = 0;
    check(i8ie_event_query(ev, &d));
    return d != 0;
  }
  py::object result() {
    if (ready) return ready;
    if (ev) {
      py::gil_scoped_release nogil;
      check(i8ie_event_synchronize(ev));
    }
    py::array_t<T> out(shape);
    if (bytes) std::memcpy(out.mutable_data(), buf, bytes);
    release();
    ready = out;
    return ready;
  }
  void release() {
    if (ev) event_put(ev);
    if (buf) pinned_put(buf, cap);
    ev = nullptr;
    buf = nullptr;
  }
  ~HostFuture() {
    if (!rt().ctx) return;
    if (ev) i8ie_event_synchronize(ev);  // the copy may still be writing the staging block
    release();
  }
};
template <typename T>
std::shared_ptr<typename Tensor<T>::HostFuture> Tensor<T>::numpy_async() {
  realize();
  auto f = std::make_shared<HostFuture>();
  f->shape = shape;
  if (!st || size == 0 || st->host_valid) {
    f->ready = numpy();
    return f;
  }
  st->to_nchw();
  f->bytes = (size_t)size * sizeof(T);
  f->cap = pinned_capacity(f->bytes);
  f->buf = pinned_take(f->cap);
  check(i8ie_memcpy_d2h_async(ctx(), f->buf, st->device_ptr(), f->bytes, 0));
  f->ev = event_take();
  check(i8ie_event_record(ctx(), f->ev, 0));
  return f;
}

// float32 ndarray in pinned host memory: i8ie.tensor() of it (or of a slice of it) uploads asynchronously
py::array_t<float> pinned_empty(std::vector<ssize_t> shape) {
  size_t n = 1;
  for (ssize_t d : shape) {
    if (d < 0) throw std::runtime_error("i8ie: pinned_empty: negative dimension");
    n *= (size_t)d;
  }
  void* p = nullptr;
  check(i8ie_host_malloc(ctx(), std::max<size_t>(n * sizeof(float), 4), &p));
  ppool().user_blocks += 1;
  py::capsule owner(p, [](void* q) {
    if (rt().ctx) i8ie_host_free(rt().ctx, q);
    ppool().user_blocks -= 1;
  });
  return py::array_t<float>(shape, static_cast<float*>(p), owner);
}

Tensor<float> tensor_from_numpy(py::array_t<float, py::array::c_style | py::array::forcecast> a) {
  Tensor<float> t;  // include/tensor.h:40-47: copies the ndarray
  t.shape.assign(a.shape(), a.shape() + a.ndim());
  t.size = a.size();
  t.st = std::make_shared<Storage>();
  t.st->bytes = (size_t)t.size * sizeof(float);
  int pinned = 0;
  if (ppool().user_blocks > 0 && t.st->bytes > 0)
    check(i8ie_host_is_pinned(ctx(), a.data(), t.st->bytes, &pinned));
  if (pinned) {
    // The copy the reference makes at construction becomes an asynchronous upload on the transfer
    // stream; the ndarray must stay unchanged until wait_upload() (or the first read-back) returns.
    Storage& st = *t.st;
    auto& uc = upload_cache()[st.bytes];
    if (!uc.empty()) {
      UploadSlot slot = uc.back();
      uc.pop_back();
      check(i8ie_stream_wait_event(ctx(), slot.free_ev, 1));
      event_put(slot.free_ev);
      st.dev = slot.dev;
    } else {
      check(i8ie_malloc(ctx(), st.bytes, &st.dev));
      // a block fresh from the allocator may still be read by queued compute-stream work
      i8ie_event* e = event_take();
      check(i8ie_event_record(ctx(), e, 0));
      check(i8ie_stream_wait_event(ctx(), e, 1));
      event_put(e);
    }
    st.upload_block = true;
    st.keep = a;
    check(i8ie_memcpy_h2d_async(ctx(), st.dev, a.data(), st.bytes, 1));
    st.ready_ev = event_take();
    check(i8ie_event_record(ctx(), st.ready_ev, 1));
    return t;
  }
  t.st->host.resize(t.st->bytes);
  if (t.st->bytes) std::memcpy(t.st->host.data(), a.data(), t.st->bytes);
  t.st->host_valid = true;
  return t;
}

// Zero-copy view of foreign device memory (a torch tensor's data_ptr(), any hipMalloc block): the tensor reads it
// in place; `owner` is kept alive as long as the tensor (or a reshape view of it) lives.  The memory must be
// complete on this module's stream (share the stream with use_stream(), or synchronise the producer first).
Tensor<float> tensor_from_device(uintptr_t ptr, std::vector<ssize_t> shape, py::object owner) {
  if (ptr == 0 || (ptr & 3u) != 0) throw std::runtime_error("i8ie: tensor_from_device: null or misaligned pointer");
  Tensor<float> t;
  t.shape = std::move(shape);
  t.size = 1;
  for (ssize_t d : t.shape) {
    if (d <= 0) throw std::runtime_error("i8ie: tensor_from_device: non-positive dimension");
    t.size *= d;
  }
  (void)ctx();
  t.st = std::make_shared<Storage>();
  t.st->bytes = (size_t)t.size * sizeof(float);
  t.st->dev = reinterpret_cast<void*>(ptr);
  t.st->external = true;
  t.st->keep = std::move(owner);
  return t;
}

template <typename T>
void bind_tensor(py::module_& m, const char* name) {
  using Fut = typename Tensor<T>::HostFuture;
  py::class_<Fut, std::shared_ptr<Fut>>(m, (std::string(name) + "_HostFuture").c_str())
      .def("done", &Fut::done)
      .def("result", &Fut::result);
  py::class_<Tensor<T>>(m, name)
      .def(py::init<>())
      .def("numpy", &Tensor<T>::numpy)
      .def("numpy_async", &Tensor<T>::numpy_async)
      .def("wait_upload", [](Tensor<T>& t) { if (t.st) t.st->wait_upload_on_host(); })
      .def("zero_point", [](const Tensor<T>& t) { return t.zero_point; })
      .def("scale", [](const Tensor<T>& t) { return t.scale; })
      .def("sum",
           [](Tensor<T>& t) {  // src/pybind11.cc:18-25: sequential fp32 sum
             py::array_t<T> a = t.numpy();
             const T* p = a.data();
             float s = 0;
             for (ssize_t i = 0; i < t.size; ++i) s += p[i];
             return s;
           })
      .def("ref_count",
           [](Tensor<T>& t) {
             t.realize();
             return t.st ? (long)t.st.use_count() : 0L;
           })
      .def("reshape", [](Tensor<T>& t, std::vector<ssize_t> shape) { return t.reshape(std::move(shape)); })
      .def("layout", [](Tensor<T>& t) { t.realize(); return t.st ? t.st->layout : 0; })
      .def("prefetch", [](Tensor<T>& t) { (void)t.dptr(); })
      // device address of the (launched, NCHW-ordered) contents: for __cuda_array_interface__ exports
      .def("data_ptr", [](Tensor<T>& t) { return (uintptr_t)t.dptr(); })
      .def("shape", [](const Tensor<T>& t) { return t.shape; })
      .def("nbytes", [](const Tensor<T>& t) { return (size_t)t.size * sizeof(T); });
}

// ------------------------------------------------------------ elementwise ----
Tensor<u8_t> quantize(Tensor<float>& in, float scale, u8_t zp) {  // src/quantize_utils.cc:44-52
  Tensor<u8_t> out;
  out.shape = in.shape;
  out.size = in.size;
  out.scale = scale;
  out.zero_point = zp;
  (void)in.dptr();  // make sure the FP32 source is on the device
  std::shared_ptr<Storage> src = in.st;
  const ssize_t n = in.size;
  out.qsrc = src;
  out.qscale = scale;
  out.qzp = zp;
  // deferred: a first conv layer that accepts FP32 input consumes `src` directly (fused quantize)
  out.pend = make_pend(
      [src, n, scale, zp](bool relu, int, bool) {
        auto st = device_storage((size_t)n);
        check(i8ie_quantize_f32_u8(ctx(), (const float*)src->device_ptr(), (uint8_t*)st->dev, n, scale, zp));
        if (relu) check(i8ie_relu_u8(ctx(), (const uint8_t*)st->dev, (uint8_t*)st->dev, n, zp));
        return st;
      });
  return out;
}
Tensor<float> dequantize(Tensor<u8_t>& in) {  // src/quantize_utils.cc:54-58
  if (in.pend && in.pend_f32) {  // classifier head still pending: layer + (relu) + dequantize in one launch
    Tensor<float> out;
    out.shape = in.shape;
    out.size = in.size;
    out.st = (*in.pend_f32)(in.pend_relu);
    return out;
  }
  Tensor<float> out(in.shape);
  check(i8ie_dequantize_u8_f32(ctx(), in.dptr(), out.dptr(), in.size, in.scale, in.zero_point));
  return out;
}
Tensor<u8_t> relu_u8(Tensor<u8_t>& in) {  // src/functional.cc:15-26
  if (in.pend && !in.pend_relu) {
    // `in` is a layer output that has not been launched: record layer+relu as one launch.
    // `in` itself stays pending (if it is ever observed it launches without the relu).
    Tensor<u8_t> out;
    out.shape = in.shape;
    out.size = in.size;
    out.scale = in.scale;
    out.zero_point = in.zero_point;
    out.pend = in.pend;
    out.pend_f32 = in.pend_f32;
    out.pend_relu = true;  // (qsrc is not carried over: relu(quantize(x)) is not a plain quantize)
    return out;
  }
  const uint8_t* src = in.shape.size() == 4 ? in.dptr_any() : in.dptr();
  Tensor<u8_t> out(in.shape);
  out.scale = in.scale;
  out.zero_point = in.zero_point;
  if (in.st->layout == I8IE_LAYOUT_NHWC) {  // elementwise over the physical buffer (border bytes = zp stay zp)
    out.st = device_storage(in.st->bytes);
    out.st->set_nhwc(in.shape, in.st->border);
  }
  check(i8ie_relu_u8(ctx(), src, (uint8_t*)out.st->dev, (int64_t)in.st->bytes, in.zero_point));
  return out;
}
Tensor<float> relu_f32(Tensor<float>& in) {  // src/functional.cc:5-13
  Tensor<float> out(in.shape);
  check(i8ie_relu_f32(ctx(), in.dptr(), out.dptr(), in.size));
  return out;
}
template <typename T>
std::vector<ssize_t> pool_shape(const Tensor<T>& in, ssize_t k, ssize_t s) {
  if (in.shape.size() != 4) throw std::runtime_error("i8ie: max_pool2d expects an NCHW tensor");
  if (k <= 0 || s <= 0) throw std::runtime_error("i8ie: max_pool2d: kernel_size and stride must be positive");
  if (k > in.shape[2] || k > in.shape[3]) throw std::runtime_error("i8ie: max_pool2d: window larger than input");
  return {in.shape[0], in.shape[1], (in.shape[2] - k) / s + 1, (in.shape[3] - k) / s + 1};
}
Tensor<u8_t> max_pool2d_u8(Tensor<u8_t>& in, ssize_t k, ssize_t s) {  // src/functional.cc:36-64
  Tensor<u8_t> out;
  out.shape = pool_shape(in, k, s);
  out.size = 1;
  for (ssize_t d : out.shape) out.size *= d;
  out.scale = in.scale;
  out.zero_point = in.zero_point;
  Tensor<u8_t> src = in;
  const std::vector<ssize_t> ishp = in.shape, oshp = out.shape;
  const u8_t zp = in.zero_point;
  const int kk = (int)k, ss = (int)s;
  // deferred so that a consuming conv can ask for a zero-point border around the result
  out.pend = make_pend(
      [src, ishp, oshp, zp, kk, ss](bool relu, int border, bool s8) mutable {
        if (src.pend && src.pend->fn_pool && src.pend.use_count() == 1 && src.pend->made.empty()) {
          // relu(layer(x)) still pending, nobody else holds it (no second consumer, not observable any more), and its
          // kernel pools in the epilogue: one launch, no unpooled tensor.  With another holder the layer launches once,
          // unfused, and every consumer reads that result (a recorded launch never runs twice).
          // (a relu behind the pool folds in as well: max-pool and relu commute, both are monotone)
          std::shared_ptr<Storage> st = src.pend->fn_pool(src.pend_relu || relu, border, s8, kk, ss);
          if (st) return st;
        }
        const uint8_t* ip = src.dptr_any();
        std::shared_ptr<Storage> st;
        const size_t logical = (size_t)oshp[0] * oshp[1] * oshp[2] * oshp[3];
        if (src.st->layout == I8IE_LAYOUT_NHWC && ishp[1] % 16 == 0) {
          st = nhwc_storage(oshp, border, zp);
          check(i8ie_maxpool2d_u8_nhwc(ctx(), ip, src.st->border, (uint8_t*)st->dev, border, (int)ishp[0], (int)ishp[1],
                                       (int)ishp[2], (int)ishp[3], kk, ss));
        } else {
          st = device_storage(logical);
          check(i8ie_maxpool2d_u8(ctx(), src.dptr(), (uint8_t*)st->dev, (int)ishp[0], (int)ishp[1], (int)ishp[2],
                                  (int)ishp[3], kk, ss));
        }
        if (relu) check(i8ie_relu_u8(ctx(), (const uint8_t*)st->dev, (uint8_t*)st->dev, (int64_t)st->bytes, zp));
        return st;
      });
  return out;
}
// the s8 instantiations of the generic templates (src/functional.cc:5-13, 36-64, registered at :78-82)
Tensor<s8_t> relu_s8(Tensor<s8_t>& in) {
  if (!in.st) return Tensor<s8_t>();  // default-constructed: nothing to do
  Tensor<s8_t> out(in.shape);  // (the generic relu does not carry scale / zero point over: src/functional.cc:7)
  if (in.size > 0) check(i8ie_relu_s8(ctx(), (const int8_t*)in.dptr(), (int8_t*)out.dptr(), in.size));
  return out;
}
Tensor<s8_t> max_pool2d_s8(Tensor<s8_t>& in, ssize_t k, ssize_t s) {
  Tensor<s8_t> out(pool_shape(in, k, s));
  out.scale = in.scale;  // src/functional.cc:43-44
  out.zero_point = in.zero_point;
  check(i8ie_maxpool2d_s8(ctx(), (const int8_t*)in.dptr(), (int8_t*)out.dptr(), (int)in.shape[0], (int)in.shape[1],
                          (int)in.shape[2], (int)in.shape[3], (int)k, (int)s));
  return out;
}
Tensor<float> max_pool2d_f32(Tensor<float>& in, ssize_t k, ssize_t s) {
  Tensor<float> out(pool_shape(in, k, s));
  check(i8ie_maxpool2d_f32(ctx(), in.dptr(), out.dptr(), (int)in.shape[0], (int)in.shape[1], (int)in.shape[2],
                           (int)in.shape[3], (int)k, (int)s));
  return out;
}

// -------------------------------------------------------------- calibrator ----
// src/calibrator.cc:6-37, include/calibrator.h:4-14
constexpr ssize_t kNumSamples = 1000;
struct Calibrator {
  std::array<float, kNumSamples> samples{};  // value-initialised (make_unique<Calibrator>(), src/layer.cc:33)
  ssize_t cnt = 0;
  // device-side sampling (i8ie_calib_sample_f32): the 1000 slots live on the GPU until get_range()
  float* samples_dev = nullptr;
  int* scratch_dev = nullptr;
  int64_t seen = 0;
  uint64_t dev_seed = 0;
  Calibrator() = default;
  Calibrator(const Calibrator&) = delete;
  Calibrator& operator=(const Calibrator&) = delete;
  ~Calibrator() {
    if (samples_dev) i8ie_free(rt().ctx, samples_dev);
    if (scratch_dev) i8ie_free(rt().ctx, scratch_dev);
  }
  static bool on_device() { return rt().calib_mode == 2 || (rt().calib_mode == 0 && rt().calib_seed < 0); }
  void sample_device(const float* dptr, ssize_t n) {
    if (!samples_dev) {
      check(i8ie_malloc(rt().ctx, kNumSamples * sizeof(float), (void**)&samples_dev));
      check(i8ie_malloc(rt().ctx, kNumSamples * sizeof(int), (void**)&scratch_dev));
      check(i8ie_memset(rt().ctx, samples_dev, 0, kNumSamples * sizeof(float)));
      if (rt().calib_seed >= 0) {
        dev_seed = (uint64_t)rt().calib_seed;
      } else {
        std::random_device rd;
        dev_seed = ((uint64_t)rd() << 32) | rd();
      }
    }
    check(i8ie_calib_sample_f32(rt().ctx, dptr, (int64_t)n, seen, dev_seed, samples_dev, scratch_dev));
    seen += n;
    cnt = seen < kNumSamples ? (ssize_t)seen : kNumSamples;
  }
  void sample(const float* data, ssize_t n) {
    std::mt19937 rng;
    if (rt().calib_seed >= 0) {
      rng.seed((unsigned)rt().calib_seed);
    } else {
      std::random_device rd;
      rng.seed(rd());
    }
    std::uniform_int_distribution<ssize_t> dist(0, kNumSamples * 2);  // about half get sampled
    for (ssize_t i = 0; i < n; ++i) {
      if (cnt < kNumSamples) {
        samples[cnt++] = data[i];
      } else {
        ssize_t idx = dist(rng);
        if (idx < kNumSamples) samples[idx] = data[i];
      }
    }
  }
  std::tuple<float, u8_t> get_range(float quantile) {
    if (samples_dev) check(i8ie_memcpy_d2h(rt().ctx, samples.data(), samples_dev, kNumSamples * sizeof(float)));
    std::sort(samples.begin(), samples.end());
    float out_min = samples[(size_t)((1.0 - quantile) * cnt)];
    float out_max = samples[(size_t)(quantile * (cnt - 1))];
    out_min = std::fmin(out_min, 0.);
    out_max = std::fmax(out_max, 0.);
    u8_t zp = (u8_t)(255 * (0 - out_min) / (out_max - out_min + 1e-09));
    float scale = (zp == 0) ? (out_max - out_min) / 255 : (0 - out_min) / zp;
    if (scale == 0) scale = 1;
    return std::make_tuple(scale, zp);
  }
};

// ------------------------------------------------------------------ layers ----
class BaseLayer {
 public:
  BaseLayer(std::vector<ssize_t> wshape, ssize_t out_channel) : wshape_(std::move(wshape)) {
    ssize_t n = 1;
    for (ssize_t d : wshape_) n *= d;
    w_.assign((size_t)n, 0.0f);
    b_.assign((size_t)out_channel, 0.0f);
    has_fp32_ = true;
  }
  BaseLayer(py::array_t<float, py::array::c_style | py::array::forcecast> w,
            py::array_t<float, py::array::c_style | py::array::forcecast> b) {
    has_fp32_ = true;
    load_weight(w);
    load_bias(b);
  }
  virtual ~BaseLayer() { release_fp32_dev(); }
  BaseLayer(const BaseLayer&) = delete;
  BaseLayer& operator=(const BaseLayer&) = delete;

  void load_weight(py::array_t<float, py::array::c_style | py::array::forcecast> w) {  // include/layer.h:15-20
    if (!has_fp32_) throw std::runtime_error("i8ie: load_weight: layer is already converted");
    wshape_.assign(w.shape(), w.shape() + w.ndim());
    w_.assign(w.data(), w.data() + w.size());
    release_fp32_dev();
  }
  void load_bias(py::array_t<float, py::array::c_style | py::array::forcecast> b) {  // include/layer.h:21-26
    if (!has_fp32_) throw std::runtime_error("i8ie: load_bias: layer is already converted");
    b_.assign(b.data(), b.data() + b.size());
    release_fp32_dev();
  }
  void prepare() {  // src/layer.cc:28-35
    if (is_quantized_) {
      std::cerr << "already quantized" << std::endl;
      return;
    }
    cal_ = std::make_unique<Calibrator>();
    is_preparing_ = true;
  }
  void convert() {  // src/layer.cc:36-54
    if (is_quantized_) {
      std::cerr << "already quantized" << std::endl;
      return;
    }
    if (!is_preparing_) {
      if (!qparams_overridden_) std::cerr << "No prepared, use default config" << std::endl;
    } else {
      float s;
      u8_t z;
      std::tie(s, z) = cal_->get_range(1);
      if (!qparams_overridden_) {
        scale_ = s;
        zero_point_ = z;
      }
      cal_.reset();
    }
    check_shapes();
    const ssize_t n = (ssize_t)b_.size();
    qw_.resize(w_.size());
    qb_.resize(b_.size());
    check(i8ie_quantize_weight(w_.data(), (int64_t)w_.size(), b_.data(), (int64_t)b_.size(),
                               reinterpret_cast<int8_t*>(qw_.data()), reinterpret_cast<int8_t*>(qb_.data()),
                               &w_scale_));
    i8ie_layer* raw = make_handle(n);
    q_ = std::shared_ptr<i8ie_layer>(raw, [](i8ie_layer* l) { i8ie_layer_destroy(l); });
    check(i8ie_layer_set_output_qparams(q_.get(), scale_, zero_point_));
    is_preparing_ = false;
    is_quantized_ = true;
    std::vector<float>().swap(w_);  // src/layer.cc:52-53: FP32 weights are released
    std::vector<float>().swap(b_);
    has_fp32_ = false;
    release_fp32_dev();
  }
  void set_output_qparams(float s, int zp) {  // additive: inject what calibration would produce
    if (zp < 0 || zp > 255) throw std::runtime_error("i8ie: zero point must be in [0, 255]");
    scale_ = s;
    zero_point_ = (u8_t)zp;
    qparams_overridden_ = true;
    if (q_) check(i8ie_layer_set_output_qparams(q_.get(), scale_, zero_point_));
  }
  std::tuple<float, int> output_qparams() const { return std::make_tuple(scale_, (int)zero_point_); }
  // additive: restore a converted layer from saved INT8 weights (what convert() would have produced)
  void load_quantized(py::array_t<s8_t, py::array::c_style | py::array::forcecast> qw,
                      py::array_t<s8_t, py::array::c_style | py::array::forcecast> qb, float w_scale, float s_out,
                      int zp_out) {
    if (zp_out < 0 || zp_out > 255) throw std::runtime_error("i8ie: zero point must be in [0, 255]");
    std::vector<ssize_t> shp(qw.shape(), qw.shape() + qw.ndim());
    if (shp.size() != wshape_.size()) throw std::runtime_error("i8ie: load_quantized: weight rank mismatch");
    wshape_ = shp;
    if (qb.size() != wshape_[0]) throw std::runtime_error("i8ie: load_quantized: bias size mismatch");
    qw_.assign(qw.data(), qw.data() + qw.size());
    qb_.assign(qb.data(), qb.data() + qb.size());
    w_scale_ = w_scale;
    scale_ = s_out;
    zero_point_ = (u8_t)zp_out;
    qparams_overridden_ = true;
    i8ie_layer* raw = make_handle((ssize_t)qb_.size());
    q_ = std::shared_ptr<i8ie_layer>(raw, [](i8ie_layer* l) { i8ie_layer_destroy(l); });
    check(i8ie_layer_set_output_qparams(q_.get(), scale_, zero_point_));
    cal_.reset();
    is_preparing_ = false;
    is_quantized_ = true;
    std::vector<float>().swap(w_);
    std::vector<float>().swap(b_);
    has_fp32_ = false;
    release_fp32_dev();
  }
  py::array_t<s8_t> q_weight() const {
    need_quantized();
    py::array_t<s8_t> a(wshape_);
    std::memcpy(a.mutable_data(), qw_.data(), qw_.size());
    return a;
  }
  py::array_t<s8_t> q_bias() const {
    need_quantized();
    py::array_t<s8_t> a((ssize_t)qb_.size());
    std::memcpy(a.mutable_data(), qb_.data(), qb_.size());
    return a;
  }
  float weight_scale() const {
    need_quantized();
    return w_scale_;
  }
  bool is_quantized() const { return is_quantized_; }

 protected:
  virtual void check_shapes() const = 0;
  virtual i8ie_layer* make_handle(ssize_t n) = 0;
  // Deferred INT8 forward: returns a tensor whose launch happens when it is first needed.
  Tensor<u8_t> defer(Tensor<u8_t>& in, std::vector<ssize_t> oshape, int m, int h, int w, bool spatial) {
    Tensor<u8_t> out;
    out.shape = std::move(oshape);
    out.size = 1;
    for (ssize_t d : out.shape) out.size *= d;
    out.scale = scale_;
    out.zero_point = zero_point_;
    std::shared_ptr<i8ie_layer> handle = q_;
    Tensor<u8_t> src = in;  // shares the input's storage / pending launch
    const float s_in = in.scale;
    const u8_t zp_in = in.zero_point;
    const int zp_out = zero_point_;
    const std::vector<ssize_t> oshp = out.shape;
    const size_t obytes = (size_t)out.size;
    // One recorded conv / linear forward, with or without a max-pool folded in behind it (pk > 1).  Negotiates the
    // layouts at both ends with the library: a conv whose kernel reads re-biased bytes asks its producer for them, and
    // stores re-biased itself when its consumer asked and its kernel can (i8ie_layer_rebiased_io).
    auto run = [handle, s_in, zp_in, zp_out, m, spatial, oshp, obytes](Tensor<u8_t>& src, int h, int w, bool relu, int border, bool s8,
                                                                        int pk, int ps) -> std::shared_ptr<Storage> {
      int out_layout = I8IE_LAYOUT_NCHW, pad = 0;
      if (spatial) {
        check(i8ie_layer_preferred_layout(handle.get(), &out_layout));
        check(i8ie_layer_padding(handle.get(), &pad));
      }
      const bool pool = pk > 1 || (pk == 1 && ps > 1);  // (a subsampling 1 x 1 window is a pool: the library runs it unfused)
      if (pool && out_layout != I8IE_LAYOUT_NHWC) return nullptr;  // (the caller pools the unfused result)
      std::vector<ssize_t> rshp = oshp;
      if (pool) {
        rshp[2] = (oshp[2] - pk) / ps + 1;
        rshp[3] = (oshp[3] - pk) / ps + 1;
      }
      int reads = 0, stores = 0;
      if (spatial && out_layout == I8IE_LAYOUT_NHWC) check(i8ie_layer_rebiased_io(handle.get(), m, h, w, pool ? pk : 0, ps, &reads, &stores));
      const bool st_s8 = s8 && stores;
      if (spatial && out_layout == I8IE_LAYOUT_NHWC && src.pend && src.qsrc && !src.pend_relu) {
        int yes = 0;
        check(i8ie_layer_accepts_f32_input(handle.get(), h, w, &yes));
        if (yes) {  // quantize + conv (+ relu) (+ max-pool) in one kernel, reading the FP32 input
          auto st = nhwc_storage(rshp, border, zp_out, st_s8);
          check(i8ie_layer_forward_f32_input_pool(handle.get(), (const float*)src.qsrc->device_ptr(), m, h, w, src.qscale, src.qzp,
                                                  relu ? 1 : 0, pool ? pk : 0, ps, (uint8_t*)st->dev,
                                                  st_s8 ? I8IE_LAYOUT_NHWC_S8 : I8IE_LAYOUT_NHWC, border, nullptr));
          return st;
        }
      }
      const uint8_t* ip;
      if (spatial) {
        // ask a still-pending producer for a zero-point border that covers this conv's padding (and for re-biased bytes
        // when this layer's kernel reads them as they are)
        src.realize(out_layout == I8IE_LAYOUT_NHWC ? pad : 0, reads != 0);
        ip = (src.st && src.st->s8 && reads) ? src.dptr_raw() : src.dptr_any();
      } else {
        src.realize(0);
        Storage* s = src.st.get();
        // x.reshape(n, -1) of an NHWC activation: the Linear layer walks K in (h, w, c) order instead
        if (s && s->layout == I8IE_LAYOUT_NHWC && s->border == 0 && s->dn == m && s->dh * s->dw > 1 &&
            (ssize_t)s->dn * s->dc * s->dh * s->dw == src.size) {
          ip = src.dptr_any();
          h = s->dh;
          w = s->dw;
        } else {
          ip = src.dptr();
        }
      }
      const int in_layout = src.st->s8 ? I8IE_LAYOUT_NHWC_S8 : src.st->layout, in_border = src.st->border;
      const int ob = out_layout == I8IE_LAYOUT_NHWC ? border : 0;
      auto st = out_layout == I8IE_LAYOUT_NHWC ? nhwc_storage(rshp, ob, zp_out, st_s8) : device_storage(obytes);
      const int ol = st_s8 ? I8IE_LAYOUT_NHWC_S8 : out_layout;
      if (pool)
        check(i8ie_layer_forward_pool(handle.get(), ip, in_layout, in_border, m, h, w, s_in, zp_in, relu ? 1 : 0, pk, ps,
                                      (uint8_t*)st->dev, ol, ob, nullptr));
      else
        check(i8ie_layer_forward_fused(handle.get(), ip, in_layout, in_border, m, h, w, s_in, zp_in, relu ? 1 : 0,
                                       (uint8_t*)st->dev, ol, ob, nullptr));
      // (the input is released when the last tensor holding this closure lets go of it: right after the
      // launch in `x = relu(layer(x))`, later if the un-fused result is observed as well)
      return st;
    };
    out.pend = make_pend([run, src, h, w](bool relu, int border, bool s8) mutable { return run(src, h, w, relu, border, s8, 0, 0); });
    if (spatial) {
      // max_pool2d right behind this (relu'd) conv: one call of the library, which folds the pool into the convolution's
      // epilogue where a kernel of its can and runs the max-pool kernel behind it otherwise
      Tensor<u8_t> srcp = in;
      out.pend->fn_pool = [run, srcp, h, w](bool relu, int border, bool s8, int pk, int ps) mutable {
        return run(srcp, h, w, relu, border, s8, pk, ps);
      };
    }
    if (!spatial && out.shape.size() == 2 && out.shape[1] <= 16) {
      // dequantize(layer(x)) of a classifier head: one fused launch (i8ie_layer_forward_dequant)
      Tensor<u8_t> src2 = in;
      out.pend_f32 = std::make_shared<std::function<std::shared_ptr<Storage>(bool)>>(
          [handle, src2, s_in, zp_in, m, obytes](bool relu) mutable {
            src2.realize(0);
            Storage* s = src2.st.get();
            const uint8_t* ip;
            int lay = I8IE_LAYOUT_NCHW, hh = 0, ww = 0;
            if (s && s->layout == I8IE_LAYOUT_NHWC && s->border == 0 && s->dn == m && s->dh * s->dw > 1 &&
                (ssize_t)s->dn * s->dc * s->dh * s->dw == src2.size) {
              ip = src2.dptr_any();
              lay = I8IE_LAYOUT_NHWC;
              hh = s->dh;
              ww = s->dw;
            } else {
              ip = src2.dptr();
            }
            auto f = device_storage(obytes * sizeof(float));
            auto q = device_storage(obytes);  // u8 side output: only written when the fused kernel does not apply
            check(i8ie_layer_forward_dequant(handle.get(), ip, lay, m, hh, ww, s_in, zp_in, relu ? 1 : 0,
                                             (uint8_t*)q->dev, (float*)f->dev));
            return f;
          });
    }
    return out;
  }
  void need_quantized() const {
    if (!is_quantized_) throw std::runtime_error("i8ie: layer is not converted (call convert() first)");
  }
  void need_fp32() const {
    if (!has_fp32_)
      throw std::runtime_error("i8ie: FP32 input given to a converted layer (its FP32 weights were released)");
  }
  void upload_fp32() {
    if (w_dev_) return;
    check(i8ie_malloc(ctx(), w_.size() * 4, (void**)&w_dev_));
    check(i8ie_malloc(ctx(), b_.size() * 4, (void**)&b_dev_));
    check(i8ie_memcpy_h2d(ctx(), w_dev_, w_.data(), w_.size() * 4));
    check(i8ie_memcpy_h2d(ctx(), b_dev_, b_.data(), b_.size() * 4));
  }
  void release_fp32_dev() {
    if (rt().ctx) {
      if (w_dev_) i8ie_free(rt().ctx, w_dev_);
      if (b_dev_) i8ie_free(rt().ctx, b_dev_);
    }
    w_dev_ = b_dev_ = nullptr;
  }
  void maybe_sample(Tensor<float>& out) {  // src/conv2d.cc:94-96, src/fully_connected.cc:17-19
    if (!is_preparing_) return;
    if (Calibrator::on_device()) {  // no copy of the layer output to the host
      cal_->sample_device(out.dptr(), out.size);
      return;
    }
    py::array_t<float> a = out.numpy();
    cal_->sample(a.data(), out.size);
  }

  std::vector<ssize_t> wshape_;
  std::vector<float> w_, b_;
  bool has_fp32_ = false;
  float* w_dev_ = nullptr;
  float* b_dev_ = nullptr;
  std::vector<s8_t> qw_, qb_;
  float w_scale_ = 1;
  std::unique_ptr<Calibrator> cal_;
  bool is_preparing_ = false;
  bool is_quantized_ = false;
  bool qparams_overridden_ = false;
  float scale_ = 1;       // include/layer.h:46
  u8_t zero_point_ = 0;   // include/layer.h:47
  std::shared_ptr<i8ie_layer> q_;
};

class Linear : public BaseLayer {
 public:
  Linear(ssize_t in_channel, ssize_t out_channel) : BaseLayer({out_channel, in_channel}, out_channel) {}
  using BaseLayer::BaseLayer;

  Tensor<float> forward_f32(Tensor<float>& in) {  // src/fully_connected.cc:5-21
    need_fp32();
    check_shapes();
    const ssize_t n = wshape_[0], k = wshape_[1];
    if (in.shape.empty() || in.size % k != 0 || in.shape.back() != k)
      throw std::runtime_error("i8ie: Linear: input's last dimension must equal in_features");
    const ssize_t m = in.size / k;
    upload_fp32();
    Tensor<float> out({m, n});
    check(i8ie_linear_f32(ctx(), in.dptr(), (int)m, (int)k, w_dev_, b_dev_, (int)n, out.dptr()));
    maybe_sample(out);
    return out;
  }
  std::tuple<Tensor<u8_t>, py::object> forward_u8(Tensor<u8_t>& in, bool want_acc) {  // src/fully_connected.cc:22-52
    need_quantized();
    const ssize_t n = wshape_[0], k = wshape_[1];
    if (in.shape.empty() || in.shape.back() != k)
      throw std::runtime_error("i8ie: Linear: input's last dimension must equal in_features");
    const ssize_t m = in.size / k;
    if (!want_acc) return std::make_tuple(defer(in, {m, n}, (int)m, 0, 0, false), py::object(py::none()));
    Tensor<u8_t> out({m, n});
    out.scale = scale_;
    out.zero_point = zero_point_;
    Tensor<int32_t> acc({m, n});
    check(i8ie_layer_forward(q_.get(), in.dptr(), (int)m, 0, 0, in.scale, in.zero_point, out.dptr(), acc.dptr()));
    py::object acc_np = acc.numpy();
    return std::make_tuple(std::move(out), acc_np);
  }

 protected:
  void check_shapes() const override {
    if (wshape_.size() != 2) throw std::runtime_error("i8ie: Linear weight must be [out_features, in_features]");
    if (has_fp32_ && (ssize_t)b_.size() != wshape_[0]) throw std::runtime_error("i8ie: Linear bias size mismatch");
  }
  i8ie_layer* make_handle(ssize_t n) override {
    i8ie_layer* l = nullptr;
    check(i8ie_linear_create(ctx(), reinterpret_cast<const int8_t*>(qw_.data()),
                             reinterpret_cast<const int8_t*>(qb_.data()), (int)n, (int)wshape_[1], w_scale_, &l));
    return l;
  }
};

class Conv2d : public BaseLayer {
 public:
  Conv2d(ssize_t in_channel, ssize_t out_channel, ssize_t kernel_size, ssize_t stride, ssize_t padding)
      : BaseLayer({out_channel, in_channel, kernel_size, kernel_size}, out_channel),
        stride_(stride), padding_(padding) {
    if (stride == 0) throw std::runtime_error("i8ie: Conv2d: stride must not be 0");  // include/conv2d.h:12-14
    if (stride < 0 || padding < 0) throw std::runtime_error("i8ie: Conv2d: negative stride/padding");
  }
  // include/conv2d.h:16-17 leaves stride_/padding_ uninitialised for these two
  // constructors; they are 1 / 0 here.
  Conv2d(py::array_t<float, py::array::c_style | py::array::forcecast> w,
         py::array_t<float, py::array::c_style | py::array::forcecast> b)
      : BaseLayer(w, b) {}

  std::vector<ssize_t> out_shape(const std::vector<ssize_t>& s) const {
    if (s.size() != 4) throw std::runtime_error("i8ie: Conv2d expects an NCHW tensor");
    if (s[1] != wshape_[1]) throw std::runtime_error("i8ie: Conv2d: input channels do not match the weight");
    const ssize_t kh = wshape_[2], kw = wshape_[3];
    if (s[2] - kh + 2 * padding_ < 0 || s[3] - kw + 2 * padding_ < 0)
      throw std::runtime_error("i8ie: Conv2d: kernel larger than the padded input");
    return {s[0], wshape_[0], (s[2] - kh + 2 * padding_) / stride_ + 1, (s[3] - kw + 2 * padding_) / stride_ + 1};
  }
  Tensor<float> forward_f32(Tensor<float>& in) {  // src/conv2d.cc:63-98
    need_fp32();
    check_shapes();
    Tensor<float> out(out_shape(in.shape));
    upload_fp32();
    check(i8ie_conv2d_f32(ctx(), in.dptr(), (int)in.shape[0], (int)in.shape[1], (int)in.shape[2], (int)in.shape[3],
                          w_dev_, b_dev_, (int)wshape_[0], (int)wshape_[2], (int)wshape_[3], (int)stride_,
                          (int)padding_, out.dptr()));
    maybe_sample(out);
    return out;
  }
  std::tuple<Tensor<u8_t>, py::object> forward_u8(Tensor<u8_t>& in, bool want_acc) {  // src/conv2d.cc:100-142
    need_quantized();
    std::vector<ssize_t> oshape = out_shape(in.shape);
    const int n = (int)in.shape[0], h = (int)in.shape[2], w = (int)in.shape[3];
    if (!want_acc) return std::make_tuple(defer(in, oshape, n, h, w, true), py::object(py::none()));
    Tensor<u8_t> out(oshape);
    out.scale = scale_;
    out.zero_point = zero_point_;
    Tensor<int32_t> acc({out.shape[0], out.shape[2] * out.shape[3], out.shape[1]});
    check(i8ie_layer_forward(q_.get(), in.dptr(), n, h, w, in.scale, in.zero_point, out.dptr(), acc.dptr()));
    py::object acc_np = acc.numpy();
    return std::make_tuple(std::move(out), acc_np);
  }

 protected:
  void check_shapes() const override {
    if (wshape_.size() != 4) throw std::runtime_error("i8ie: Conv2d weight must be [out, in, kh, kw]");
    if (has_fp32_ && (ssize_t)b_.size() != wshape_[0]) throw std::runtime_error("i8ie: Conv2d bias size mismatch");
  }
  i8ie_layer* make_handle(ssize_t n) override {
    i8ie_layer* l = nullptr;
    check(i8ie_conv2d_create(ctx(), reinterpret_cast<const int8_t*>(qw_.data()),
                             reinterpret_cast<const int8_t*>(qb_.data()), (int)n, (int)wshape_[1], (int)wshape_[2],
                             (int)wshape_[3], (int)stride_, (int)padding_, w_scale_, &l));
    return l;
  }

 private:
  ssize_t stride_ = 1;
  ssize_t padding_ = 0;
};

template <typename L>
void bind_layer_common(py::class_<L>& c) {
  c.def("load_weight", &L::load_weight)
      .def("load_bias", &L::load_bias)
      .def("prepare", &L::prepare)
      .def("convert", &L::convert)
      .def("__call__", [](L& l, Tensor<float>& x) { return l.forward_f32(x); })
      .def("__call__", [](L& l, Tensor<u8_t>& x) { return std::get<0>(l.forward_u8(x, false)); })
      .def("forward_debug", [](L& l, Tensor<u8_t>& x) { return l.forward_u8(x, true); })
      .def("set_output_qparams", &L::set_output_qparams, py::arg("scale"), py::arg("zero_point"))
      .def("output_qparams", &L::output_qparams)
      .def("load_quantized", &L::load_quantized, py::arg("q_weight"), py::arg("q_bias"), py::arg("weight_scale"),
           py::arg("out_scale"), py::arg("out_zero_point"))
      .def("q_weight", &L::q_weight)
      .def("q_bias", &L::q_bias)
      .def("weight_scale", &L::weight_scale)
      .def("is_quantized", &L::is_quantized);
}

}  // namespace

PYBIND11_MODULE(_CXX_i8ie, m) {
  m.doc() = "i8ie extension module: MI355X (gfx950) HIP backend behind include/i8ie_hip.h";

  // the reference registers Tensor<T> under typeid(...).name() (src/pybind11.cc:12)
  bind_tensor<float>(m, "6TensorIfE");
  bind_tensor<u8_t>(m, "6TensorIhE");
  bind_tensor<s8_t>(m, "6TensorIcE");
  py::class_<Tensor<int32_t>>(m, "6TensorIiE").def("numpy", &Tensor<int32_t>::numpy);

  m.def("tensor", &tensor_from_numpy);  // src/pybind11.cc:38-40
  m.def("tensor_from_device", &tensor_from_device);
  // additive (the reference can only default-construct its s8 tensors from Python): an s8 tensor from an ndarray,
  // so that the s8 overloads of relu / max_pool2d can be exercised
  m.def("tensor_s8", [](py::array_t<s8_t, py::array::c_style | py::array::forcecast> a, float scale, int zp) {
    Tensor<s8_t> t(std::vector<ssize_t>(a.shape(), a.shape() + a.ndim()));
    t.scale = scale;
    t.zero_point = (u8_t)zp;
    if (t.size > 0) {
      py::gil_scoped_release nogil;
      check(i8ie_memcpy_h2d(ctx(), t.dptr(), a.data(), (size_t)t.size));
    }
    return t;
  }, py::arg("array"), py::arg("scale") = 1.0f, py::arg("zero_point") = 0);
  m.def("quantize", &quantize);         // src/pybind11.cc:41-45
  m.def("dequantize", &dequantize);     // src/pybind11.cc:46-48
  m.def("relu", &relu_f32);             // src/functional.cc:73-75
  m.def("relu", &relu_u8);
  m.def("max_pool2d", &max_pool2d_f32);  // src/functional.cc:68-72
  m.def("max_pool2d", &max_pool2d_u8);
  m.def("relu", &relu_s8);               // src/functional.cc:81
  m.def("max_pool2d", &max_pool2d_s8);

  {
    py::class_<Linear> c(m, "Linear");  // src/fully_connected.cc:54-72
    c.def(py::init<py::array_t<float, py::array::c_style | py::array::forcecast>,
                   py::array_t<float, py::array::c_style | py::array::forcecast>>())
        .def(py::init([](Tensor<float>& w, Tensor<float>& b) { return new Linear(w.numpy(), b.numpy()); }))
        .def(py::init<ssize_t, ssize_t>());
    bind_layer_common(c);
  }
  {
    py::class_<Conv2d> c(m, "Conv2d");  // src/conv2d.cc:144-165
    c.def(py::init<py::array_t<float, py::array::c_style | py::array::forcecast>,
                   py::array_t<float, py::array::c_style | py::array::forcecast>>())
        .def(py::init([](Tensor<float>& w, Tensor<float>& b) { return new Conv2d(w.numpy(), b.numpy()); }))
        .def(py::init<ssize_t, ssize_t, ssize_t, ssize_t, ssize_t>(), py::arg("in_channels"),
             py::arg("out_channels"), py::arg("kernel_size"), py::arg("stride") = 1, py::arg("padding") = 0);
    bind_layer_common(c);
  }

  // ---- additive runtime controls -------------------------------------------------
  m.def("abi_version", []() { return i8ie_version(); });
  m.def("device_count", []() {
    int n = 0;
    check(i8ie_device_count(&n));
    return n;
  });
  m.def("set_device", [](int d) {
    if (rt().ctx && rt().device != d) throw std::runtime_error("i8ie: set_device after the context was created");
    rt().device = d;
  });
  m.def("device", []() {
    (void)ctx();
    return rt().device;
  });
  m.def("synchronize", []() {
    py::gil_scoped_release nogil;
    check(i8ie_sync(ctx()));
  });
  m.def("stream", []() { return (uintptr_t)i8ie_ctx_stream(ctx()); });
  m.def("memory_stats", []() {
    size_t live = 0, cached = 0, allocs = 0;
    check(i8ie_memory_stats(ctx(), &live, &cached, &allocs));
    return py::make_tuple(live, cached, allocs);
  });
  m.def("pinned_empty", &pinned_empty);
  m.def("trim", []() {
    for (auto& kv : border_cache())
      for (void* p : kv.second) i8ie_free(ctx(), p);
    border_cache().clear();
    for (auto& kv : upload_cache())
      for (UploadSlot& u : kv.second) {
        check(i8ie_stream_wait_event(ctx(), u.free_ev, 0));  // order the reuse behind the recorded reader
        event_put(u.free_ev);
        i8ie_free(ctx(), u.dev);
      }
    upload_cache().clear();
    for (auto& kv : ppool().free_blocks)
      for (void* p : kv.second) i8ie_host_free(ctx(), p);
    ppool().free_blocks.clear();
    check(i8ie_trim(ctx()));
  });
  m.def("force_fallback", [](bool on) { check(i8ie_ctx_set_option(ctx(), I8IE_OPT_FORCE_FALLBACK, on ? 1 : 0)); });
  m.def("profile_start",
        [](bool mfma_only, int stride) {
          check(i8ie_ctx_set_option(ctx(), I8IE_OPT_PROFILE_STRIDE, stride));
          check(i8ie_profile_start(ctx(), mfma_only ? 1 : 0));
        },
        py::arg("mfma_only") = false, py::arg("stride") = 1);
  m.def("profile_stop", []() {
    std::vector<i8ie_profile_entry> e(64);
    int n = 0;
    {
      py::gil_scoped_release nogil;
      check(i8ie_profile_stop(ctx(), e.data(), (int)e.size(), &n));
    }
    py::dict out;
    for (int i = 0; i < n && i < (int)e.size(); ++i)
      out[py::str(e[i].name)] = py::make_tuple(e[i].launches, e[i].total_ms, e[i].total_ops, e[i].total_bytes);
    return out;
  });
  // run on a borrowed hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); call before the first op
  m.def("use_stream", [](uintptr_t stream, int device) {
    if (rt().ctx) throw std::runtime_error("i8ie: use_stream after the context was created");
    rt().device = device;
    check(i8ie_ctx_create_on_stream(device, (void*)stream, &rt().ctx));
  });
  m.def("set_calibration_seed", [](long seed) { rt().calib_seed = seed; });
  // where Calibrator::sample runs: "auto" (seeded: host replay of the reference's mt19937 stream; unseeded: on the
  // device), "host", "device" (seeded or not: counter-based draws, reproducible for a given seed)
  m.def("set_calibration_mode", [](const std::string& mode) {
    if (mode == "auto") rt().calib_mode = 0;
    else if (mode == "host") rt().calib_mode = 1;
    else if (mode == "device") rt().calib_mode = 2;
    else throw std::invalid_argument("set_calibration_mode: auto | host | device");
  });
  m.def("calibration_mode", []() { return std::string(rt().calib_mode == 0 ? "auto" : rt().calib_mode == 1 ? "host" : "device"); });
  // the device sampler on its own: feed chunks (device tensors), then the 1000 slots and get_range()
  m.def("calibrator_device_samples", [](std::vector<py::array_t<float, py::array::c_style | py::array::forcecast>> chunks, long seed) {
    const long keep_seed = rt().calib_seed;
    rt().calib_seed = seed;
    Calibrator cal;
    try {
      for (auto& c : chunks) {
        float* d = nullptr;
        check(i8ie_malloc(ctx(), (size_t)c.size() * 4 + 4, (void**)&d));
        check(i8ie_memcpy_h2d(ctx(), d, c.data(), (size_t)c.size() * 4));
        cal.sample_device(d, (ssize_t)c.size());
        check(i8ie_sync(ctx()));
        i8ie_free(ctx(), d);
      }
    } catch (...) {
      rt().calib_seed = keep_seed;
      throw;
    }
    rt().calib_seed = keep_seed;
    py::array_t<float> slots(kNumSamples);
    check(i8ie_memcpy_d2h(ctx(), slots.mutable_data(), cal.samples_dev, kNumSamples * sizeof(float)));
    float s;
    u8_t z;
    std::tie(s, z) = cal.get_range(1);
    return py::make_tuple(slots, (long)cal.cnt, s, (int)z);
  });
  // the layers' calibrator on its own (host code, no GPU): feed chunks through sample(), then get_range()
  m.def("calibrator_range",
        [](std::vector<py::array_t<float, py::array::c_style | py::array::forcecast>> chunks, float quantile) {
          Calibrator cal;
          for (auto& c : chunks) cal.sample(c.data(), c.size());
          auto [scale, zp] = cal.get_range(quantile);
          return py::make_tuple(scale, (int)zp);
        });
  // ---- whole-forward replay as one HIP graph (include/i8ie_hip.h, i8ie_graph_*) ----------------------------
  struct Graph {
    i8ie_graph* g = nullptr;
    ~Graph() { if (g) i8ie_graph_destroy(g); }
  };
  py::class_<Graph, std::shared_ptr<Graph>>(m, "Graph")
      .def("launch", [](Graph& gr) { check(i8ie_graph_launch(gr.g)); })
      .def("nodes", [](Graph& gr) {
        int k = 0, n = 0;
        check(i8ie_graph_nodes(gr.g, &k, &n));
        return py::make_tuple(k, n);
      });
  m.def("graph_begin", []() { check(i8ie_graph_begin(ctx())); });
  m.def("graph_end", []() {
    auto gr = std::make_shared<Graph>();
    check(i8ie_graph_end(ctx(), &gr->g));
    return gr;
  });
  // overwrite a resident FP32 tensor's device buffer (the input a captured graph reads) from host memory
  m.def("upload_into", [](Tensor<float>& t, py::array_t<float, py::array::c_style | py::array::forcecast> a) {
    if ((size_t)a.size() != (size_t)t.size) throw std::invalid_argument("upload_into: size mismatch");
    check(i8ie_memcpy_h2d(ctx(), t.dptr(), a.data(), (size_t)t.size * 4));
    if (t.st) {  // a host mirror of the old values (small tensors made from an ndarray) is stale now
      t.st->host_valid = false;
      std::vector<unsigned char>().swap(t.st->host);
    }
  });

  // raw device copy into / out of foreign HIP memory (e.g. a torch tensor's data_ptr) for the
  // multi-GPU logits gather; both sides must be used on this module's stream or synchronised.
  m.def("copy_to_ptr", [](Tensor<float>& t, uintptr_t dst) {
    check(i8ie_memcpy_d2d(ctx(), (void*)dst, t.dptr(), (size_t)t.size * 4));
  });
  m.def("copy_to_ptr", [](Tensor<u8_t>& t, uintptr_t dst) {
    check(i8ie_memcpy_d2d(ctx(), (void*)dst, t.dptr(), (size_t)t.size));
  });
}
