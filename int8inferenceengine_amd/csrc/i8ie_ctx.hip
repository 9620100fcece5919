// i8ie_ctx.hip -- context, errors, device memory of libi8ie_hip.so.
// Replaces the reference's host-side `new T[]` + py::capsule ownership
// (include/tensor.h:26-61) with explicit device allocations on one HIP stream.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <unordered_map>

#include "i8ie_internal.h"

// Stream-ordered caching allocator.  Every consumer of a block (kernels, copies)
// runs on the ctx's one stream, so a block handed back by the host can be given
// to a later op at once: that op is queued behind everything that still reads
// the block.  Avoids a hipStreamSynchronize + hipFree each time the Python side
// drops an intermediate tensor in the middle of a forward pass.
struct I8iePool {
  std::multimap<size_t, void*> free_blocks;      // size -> block
  std::unordered_map<void*, size_t> live;        // block -> size
  size_t bytes_live = 0, bytes_cached = 0, n_hip_malloc = 0;
};
// ---- HIP-event profiler ---------------------------------------------------------
#include <string>
#include <vector>
struct I8ieProfRec {
  std::string name;
  double ops, bytes;
  hipEvent_t e0, e1;
};
struct I8ieProf {
  std::vector<I8ieProfRec> recs;
  std::vector<hipEvent_t> spare;
};
static hipEvent_t prof_event(I8ieProf* p) {
  hipEvent_t e = nullptr;
  if (!p->spare.empty()) {
    e = p->spare.back();
    p->spare.pop_back();
  } else {
    (void)hipEventCreate(&e);
  }
  return e;
}
void i8ie_prof_begin(i8ie_ctx* ctx, const char* name, double ops, double bytes) {
  I8ieProf* p = static_cast<I8ieProf*>(ctx->prof);
  I8ieProfRec r{name, ops, bytes, prof_event(p), prof_event(p)};
  (void)hipEventRecord(r.e0, ctx->stream);
  p->recs.push_back(r);
}
void i8ie_prof_end(i8ie_ctx* ctx) {
  I8ieProf* p = static_cast<I8ieProf*>(ctx->prof);
  (void)hipEventRecord(p->recs.back().e1, ctx->stream);
}

static inline I8iePool* pool_of(i8ie_ctx* c) { return static_cast<I8iePool*>(c->pool); }
static inline size_t pool_round(size_t b) {
  if (b < 256) b = 256;
  if (b <= ((size_t)1 << 20)) return i8ie_align_up(b, 256);
  return i8ie_align_up(b, (size_t)1 << 20);
}
static void pool_trim(i8ie_ctx* c) {
  I8iePool* p = pool_of(c);
  (void)hipStreamSynchronize(c->stream);
  for (auto& kv : p->free_blocks) (void)hipFree(kv.second);
  p->free_blocks.clear();
  p->bytes_cached = 0;
}

static thread_local char g_err[512] = "";

void i8ie_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* i8ie_last_error(void) { return g_err; }
int i8ie_version(void) { return 1; }

int i8ie_device_count(int* n) {
  I8IE_REQUIRE(n != nullptr, "null output");
  I8IE_HIP_TRY(hipGetDeviceCount(n));
  return I8IE_OK;
}

static int ctx_make(int device, hipStream_t borrowed, bool borrow, i8ie_ctx** out) {
  I8IE_REQUIRE(out != nullptr, "null output");
  int ndev = 0;
  I8IE_HIP_TRY(hipGetDeviceCount(&ndev));
  I8IE_REQUIRE(device >= 0 && device < ndev, "no such device");
  I8IE_HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  I8IE_HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    i8ie_set_error("i8ie_ctx_create: device %d is %s; this library is built for gfx950 only", device,
                   prop.gcnArchName);
    return I8IE_ERR_STATE;
  }
  i8ie_ctx* c = new (std::nothrow) i8ie_ctx();
  if (!c) return I8IE_ERR_OOM;
  c->pool = new (std::nothrow) I8iePool();
  if (!c->pool) {
    delete c;
    return I8IE_ERR_OOM;
  }
  c->device = device;
  if (const char* e = std::getenv("I8IE_KERNEL_VARIANT")) c->variant = std::atoi(e);  // A/B aid, see I8IE_OPT_KERNEL_VARIANT
  if (borrow) {
    c->stream = borrowed;
    c->own_stream = false;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      i8ie_set_error("hipStreamCreate: %s", hipGetErrorString(e));
      delete pool_of(c);
      delete c;
      return I8IE_ERR_HIP;
    }
    c->own_stream = true;
  }
  *out = c;
  return I8IE_OK;
}

int i8ie_ctx_create(int device, i8ie_ctx** out) { return ctx_make(device, nullptr, false, out); }
int i8ie_ctx_create_on_stream(int device, void* hip_stream, i8ie_ctx** out) {
  return ctx_make(device, (hipStream_t)hip_stream, true, out);
}

int i8ie_ctx_destroy(i8ie_ctx* ctx) {
  if (!ctx) return I8IE_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  pool_trim(ctx);
  for (auto& kv : pool_of(ctx)->live) (void)hipFree(kv.first);  // leaked by the caller
  delete pool_of(ctx);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->retired_ws) {
    auto* r = static_cast<std::vector<void*>*>(ctx->retired_ws);
    for (void* w : *r) (void)hipFree(w);
    delete r;
  }
  if (ctx->copy_stream) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
  }
  if (ctx->pinned) {
    auto* m = static_cast<std::unordered_map<void*, size_t>*>(ctx->pinned);
    for (auto& kv : *m) (void)hipHostFree(kv.first);  // leaked by the caller
    delete m;
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return I8IE_OK;
}

void* i8ie_ctx_stream(i8ie_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int i8ie_sync(i8ie_ctx* ctx) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8IE_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return I8IE_OK;
}

int i8ie_malloc(i8ie_ctx* ctx, size_t bytes, void** dev) {
  I8IE_REQUIRE(ctx != nullptr && dev != nullptr, "null argument");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  *dev = nullptr;
  I8iePool* p = pool_of(ctx);
  const size_t want = pool_round(bytes);
  auto it = p->free_blocks.lower_bound(want);
  if (it != p->free_blocks.end() && it->first <= want + want / 4) {
    *dev = it->second;
    p->bytes_cached -= it->first;
    p->live[*dev] = it->first;
    p->bytes_live += it->first;
    p->free_blocks.erase(it);
    return I8IE_OK;
  }
  hipError_t e = hipMalloc(dev, want);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    pool_trim(ctx);  // give cached blocks back and retry once
    e = hipMalloc(dev, want);
  }
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    *dev = nullptr;
    i8ie_set_error("i8ie_malloc: out of device memory (%zu bytes)", bytes);
    return I8IE_ERR_OOM;
  }
  I8IE_HIP_TRY(e);
  p->n_hip_malloc++;
  p->live[*dev] = want;
  p->bytes_live += want;
  return I8IE_OK;
}

int i8ie_free(i8ie_ctx* ctx, void* dev) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  if (!dev) return I8IE_OK;
  I8iePool* p = pool_of(ctx);
  auto it = p->live.find(dev);
  I8IE_REQUIRE(it != p->live.end(), "pointer was not allocated by i8ie_malloc on this ctx");
  if (ctx->capture) {  // the graph being captured has this address baked in: it owns the block until it is destroyed
    static_cast<std::vector<void*>*>(ctx->capture)->push_back(dev);
    return I8IE_OK;
  }
  const size_t sz = it->second;
  p->live.erase(it);
  p->bytes_live -= sz;
  p->free_blocks.emplace(sz, dev);  // no sync: reuse is ordered by the ctx's stream
  p->bytes_cached += sz;
  return I8IE_OK;
}

int i8ie_trim(i8ie_ctx* ctx) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  pool_trim(ctx);
  return I8IE_OK;
}

int i8ie_memory_stats(i8ie_ctx* ctx, size_t* bytes_live, size_t* bytes_cached, size_t* n_device_allocs) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  I8iePool* p = pool_of(ctx);
  if (bytes_live) *bytes_live = p->bytes_live;
  if (bytes_cached) *bytes_cached = p->bytes_cached + ctx->ws_bytes;
  if (n_device_allocs) *n_device_allocs = p->n_hip_malloc;
  return I8IE_OK;
}

int i8ie_profile_start(i8ie_ctx* ctx, int mfma_kernels_only) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  if (!ctx->prof) ctx->prof = new (std::nothrow) I8ieProf();
  I8IE_REQUIRE(ctx->prof != nullptr, "out of host memory");
  ctx->prof_mfma_only = mfma_kernels_only ? 1 : 0;
  ctx->prof_seen = 0;
  return I8IE_OK;
}

int i8ie_profile_stop(i8ie_ctx* ctx, i8ie_profile_entry* entries, int max_entries, int* n_entries) {
  I8IE_REQUIRE(ctx != nullptr && n_entries != nullptr, "null argument");
  *n_entries = 0;
  if (!ctx->prof) return I8IE_OK;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8IE_HIP_TRY(hipStreamSynchronize(ctx->stream));
  I8ieProf* p = static_cast<I8ieProf*>(ctx->prof);
  std::map<std::string, i8ie_profile_entry> agg;
  for (auto& r : p->recs) {
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    i8ie_profile_entry& e = agg[r.name];
    if (e.launches == 0) {
      memset(&e, 0, sizeof(e));
      strncpy(e.name, r.name.c_str(), sizeof(e.name) - 1);
    }
    e.launches += 1;
    e.total_ms += ms;
    e.total_ops += r.ops;
    e.total_bytes += r.bytes;
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  for (hipEvent_t e : p->spare) (void)hipEventDestroy(e);
  delete p;
  ctx->prof = nullptr;
  for (auto& kv : agg) {
    if (entries && *n_entries < max_entries) entries[*n_entries] = kv.second;
    *n_entries += 1;
  }
  return I8IE_OK;
}

int i8ie_memcpy_h2d(i8ie_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  if (bytes == 0) return I8IE_OK;
  I8IE_REQUIRE(dst_dev != nullptr && src_host != nullptr, "null pointer");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8IE_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  // the host buffer may be pageable and freed by the caller right after return
  I8IE_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return I8IE_OK;
}

int i8ie_memcpy_d2h(i8ie_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  if (bytes == 0) return I8IE_OK;
  I8IE_REQUIRE(dst_host != nullptr && src_dev != nullptr, "null pointer");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8IE_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  I8IE_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return I8IE_OK;
}

int i8ie_memcpy_d2d(i8ie_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  if (bytes == 0) return I8IE_OK;
  I8IE_REQUIRE(dst_dev != nullptr && src_dev != nullptr, "null pointer");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8IE_HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return I8IE_OK;
}

// ---- asynchronous transfers: pinned host blocks, a transfer stream, events -------------------
struct i8ie_event {
  hipEvent_t ev;
  int device;
};

static std::unordered_map<void*, size_t>* pinned_of(i8ie_ctx* ctx) {
  if (!ctx->pinned) ctx->pinned = new std::unordered_map<void*, size_t>();
  return static_cast<std::unordered_map<void*, size_t>*>(ctx->pinned);
}
// the block [p, p+bytes) lies inside one i8ie_host_malloc allocation of this ctx
static bool is_pinned(i8ie_ctx* ctx, const void* p, size_t bytes) {
  if (!ctx->pinned) return false;
  for (auto& kv : *pinned_of(ctx)) {
    const char* base = static_cast<const char*>(kv.first);
    const char* q = static_cast<const char*>(p);
    if (q >= base && q + bytes <= base + kv.second) return true;
  }
  return false;
}
static int stream_for(i8ie_ctx* ctx, int on_copy_stream, hipStream_t* out) {
  if (!on_copy_stream) {
    *out = ctx->stream;
    return I8IE_OK;
  }
  if (!ctx->copy_stream) I8IE_HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  *out = ctx->copy_stream;
  return I8IE_OK;
}

int i8ie_host_malloc(i8ie_ctx* ctx, size_t bytes, void** host) {
  I8IE_REQUIRE(ctx != nullptr && host != nullptr, "null argument");
  *host = nullptr;
  if (bytes == 0) return I8IE_OK;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    i8ie_set_error("hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? I8IE_ERR_OOM : I8IE_ERR_HIP;
  }
  (*pinned_of(ctx))[p] = bytes;
  *host = p;
  return I8IE_OK;
}

int i8ie_host_free(i8ie_ctx* ctx, void* host) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  if (!host) return I8IE_OK;
  auto* m = pinned_of(ctx);
  auto it = m->find(host);
  I8IE_REQUIRE(it != m->end(), "i8ie_host_free: not a block of i8ie_host_malloc");
  m->erase(it);
  I8IE_HIP_TRY(hipHostFree(host));  // waits for transfers that still use the block
  return I8IE_OK;
}

int i8ie_host_is_pinned(i8ie_ctx* ctx, const void* p, size_t bytes, int* yes) {
  I8IE_REQUIRE(ctx != nullptr && yes != nullptr, "null argument");
  *yes = (p != nullptr && is_pinned(ctx, p, bytes)) ? 1 : 0;
  return I8IE_OK;
}

int i8ie_memcpy_h2d_async(i8ie_ctx* ctx, void* dst_dev, const void* src_pinned, size_t bytes, int on_copy_stream) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  if (bytes == 0) return I8IE_OK;
  I8IE_REQUIRE(dst_dev != nullptr && src_pinned != nullptr, "null pointer");
  I8IE_REQUIRE(is_pinned(ctx, src_pinned, bytes), "i8ie_memcpy_h2d_async: source is not i8ie_host_malloc memory");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s;
  if (int rc = stream_for(ctx, on_copy_stream, &s)) return rc;
  I8IE_HIP_TRY(hipMemcpyAsync(dst_dev, src_pinned, bytes, hipMemcpyHostToDevice, s));
  return I8IE_OK;
}

int i8ie_memcpy_d2h_async(i8ie_ctx* ctx, void* dst_pinned, const void* src_dev, size_t bytes, int on_copy_stream) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  if (bytes == 0) return I8IE_OK;
  I8IE_REQUIRE(dst_pinned != nullptr && src_dev != nullptr, "null pointer");
  I8IE_REQUIRE(is_pinned(ctx, dst_pinned, bytes), "i8ie_memcpy_d2h_async: destination is not i8ie_host_malloc memory");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s;
  if (int rc = stream_for(ctx, on_copy_stream, &s)) return rc;
  I8IE_HIP_TRY(hipMemcpyAsync(dst_pinned, src_dev, bytes, hipMemcpyDeviceToHost, s));
  return I8IE_OK;
}

int i8ie_event_create(i8ie_ctx* ctx, i8ie_event** out) {
  I8IE_REQUIRE(ctx != nullptr && out != nullptr, "null argument");
  *out = nullptr;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  i8ie_event* e = new (std::nothrow) i8ie_event();
  if (!e) return I8IE_ERR_OOM;
  e->device = ctx->device;
  hipError_t rc = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);
  if (rc != hipSuccess) {
    delete e;
    i8ie_set_error("hipEventCreate: %s", hipGetErrorString(rc));
    return I8IE_ERR_HIP;
  }
  *out = e;
  return I8IE_OK;
}

int i8ie_event_record(i8ie_ctx* ctx, i8ie_event* ev, int on_copy_stream) {
  I8IE_REQUIRE(ctx != nullptr && ev != nullptr, "null argument");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s;
  if (int rc = stream_for(ctx, on_copy_stream, &s)) return rc;
  I8IE_HIP_TRY(hipEventRecord(ev->ev, s));
  return I8IE_OK;
}

int i8ie_stream_wait_event(i8ie_ctx* ctx, i8ie_event* ev, int copy_stream_waits) {
  I8IE_REQUIRE(ctx != nullptr && ev != nullptr, "null argument");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s;
  if (int rc = stream_for(ctx, copy_stream_waits, &s)) return rc;
  I8IE_HIP_TRY(hipStreamWaitEvent(s, ev->ev, 0));
  return I8IE_OK;
}

int i8ie_event_synchronize(i8ie_event* ev) {
  I8IE_REQUIRE(ev != nullptr, "null event");
  I8IE_HIP_TRY(hipSetDevice(ev->device));
  I8IE_HIP_TRY(hipEventSynchronize(ev->ev));
  return I8IE_OK;
}

int i8ie_event_query(i8ie_event* ev, int* done) {
  I8IE_REQUIRE(ev != nullptr && done != nullptr, "null argument");
  I8IE_HIP_TRY(hipSetDevice(ev->device));
  hipError_t rc = hipEventQuery(ev->ev);
  if (rc != hipSuccess && rc != hipErrorNotReady) {
    i8ie_set_error("hipEventQuery: %s", hipGetErrorString(rc));
    return I8IE_ERR_HIP;
  }
  *done = rc == hipSuccess;
  return I8IE_OK;
}

int i8ie_event_destroy(i8ie_event* ev) {
  if (!ev) return I8IE_OK;
  (void)hipSetDevice(ev->device);
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return I8IE_OK;
}

int i8ie_memset(i8ie_ctx* ctx, void* dst_dev, int byte, size_t bytes) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  if (bytes == 0) return I8IE_OK;
  I8IE_REQUIRE(dst_dev != nullptr, "null pointer");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8IE_HIP_TRY(hipMemsetAsync(dst_dev, byte, bytes, ctx->stream));
  return I8IE_OK;
}

struct i8ie_graph {
  i8ie_ctx* ctx;
  hipGraph_t graph;
  hipGraphExec_t exec;
  std::vector<void*> held;  // blocks freed during the capture
};

int i8ie_ctx_is_capturing(i8ie_ctx* ctx, int* yes) {
  I8IE_REQUIRE(ctx != nullptr && yes != nullptr, "null argument");
  *yes = ctx->capture != nullptr ? 1 : 0;
  return I8IE_OK;
}

int i8ie_graph_begin(i8ie_ctx* ctx) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  I8IE_REQUIRE(ctx->capture == nullptr, "a capture is already open on this ctx");
  I8IE_REQUIRE(ctx->prof == nullptr, "per-kernel timing is on: stop it before capturing a graph");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8IE_HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
  ctx->capture = new std::vector<void*>();
  return I8IE_OK;
}

int i8ie_graph_end(i8ie_ctx* ctx, i8ie_graph** out) {
  I8IE_REQUIRE(ctx != nullptr && out != nullptr, "null argument");
  I8IE_REQUIRE(ctx->capture != nullptr, "no capture is open on this ctx");
  std::vector<void*>* held = static_cast<std::vector<void*>*>(ctx->capture);
  ctx->capture = nullptr;
  auto give_back = [&]() {
    for (void* b : *held) (void)i8ie_free(ctx, b);
    delete held;
  };
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
  if (e != hipSuccess || graph == nullptr) {
    (void)hipGetLastError();
    give_back();
    i8ie_set_error("hipStreamEndCapture: %s (a call inside the capture synchronised or allocated?)", hipGetErrorString(e));
    return I8IE_ERR_HIP;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    (void)hipGraphDestroy(graph);
    give_back();
    i8ie_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
    return I8IE_ERR_HIP;
  }
  i8ie_graph* g = new i8ie_graph{ctx, graph, exec, std::move(*held)};
  delete held;
  ctx->live_graphs += 1;  // (pins the workspace: i8ie_ws_reserve)
  *out = g;
  return I8IE_OK;
}

int i8ie_graph_launch(i8ie_graph* g) {
  I8IE_REQUIRE(g != nullptr, "null graph");
  I8IE_REQUIRE(g->ctx->capture == nullptr, "a graph cannot be replayed inside a capture");
  I8IE_HIP_TRY(hipGraphLaunch(g->exec, g->ctx->stream));
  return I8IE_OK;
}

int i8ie_graph_nodes(i8ie_graph* g, int* kernel_nodes, int* all_nodes) {
  I8IE_REQUIRE(g != nullptr, "null graph");
  size_t n = 0;
  I8IE_HIP_TRY(hipGraphGetNodes(g->graph, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  if (n) I8IE_HIP_TRY(hipGraphGetNodes(g->graph, nodes.data(), &n));
  int k = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    I8IE_HIP_TRY(hipGraphNodeGetType(nodes[i], &t));
    if (t == hipGraphNodeTypeKernel) ++k;
  }
  if (kernel_nodes) *kernel_nodes = k;
  if (all_nodes) *all_nodes = (int)n;
  return I8IE_OK;
}

int i8ie_graph_destroy(i8ie_graph* g) {
  if (!g) return I8IE_OK;
  (void)hipStreamSynchronize(g->ctx->stream);  // no replay may still be reading the held blocks
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  for (void* b : g->held) (void)i8ie_free(g->ctx, b);
  i8ie_ctx* ctx = g->ctx;
  if (--ctx->live_graphs == 0 && ctx->retired_ws) {  // no replay can touch an outgrown workspace any more
    auto* r = static_cast<std::vector<void*>*>(ctx->retired_ws);
    for (void* w : *r) (void)hipFree(w);
    delete r;
    ctx->retired_ws = nullptr;
  }
  delete g;
  return I8IE_OK;
}

}  // extern "C"

int i8ie_ws_reserve(i8ie_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return I8IE_OK;
  I8IE_REQUIRE(ctx->capture == nullptr, "workspace growth inside a graph capture: run the same calls once eagerly first");
  I8IE_HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->ws) {
    if (ctx->live_graphs > 0) {
      // captured graphs replay kernels that read and write this workspace: it stays allocated until the last of
      // them is destroyed; eager calls move on to the larger one
      if (!ctx->retired_ws) ctx->retired_ws = new std::vector<void*>();
      static_cast<std::vector<void*>*>(ctx->retired_ws)->push_back(ctx->ws);
    } else {
      I8IE_HIP_TRY(hipFree(ctx->ws));
    }
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  size_t want = i8ie_align_up(bytes + bytes / 8, (size_t)1 << 20);
  hipError_t e = hipMalloc(&ctx->ws, want);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    i8ie_set_error("workspace: out of device memory (%zu bytes)", want);
    return I8IE_ERR_OOM;
  }
  I8IE_HIP_TRY(e);
  ctx->ws_bytes = want;
  return I8IE_OK;
}
