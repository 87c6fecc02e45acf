// engine_capi.hip — C-ABI implementation (include/kangaroo_amd.h) over the HIP walk kernel.
//
// Host half of the drop-in for the reference's GPUEngine (math/gpu/engine.cu).  Differences that
// are deliberate and MI355X-first:
//   * two private streams per engine and two sets of launch buffers (DP ring, jump-index history, DP chains):
//     kg_launch queues launch i+1 BEFORE it waits for launch i, so the GPU never idles between launches, and the
//     history kernel of launch i (distances, DP distances, the PCIe writes of the items) runs on the second stream
//     underneath the walk kernel of launch i+1 (the reference re-launches only after unpacking, engine.cu:515-538);
//   * DP items land directly in host-visible pinned memory; only the 4-byte counter is copied;
//   * the herd is stored as 12 lane-contiguous uint64 planes indexed by host kangaroo index
//     (no per-block pinned repacking loop as in engine.cu:289-320);
//   * SetKangaroo is a 1-thread kernel on the same stream instead of twelve blocking 8-byte
//     copies (engine.cu:385-425), so it is ordered between launches without stalling the host.
// There is NO CPU fallback: every entry point that computes needs a gfx950 device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kangaroo_amd.h"
#include "walk_common.h"
#include "walk_kernel_rot.h"
#include "walk_kernel_stream.h"
#include "herd_kernel.h"
#include "kat_kernel.h"
#include "../../include/kangaroo_amd_host.h"

using namespace kg;

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return fail(KG_ERR_HIP, std::string("GPUEngine: ") + #expr + ": " + hipGetErrorString(_e));     \
  } while (0)

// group order n (math/secp256k1.cpp:17), little-endian limbs
static const u64 ORDER_N[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL,
                               0xFFFFFFFFFFFFFFFFULL};

// r = a + b mod n for a, b < 2^256 (Int::ModAddK1order, math/cpu/intmod.cpp:837-842)
static void order_add(u64 r[4], const u64 a[4], const u64 b[4]) {
  unsigned __int128 c = 0;
  u64 t[4], s[4];
  for (int i = 0; i < 4; ++i) {
    c += (unsigned __int128)a[i] + b[i];
    t[i] = (u64)c;
    c >>= 64;
  }
  u64 carry = (u64)c, br = 0;
  for (int i = 0; i < 4; ++i) {
    unsigned __int128 d = (unsigned __int128)t[i] - ORDER_N[i] - br;
    s[i] = (u64)d;
    br = (u64)(d >> 64) & 1;
  }
  const bool negative = (carry == 0) && br;
  memcpy(r, negative ? t : s, 32);
}
// r = a - b mod n (Int::ModSubK1order, math/cpu/intmod.cpp:844-848)
static void order_sub(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 t[4], br = 0;
  for (int i = 0; i < 4; ++i) {
    unsigned __int128 d = (unsigned __int128)a[i] - b[i] - br;
    t[i] = (u64)d;
    br = (u64)(d >> 64) & 1;
  }
  if (br) {
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) {
      c += (unsigned __int128)t[i] + ORDER_N[i];
      t[i] = (u64)c;
      c >>= 64;
    }
  }
  memcpy(r, t, 32);
}

struct kg_engine {
  int gpu = 0, groups = 0, per_group = 0;
  u64 n = 0, cap = 0;
  u32 threads = 0, grid = 0;
  int K = 0, DW = 0;          // kangaroos per lane (all launch segments together), distance-sum words
  int nseg = 1, K_seg = 0;    // streaming kernel: the herd is walked in nseg launch segments of K_seg kangaroos per lane
  bool streaming = false;     // streaming kernel (walk_kernel_stream.h) instead of the register-resident one
  u32 *ws = nullptr;          // its workspace
  size_t ws_bytes = 0;
  size_t ws_laid_out = 0;     // stream_ws_need the current layout of ws was computed for
  u32 nb_run = KG_NB_RUN;
  u32 max_found = 0;          // the caller's hint (core/kangaroo.cpp:408), floor of the ring capacity
  u32 ring_cap[2] = {0, 0};   // items each DP buffer holds
  u32 ring_max = 0;           // upper bound of the ring (kg_tuning.ring_max_items or KG_RING_MAX)
  hipStream_t stream = nullptr;   // walk kernels, herd upload / download
  hipStream_t stream2 = nullptr;  // streaming kernel only: kg_hist_sum_kernel, SetKangaroo pokes, the launch's copies
  hipEvent_t ev_start[2] = {nullptr, nullptr}, ev_stop[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
  hipEvent_t ev_sum0[2] = {nullptr, nullptr}, ev_sum1[2] = {nullptr, nullptr};  // around kg_hist_sum_kernel
  hipEvent_t ev_walk[2] = {nullptr, nullptr};  // walk kernel finished, clock stamps copied (stream2 waits for it)
  std::vector<hipEvent_t> ev_seg;  // [2][nseg - 1]: walk kernel of a launch segment but the last finished
  hipEvent_t ev_poke = nullptr;   // after the last SetKangaroo poke queued on stream2
  bool pokes_queued = false;      // pokes on stream2 that the next walk kernel must wait for
  u64 *X = nullptr, *Y = nullptr, *D = nullptr;
  JumpTableDev *jt = nullptr;
  DpItemDev *out_host[2] = {nullptr, nullptr};
  DpItemDev *out_dev[2] = {nullptr, nullptr};
  u32 *count_dev = nullptr;   // [2]
  u32 *count_host = nullptr;  // pinned [2] DP counters, then [2][4] clock stamps of the streaming kernel
  u64 *stage = nullptr;       // pinned staging for herd upload/download
  size_t stage_elems = 0;
  u64 wild[4] = {0, 0, 0, 0};
  u64 dp_mask = 0;
  int max_jd_bits = 0;
  bool params_set = false, herd_set = false;
  int q[2] = {-1, -1}, qn = 0;  // launch buffers in flight, oldest first (kg_launch keeps two queued)
  int next_buf = 0;
  int pending = -1;  // completed by kick_and_wait, not yet collected
  float last_ms = 0.f, last_sum_ms = 0.f;
  u64 last_clk = 0, last_ref = 0;  // shader cycles and 100 MHz reference ticks of the last streaming launch (workgroup 0)
  int cus = 0;
  char name[256] = {0};
};

// ---- kernel dispatch ------------------------------------------------------------------------
// Shipped instantiations only: the streaming kernel (default) for 3- and 8-word distance sums, and the
// register-resident kernel for herds beyond the streaming workspace limit (K = 16, one wave per SIMD).
// Measurement variants live in tools/, not in the library.
typedef void (*walk_fn)(WalkArgs);
struct Variant {
  int K, DW, W, U;  // kangaroos per lane, distance words, waves per SIMD, window of the rotating kernel
  walk_fn fn;
};
#define KG_VARIANT_ROT(K, U, DW, W) {K, DW, W, U, kg_walk_kernel_rot<K, U, DW, W>}
static const Variant g_variants[] = {
    KG_VARIANT_ROT(16, 4, 3, 1), KG_VARIANT_ROT(16, 4, 8, 1),
#ifdef KG_EXTRA_VARIANTS
    KG_EXTRA_VARIANTS
#endif
};

// The streaming kernel does not depend on the width of the distance sums: distances are summed from the jump-index
// history by kg_hist_sum_kernel<DW> afterwards.
typedef void (*stream_fn)(StreamArgs);
#ifndef KG_STREAM_W
#define KG_STREAM_W 2  // waves per SIMD the streaming kernel is compiled for
#endif
static const stream_fn g_stream_fn = kg_walk_kernel_stream<KG_STREAM_W>;

static const Variant *find_variant(int K, int DW) {
  for (const Variant &v : g_variants)
    if (v.K == K && v.DW == DW) return &v;
  return nullptr;
}

static int bit_length256(const u64 v[4]) {
  for (int i = 3; i >= 0; --i)
    if (v[i]) return 64 * i + 64 - __builtin_clzll(v[i]);
  return 0;
}

__global__ void kg_poke_kernel(u64 *X, u64 *Y, u64 *D, u64 cap, u64 p, u64 x0, u64 x1, u64 x2, u64 x3, u64 y0,
                               u64 y1, u64 y2, u64 y3, u64 d0, u64 d1, u64 d2, u64 d3) {
  X[0 * cap + p] = x0; X[1 * cap + p] = x1; X[2 * cap + p] = x2; X[3 * cap + p] = x3;
  Y[0 * cap + p] = y0; Y[1 * cap + p] = y1; Y[2 * cap + p] = y2; Y[3 * cap + p] = y3;
  D[0 * cap + p] = d0; D[1 * cap + p] = d1; D[2 * cap + p] = d2; D[3 * cap + p] = d3;
}

// ---- device queries ------------------------------------------------------------------------------
extern "C" int kg_device_count(int *count) {
  if (!count) return fail(KG_ERR_ARG, "kg_device_count: null argument");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    return fail(KG_ERR_HIP, std::string("GPUEngine: hipGetDeviceCount ") + hipGetErrorString(e));
  }
  *count = c;
  return KG_OK;
}

extern "C" int kg_default_grid(int gpu_id, int *x, int *y) {
  if (!x || !y) return fail(KG_ERR_ARG, "kg_default_grid: null argument");
  if (*x > 0 && *y > 0) return KG_OK;
  int c = 0;
  if (kg_device_count(&c) != KG_OK) return KG_ERR_HIP;
  if (c == 0) return fail(KG_ERR_NODEVICE, "GPUEngine: There are no available device(s) that support HIP");
  if (gpu_id < 0 || gpu_id >= c) return fail(KG_ERR_ARG, "GPUEngine::GetGridSize() Invalid gpuId");
  int cus = 0;
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gpu_id));
  // 2 groups per CU x 128 threads x 128 kangaroos: 8.4 M kangaroos on 256 CUs (0.8 GB of state)
  if (*x <= 0) *x = 2 * cus;
  if (*y <= 0) *y = 128;
  return KG_OK;
}

extern "C" int kg_device_pci_bus_id(int gpu_id, char *buf, size_t buflen) {
  if (!buf || buflen < 13) return fail(KG_ERR_ARG, "kg_device_pci_bus_id: buffer of at least 13 bytes needed");
  int c = 0;
  if (kg_device_count(&c) != KG_OK) return KG_ERR_HIP;
  if (gpu_id < 0 || gpu_id >= c) return fail(KG_ERR_ARG, "kg_device_pci_bus_id: invalid gpuId");
  HIPCHK(hipDeviceGetPCIBusId(buf, (int)buflen, gpu_id));
  return KG_OK;
}

extern "C" int kg_print_device_info(void) {
  int c = 0;
  if (kg_device_count(&c) != KG_OK) return KG_ERR_HIP;
  for (int i = 0; i < c; ++i) {
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, i));
    printf("GPU #%d %s (%d CUs) (%s) (%.1f MB)\n", i, p.name, p.multiProcessorCount, p.gcnArchName,
           (double)p.totalGlobalMem / 1048576.0);
  }
  return KG_OK;
}

extern "C" void *kg_alloc_pinned(size_t size) {
  void *p = nullptr;
  if (hipHostMalloc(&p, size, hipHostMallocPortable) != hipSuccess) {
    g_err = "GPUEngine: AllocatePinnedMemory failed";
    return nullptr;
  }
  return p;
}
extern "C" void kg_free_pinned(void *p) {
  if (p) (void)hipHostFree(p);
}

extern "C" const char *kg_last_error(void) { return g_err.c_str(); }
extern "C" const char *kg_version(void) { return "kangaroo_amd 0.1 (gfx950)"; }

// ---- static workspace for kg_kick_static (the entry point that allocates nothing) -----------------------------
// File-scope device array: every GPU the library runs on holds its own copy until the process ends.  In host code
// the NAME of the array is only a stand-in (passing it to a kernel compiles and then faults on the GPU): the
// device address is looked up once per device with hipGetSymbolAddress while that device is current (kg_create)
// and only that pointer is used afterwards.
static __device__ __align__(256) unsigned char scratch[KG_STATIC_WORKSPACE_BYTES];
constexpr int KG_MAX_DEVICES = 64;
static void *g_scratch_ptr[KG_MAX_DEVICES];
static std::mutex g_scratch_mu;

// device must be current
static int scratch_lookup(int gpu) {
  if (gpu < 0 || gpu >= KG_MAX_DEVICES) return fail(KG_ERR_ARG, "GPUEngine: gpuId beyond the static workspace table");
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[gpu]) return KG_OK;
  void *ptr = nullptr;
  hipError_t he = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
  if (he != hipSuccess || !ptr)
    return fail(KG_ERR_HIP, std::string("GPUEngine: hipGetSymbolAddress(scratch): ") + hipGetErrorString(he));
  g_scratch_ptr[gpu] = ptr;
  return KG_OK;
}

// both of the engine's streams idle
static int sync_streams(kg_engine *e) {
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->stream2) HIPCHK(hipStreamSynchronize(e->stream2));
  return KG_OK;
}

// ---- DP ring ------------------------------------------------------------------------------------------
// Two host-pinned buffers the kernels append distinguished points to.  The reference sizes its buffer from the
// caller's maxFound alone and exits when a launch produces more (math/gpu/engine.cu:99,506-510; core/kangaroo.cpp:408
// passes 131072, which a default grid overflows at once at DP = 8).  Here maxFound is a floor: the ring
// holds twice what a launch is expected to report, n * nb_run / 2^dpBits, re-derived whenever the DP mask or the step
// count changes, up to KG_RING_MAX items per buffer; only beyond that are items dropped and reported (KG_ERR_LOST).
constexpr u32 KG_RING_MAX = 1u << 25;  // 33.5 M items = 2.7 GB of pinned memory per buffer

static u32 ring_want(const kg_engine *e) {
  const u64 floor_items = std::max<u64>(e->max_found, std::min<u64>(std::min<u64>(2 * e->n, 1u << 22), e->ring_max));
  if (!e->params_set) return (u32)floor_items;
  const int bits = __builtin_popcountll(e->dp_mask);
  const long double expect = (long double)e->n * (long double)e->nb_run / (long double)(bits >= 64 ? 1.8446744073709552e19L : (long double)(1ull << bits));
  const long double want = 2.0L * expect + 65536.0L;
  const u64 ceil_items = std::max<u64>(e->ring_max, e->max_found);
  const u64 w = want >= (long double)ceil_items ? ceil_items : (u64)want;
  return (u32)std::min<u64>(std::max<u64>(floor_items, w), ceil_items);
}

// (re)allocate buffer b for `items` items; nothing may be using it
static int ring_alloc(kg_engine *e, int b, u32 items) {
  if (e->out_host[b]) {
    HIPCHK(hipHostFree(e->out_host[b]));
    e->out_host[b] = nullptr;
    e->out_dev[b] = nullptr;
    e->ring_cap[b] = 0;
  }
  HIPCHK(hipHostMalloc((void **)&e->out_host[b], (size_t)items * sizeof(DpItemDev), hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void **)&e->out_dev[b], e->out_host[b], 0));
  e->ring_cap[b] = items;
  return KG_OK;
}

// grow buffer b to what the current configuration needs (never shrinks).  The stream must not hold work on b.
static int ring_ensure(kg_engine *e, int b) {
  const u32 want = ring_want(e);
  if (e->ring_cap[b] >= want) return KG_OK;
  if (sync_streams(e) != KG_OK) return KG_ERR_HIP;
  return ring_alloc(e, b, want);
}

// after kg_set_params / kg_set_nb_run: grow both buffers now if no launch is in flight or waiting to be collected
// (otherwise kg_kick grows the buffer it is about to use)
static int ring_follow_config(kg_engine *e) {
  if (e->qn > 0 || e->pending >= 0) return KG_OK;
  for (int b = 0; b < 2; ++b) {
    const int rc = ring_ensure(e, b);
    if (rc != KG_OK) return rc;
  }
  return KG_OK;
}

// The streaming kernel's launch-time workspace: the kangaroo slots and the pacing table (+ clock stamps) once, and
// per launch buffer (`nbuf` of them) the jump-index history and the DP chains (one head per kangaroo slot, one link
// per ring item; walk_kernel_stream.h).  kg_kick / kg_launch use two launch buffers, so that the history kernel of one
// launch can run underneath the walk kernel of the next; kg_kick_static uses one.
static size_t stream_ws_part(const kg_engine *e) {  // the slots of ONE launch segment (all segments use the same ones)
  return (size_t)WS_SLOT_WORDS * (size_t)e->K_seg * (size_t)e->threads * sizeof(u32);
}
static size_t stream_hist_part(const kg_engine *e) {  // 16-byte aligned
  return ((size_t)e->nb_run * (size_t)e->K * (size_t)e->threads + 15) & ~(size_t)15;
}
static size_t stream_pace_part() { return (size_t)(PACE_KEYS * 4 + PACE_TAIL) * sizeof(u32); }  // a multiple of 16
static size_t stream_head_part(const kg_engine *e) { return (size_t)e->cap * sizeof(u32); }     // cap is a multiple of 256
static size_t stream_link_part(const kg_engine *e) {                                             // 16-byte aligned
  return ((size_t)std::max(e->ring_cap[0], e->ring_cap[1]) * sizeof(DpLink) + 15) & ~(size_t)15;
}
static size_t stream_ws_need(const kg_engine *e, int nbuf) {
  return stream_ws_part(e) + stream_pace_part() +
         (size_t)nbuf * (stream_hist_part(e) + stream_head_part(e) + stream_link_part(e));
}
struct StreamWs {
  u32 *slots, *pace, *dp_head;
  unsigned char *hist;
  DpLink *dp_link;
};
// pointers of launch buffer hb (< nbuf) inside a workspace laid out for nbuf buffers
static StreamWs stream_ws_at(const kg_engine *e, void *ws, int nbuf, int hb) {
  unsigned char *p = (unsigned char *)ws;
  StreamWs w;
  w.slots = (u32 *)p;
  p += stream_ws_part(e);
  w.pace = (u32 *)p;
  p += stream_pace_part();
  w.hist = p + (size_t)hb * stream_hist_part(e);
  p += (size_t)nbuf * stream_hist_part(e);
  w.dp_head = (u32 *)(p + (size_t)hb * stream_head_part(e));
  p += (size_t)nbuf * stream_head_part(e);
  w.dp_link = (DpLink *)(p + (size_t)hb * stream_link_part(e));
  return w;
}

// ---- lifetime ---------------------------------------------------------------------------------------
static int bind(const kg_engine *e) {
  HIPCHK(hipSetDevice(e->gpu));
  return KG_OK;
}

extern "C" int kg_destroy(kg_engine *e) {
  if (!e) return KG_OK;
  (void)hipSetDevice(e->gpu);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->stream2) (void)hipStreamSynchronize(e->stream2);
  if (e->ev_poke) (void)hipEventDestroy(e->ev_poke);
  for (hipEvent_t ev : e->ev_seg)
    if (ev) (void)hipEventDestroy(ev);
  for (int b = 0; b < 2; ++b) {
    if (e->ev_walk[b]) (void)hipEventDestroy(e->ev_walk[b]);
    if (e->ev_sum0[b]) (void)hipEventDestroy(e->ev_sum0[b]);
    if (e->ev_sum1[b]) (void)hipEventDestroy(e->ev_sum1[b]);
    if (e->ev_start[b]) (void)hipEventDestroy(e->ev_start[b]);
    if (e->ev_stop[b]) (void)hipEventDestroy(e->ev_stop[b]);
    if (e->ev_done[b]) (void)hipEventDestroy(e->ev_done[b]);
    if (e->out_host[b]) (void)hipHostFree(e->out_host[b]);
  }
  if (e->X) (void)hipFree(e->X);
  if (e->Y) (void)hipFree(e->Y);
  if (e->D) (void)hipFree(e->D);
  if (e->jt) (void)hipFree(e->jt);
  if (e->count_dev) (void)hipFree(e->count_dev);
  if (e->count_host) (void)hipHostFree(e->count_host);
  if (e->stage) (void)hipHostFree(e->stage);
  if (e->ws) (void)hipFree(e->ws);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  if (e->stream2) (void)hipStreamDestroy(e->stream2);
  delete e;
  return KG_OK;
}

extern "C" int kg_create_tuned(int gpu_id, int groups, int per_group, uint32_t max_found, const kg_tuning *tune,
                               kg_engine **out) {
  if (!out) return fail(KG_ERR_ARG, "kg_create: null out pointer");
  *out = nullptr;
  if (groups <= 0 || per_group <= 0) return fail(KG_ERR_ARG, "kg_create: grid must be positive");
  const int want_mode = tune ? tune->mode : KG_MODE_AUTO;
  const int want_k = tune ? tune->k_per_lane : 0;
  if (want_mode != KG_MODE_AUTO && want_mode != KG_MODE_STREAM && want_mode != KG_MODE_REG)
    return fail(KG_ERR_ARG, "kg_create: unknown kg_tuning.mode");
  if (want_k < 0) return fail(KG_ERR_ARG, "kg_create: kg_tuning.k_per_lane must not be negative");
  int c = 0;
  if (kg_device_count(&c) != KG_OK) return KG_ERR_HIP;
  if (c == 0) return fail(KG_ERR_NODEVICE, "GPUEngine: There are no available device(s) that support HIP");
  if (gpu_id < 0 || gpu_id >= c) return fail(KG_ERR_ARG, "GPUEngine: invalid gpuId");
  hipError_t he = hipSetDevice(gpu_id);
  if (he != hipSuccess) return fail(KG_ERR_HIP, std::string("GPUEngine: ") + hipGetErrorString(he));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, gpu_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(KG_ERR_NODEVICE, std::string("GPUEngine: device is ") + prop.gcnArchName +
                                     ", this engine is built for gfx950 only");
  if (scratch_lookup(gpu_id) != KG_OK) return KG_ERR_HIP;

  kg_engine *e = new kg_engine();
  e->gpu = gpu_id;
  e->groups = groups;
  e->per_group = per_group;
  e->cus = prop.multiProcessorCount;
  e->n = (u64)KG_GROUP_SIZE * (u64)groups * (u64)per_group;
  e->streaming = want_mode != KG_MODE_REG;
  if (e->streaming) {
    // K kangaroos per lane such that the herd fills about two waves per SIMD (always even: the streaming kernel
    // handles kangaroos in pairs)
    const u64 lanes = (u64)e->cus * 4 * 64 * (u64)KG_STREAM_W;
    u64 k = want_k > 0 ? (u64)want_k : std::max<u64>((e->n + lanes - 1) / lanes, 4);
    k = std::min<u64>(std::max<u64>(k, WS_GROUP), 1024);
    k = (k + WS_GROUP - 1) / WS_GROUP * WS_GROUP;
    // The streaming kernel addresses its workspace through ONE 32-bit buffer descriptor: 96 bytes per kangaroo slot
    // must stay below 4 GiB, i.e. at most 44.7 M kangaroos per kernel launch.  A larger herd is walked in several LAUNCH
    // SEGMENTS per Launch: segment s takes kangaroos s * K_seg .. (s + 1) * K_seg - 1 of every lane, the kernels run one
    // after the other on the same workspace (the slots are scratch of one kernel launch), each with its own part of the
    // jump-index history and of the DP chains.  Round 2 fell back to the slower, spilling register-resident kernel
    // there.  The reference's own limit is its int offset, ~178 M kangaroos (engine.cu:17).
    const u64 lanes_k = ((e->n + k - 1) / k + WALK_BLOCK - 1) / WALK_BLOCK * WALK_BLOCK;
    const u64 limit = (tune && tune->ws_limit_bytes) ? tune->ws_limit_bytes : (1ull << 32);
    const u64 slot_bytes = lanes_k * (u64)(WS_SLOT_WORDS * 4);  // workspace bytes of ONE kangaroo per lane
    if (limit > (1ull << 32) || WS_GROUP * slot_bytes >= limit) {
      delete e;
      return fail(KG_ERR_ARG, "kg_create: kg_tuning.ws_limit_bytes must hold two kangaroos per lane and not exceed 4 GiB");
    }
    // launch segments: as many kangaroos per lane as fit the descriptor, an even number
    u64 kseg_max = (limit - 1) / slot_bytes;
    kseg_max -= kseg_max % WS_GROUP;
    const u64 nseg = (k + kseg_max - 1) / kseg_max;
    u64 kseg = (k + nseg - 1) / nseg;
    kseg = (kseg + WS_GROUP - 1) / WS_GROUP * WS_GROUP;
    e->nseg = (int)nseg;
    e->K_seg = (int)kseg;
    e->K = (int)(nseg * kseg);
  }
  if (!e->streaming) {
    e->K = want_k > 0 ? want_k : 16;
    if (!find_variant(e->K, 3)) {
      delete e;
      return fail(KG_ERR_ARG, "kg_create: the register-resident kernel is built for 16 kangaroos per lane");
    }
  }
  e->DW = 3;
  if (!e->streaming) e->K_seg = e->K;
  const u64 per_lane = (e->n + e->K - 1) / e->K;
  const u64 blocks = (per_lane + WALK_BLOCK - 1) / WALK_BLOCK;
  if (blocks > 0x7FFFFFFFull) {
    delete e;
    return fail(KG_ERR_ARG, "kg_create: herd too large");
  }
  e->grid = (u32)blocks;
  e->threads = e->grid * WALK_BLOCK;
  e->cap = (u64)e->threads * e->K;
  // the caller's maxFound is a hint (core/kangaroo.cpp:408) and the floor of the DP ring, see ring_want()
  e->max_found = max_found;
  e->ring_max = (tune && tune->ring_max_items) ? tune->ring_max_items : KG_RING_MAX;
  snprintf(e->name, sizeof e->name, "GPU #%d %s (%dx%d cores) Grid(%dx%d)", gpu_id,
           prop.name[0] ? prop.name : prop.gcnArchName,
           prop.multiProcessorCount, 64 * 4, groups, per_group);

#define CRT(expr)                                                                       \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      std::string m = std::string("GPUEngine: ") + #expr + ": " + hipGetErrorString(_e); \
      kg_destroy(e);                                                                    \
      return fail(KG_ERR_HIP, m);                                                       \
    }                                                                                   \
  } while (0)
  CRT(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  if (e->streaming) {
    CRT(hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking));
    CRT(hipEventCreateWithFlags(&e->ev_poke, hipEventDisableTiming));
  }
  for (int b = 0; b < 2; ++b) {
    CRT(hipEventCreateWithFlags(&e->ev_walk[b], hipEventDisableTiming));
    CRT(hipEventCreate(&e->ev_sum0[b]));
    CRT(hipEventCreate(&e->ev_sum1[b]));
    CRT(hipEventCreate(&e->ev_start[b]));
    CRT(hipEventCreate(&e->ev_stop[b]));
    CRT(hipEventCreateWithFlags(&e->ev_done[b], hipEventDisableTiming));
  }
  e->ev_seg.assign((size_t)2 * (e->nseg - 1), nullptr);
  for (hipEvent_t &ev : e->ev_seg) CRT(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const size_t plane_bytes = (size_t)e->cap * 4 * sizeof(u64);
  CRT(hipMalloc((void **)&e->X, plane_bytes));
  CRT(hipMalloc((void **)&e->Y, plane_bytes));
  CRT(hipMalloc((void **)&e->D, plane_bytes));
  CRT(hipMalloc((void **)&e->jt, sizeof(JumpTableDev)));
  CRT(hipMalloc((void **)&e->count_dev, 2 * sizeof(u32)));
  CRT(hipHostMalloc((void **)&e->count_host, (2 + 2 * PACE_TAIL) * sizeof(u32), hipHostMallocDefault));
  memset(e->count_host, 0, (2 + 2 * PACE_TAIL) * sizeof(u32));
  for (int b = 0; b < 2; ++b) {
    const u32 items = ring_want(e);
    CRT(hipHostMalloc((void **)&e->out_host[b], (size_t)items * sizeof(DpItemDev), hipHostMallocMapped));
    CRT(hipHostGetDevicePointer((void **)&e->out_dev[b], e->out_host[b], 0));
    e->ring_cap[b] = items;
  }
  e->stage_elems = (size_t)std::min<u64>(e->cap, 1u << 20);
  CRT(hipHostMalloc((void **)&e->stage, e->stage_elems * 12 * sizeof(u64), hipHostMallocDefault));
  CRT(hipMemsetAsync(e->count_dev, 0, 2 * sizeof(u32), e->stream));
  CRT(hipStreamSynchronize(e->stream));
#undef CRT
  *out = e;
  return KG_OK;
}

extern "C" int kg_create(int gpu_id, int groups, int per_group, uint32_t max_found, kg_engine **out) {
  return kg_create_tuned(gpu_id, groups, per_group, max_found, nullptr, out);
}

// ---- accessors ----------------------------------------------------------------------------------------
extern "C" int kg_device_name(const kg_engine *e, char *buf, size_t buflen) {
  if (!e || !buf || !buflen) return fail(KG_ERR_ARG, "kg_device_name: bad argument");
  snprintf(buf, buflen, "%s", e->name);
  return KG_OK;
}
extern "C" int kg_memory_bytes(const kg_engine *e, uint64_t *bytes) {
  if (!e || !bytes) return fail(KG_ERR_ARG, "kg_memory_bytes: bad argument");
  u64 b = (u64)e->cap * 12 * sizeof(u64) + ((u64)e->ring_cap[0] + e->ring_cap[1]) * sizeof(DpItemDev) + sizeof(JumpTableDev);
  if (e->streaming) b += stream_ws_need(e, 2);  // slots, pacing table, two sets of jump-index history and DP chains
  *bytes = b;
  return KG_OK;
}
extern "C" int kg_nb_thread(const kg_engine *e) { return e ? e->groups * e->per_group : 0; }
extern "C" int kg_group_size(const kg_engine *) { return KG_GROUP_SIZE; }
extern "C" uint64_t kg_nb_kangaroo(const kg_engine *e) { return e ? e->n : 0; }

extern "C" int kg_set_wild_offset(kg_engine *e, const uint64_t offset[4]) {
  if (!e || !offset) return fail(KG_ERR_ARG, "kg_set_wild_offset: bad argument");
  memcpy(e->wild, offset, 32);
  return KG_OK;
}

extern "C" int kg_set_nb_run(kg_engine *e, uint32_t nb_run) {
  if (!e || nb_run == 0) return fail(KG_ERR_ARG, "kg_set_nb_run: bad argument");
  e->nb_run = nb_run;
  if (bind(e) != KG_OK) return KG_ERR_HIP;
  return ring_follow_config(e);
}

static int pick_delta_words(const kg_engine *e) {
  // delta accumulates nb_run jump distances: needs max_jd_bits + ceil(log2(nb_run)) bits
  int lg = 0;
  while ((1ull << lg) < e->nb_run) ++lg;
  const int need = e->max_jd_bits + lg;
  return need <= 96 ? 3 : 8;
}

extern "C" int kg_set_params(kg_engine *e, uint64_t dp_mask, const uint64_t *jd, const uint64_t *jpx,
                             const uint64_t *jpy) {
  if (!e || !jd || !jpx || !jpy) return fail(KG_ERR_ARG, "kg_set_params: bad argument");
  if (bind(e) != KG_OK) return KG_ERR_HIP;
  JumpTableDev h;
  e->max_jd_bits = 0;
  for (int i = 0; i < KG_NB_JUMP; ++i) {
    u32 w[8];
    fe f;
    for (int k = 0; k < 4; ++k) {
      w[2 * k] = (u32)jpx[4 * i + k];
      w[2 * k + 1] = (u32)(jpx[4 * i + k] >> 32);
    }
    fe_from_words(f, w);
    fe_canon(f);
    for (int k = 0; k < 9; ++k) h.pxy[i][k] = f.l[k];
    for (int k = 0; k < 4; ++k) {
      w[2 * k] = (u32)jpy[4 * i + k];
      w[2 * k + 1] = (u32)(jpy[4 * i + k] >> 32);
    }
    fe_from_words(f, w);
    fe_canon(f);
    for (int k = 0; k < 9; ++k) h.pxy[i][9 + k] = f.l[k];
    for (int k = 0; k < 4; ++k) {
      h.dist[i][2 * k] = (u32)jd[4 * i + k];
      h.dist[i][2 * k + 1] = (u32)(jd[4 * i + k] >> 32);
    }
    e->max_jd_bits = std::max(e->max_jd_bits, bit_length256(jd + 4 * i));
  }
  e->dp_mask = dp_mask;
  // behind any launch in flight: its walk kernel (stream) and its history kernel (stream2) both read the table
  if (sync_streams(e) != KG_OK) return KG_ERR_HIP;
  HIPCHK(hipMemcpyAsync(e->jt, &h, sizeof h, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->params_set = true;
  return ring_follow_config(e);
}

// ---- herd upload / download --------------------------------------------------------------------------
static int wait_idle(kg_engine *e) { return sync_streams(e); }

extern "C" int kg_set_kangaroos(kg_engine *e, const uint64_t *px, const uint64_t *py, const uint64_t *d, uint64_t n) {
  if (!e || !px || !py || !d) return fail(KG_ERR_ARG, "kg_set_kangaroos: bad argument");
  if (n != e->n) return fail(KG_ERR_ARG, "kg_set_kangaroos: n must equal kg_nb_kangaroo()");
  if (bind(e) != KG_OK || wait_idle(e) != KG_OK) return KG_ERR_HIP;
  const size_t chunk = e->stage_elems;
  u64 *sx = e->stage, *sy = e->stage + 4 * chunk, *sd = e->stage + 8 * chunk;
  for (u64 p0 = 0; p0 < e->cap; p0 += chunk) {
    const u64 cnt = std::min<u64>(chunk, e->cap - p0);
    for (u64 i = 0; i < cnt; ++i) {
      const u64 p = p0 + i;
      const u64 src = p < n ? p : 0;  // padding lanes replay kangaroo 0 (never reported: p >= n)
      u64 dd[4];
      if (src & 1) order_add(dd, d + 4 * src, e->wild);  // engine.cu:305-308
      else memcpy(dd, d + 4 * src, 32);
      for (int w = 0; w < 4; ++w) {
        sx[(size_t)w * chunk + i] = px[4 * src + w];
        sy[(size_t)w * chunk + i] = py[4 * src + w];
        sd[(size_t)w * chunk + i] = dd[w];
      }
    }
    for (int w = 0; w < 4; ++w) {
      HIPCHK(hipMemcpyAsync(e->X + (u64)w * e->cap + p0, sx + (size_t)w * chunk, cnt * 8, hipMemcpyHostToDevice, e->stream));
      HIPCHK(hipMemcpyAsync(e->Y + (u64)w * e->cap + p0, sy + (size_t)w * chunk, cnt * 8, hipMemcpyHostToDevice, e->stream));
      HIPCHK(hipMemcpyAsync(e->D + (u64)w * e->cap + p0, sd + (size_t)w * chunk, cnt * 8, hipMemcpyHostToDevice, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  e->herd_set = true;
  return KG_OK;
}

// ---- herd creation on the GPU -------------------------------------------------------------
namespace {
std::once_flag g_comb_once;
std::vector<u32> g_comb_tab;  // [32][256][18]
u64 g_comb_sx[4], g_comb_sy[4];
void to_limbs18(u32 *dst, const u64 x[4], const u64 y[4]) {
  for (int c = 0; c < 2; ++c) {
    const u64 *src = c ? y : x;
    u32 w[8];
    for (int i = 0; i < 4; ++i) {
      w[2 * i] = (u32)src[i];
      w[2 * i + 1] = (u32)(src[i] >> 32);
    }
    fe f;
    fe_from_words(f, w);
    for (int i = 0; i < 9; ++i) dst[9 * c + i] = f.l[i];
  }
}
}  // namespace

extern "C" int kg_herd_from_distances(kg_engine *e, const uint64_t *d, uint64_t n, const uint64_t key_x[4],
                                      const uint64_t key_y[4]) {
  if (!e || !d || !key_x || !key_y) return fail(KG_ERR_ARG, "kg_herd_from_distances: bad argument");
  if (n != e->n) return fail(KG_ERR_ARG, "kg_herd_from_distances: n must equal kg_nb_kangaroo()");
  if (bind(e) != KG_OK || wait_idle(e) != KG_OK) return KG_ERR_HIP;
  std::call_once(g_comb_once, [] {
    g_comb_tab.resize((size_t)HERD_WINDOWS * 256 * 18);
    kgh_comb_table(g_comb_tab.data(), g_comb_sx, g_comb_sy);
  });
  HerdArgs a;
  to_limbs18(a.start[0], g_comb_sx, g_comb_sy);
  const bool have_key = (key_x[0] | key_x[1] | key_x[2] | key_x[3] | key_y[0] | key_y[1] | key_y[2] | key_y[3]) != 0;
  if (have_key) {
    u64 wx[4], wy[4];
    if (kgh_ec_add(wx, wy, g_comb_sx, g_comb_sy, key_x, key_y) != 0)
      return fail(KG_ERR_ARG, "kg_herd_from_distances: key collides with the comb start point");
    to_limbs18(a.start[1], wx, wy);
  } else {
    memcpy(a.start[1], a.start[0], sizeof a.start[0]);
  }
  // geometry of its own: about two waves per SIMD, at most 64 points per lane
  const u64 lanes_target = (u64)e->cus * 4 * 64 * 2;
  u64 K = std::min<u64>(std::max<u64>((n + lanes_target - 1) / lanes_target, 1), 64);
  u64 lanes = ((n + K - 1) / K + WALK_BLOCK - 1) / WALK_BLOCK * WALK_BLOCK;
  u32 *tab_dev = nullptr, *ws = nullptr;
  u64 *k_dev = nullptr;
  int rc = KG_OK;
  auto cleanup = [&]() {
    if (tab_dev) (void)hipFree(tab_dev);
    if (ws) (void)hipFree(ws);
    if (k_dev) (void)hipFree(k_dev);
  };
#define HT(expr)                                                                                         \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) {                                                                              \
      cleanup();                                                                                         \
      return fail(KG_ERR_HIP, std::string("GPUEngine: ") + #expr + ": " + hipGetErrorString(_e));        \
    }                                                                                                    \
  } while (0)
  HT(hipMalloc((void **)&tab_dev, g_comb_tab.size() * sizeof(u32)));
  HT(hipMalloc((void **)&k_dev, (size_t)n * 32));
  HT(hipMalloc((void **)&ws, (size_t)K * 27 * lanes * sizeof(u32)));
  HT(hipMemcpyAsync(tab_dev, g_comb_tab.data(), g_comb_tab.size() * sizeof(u32), hipMemcpyHostToDevice, e->stream));
  HT(hipMemcpyAsync(k_dev, d, (size_t)n * 32, hipMemcpyHostToDevice, e->stream));
  a.tab = tab_dev;
  a.k = k_dev;
  a.n = n;
  a.lanes = (u32)lanes;
  a.K = (u32)K;
  a.ws = ws;
  a.X = e->X;
  a.Y = e->Y;
  a.cap = e->cap;
  hipLaunchKernelGGL(kg_herd_kernel, dim3((u32)(lanes / WALK_BLOCK)), dim3(WALK_BLOCK), 0, e->stream, a);
  HT(hipGetLastError());
  // distances: wild offset applied on the host (engine.cu:305-308), planes uploaded in chunks
  {
    const size_t chunk = e->stage_elems;
    u64 *sd = e->stage;
    for (u64 p0 = 0; p0 < n; p0 += chunk) {
      const u64 cnt = std::min<u64>(chunk, n - p0);
      for (u64 i = 0; i < cnt; ++i) {
        const u64 p = p0 + i;
        u64 dd[4];
        if (p & 1) order_add(dd, d + 4 * p, e->wild);
        else memcpy(dd, d + 4 * p, 32);
        for (int w = 0; w < 4; ++w) sd[(size_t)w * chunk + i] = dd[w];
      }
      for (int w = 0; w < 4; ++w)
        HT(hipMemcpyAsync(e->D + (u64)w * e->cap + p0, sd + (size_t)w * chunk, cnt * 8, hipMemcpyHostToDevice, e->stream));
      HT(hipStreamSynchronize(e->stream));
    }
  }
  if (e->cap > n) {
    const u64 pad = e->cap - n;
    hipLaunchKernelGGL(kg_pad_kernel, dim3((u32)((pad + 255) / 256)), dim3(256), 0, e->stream, e->X, e->Y, e->D, e->cap, n);
    HT(hipGetLastError());
  }
  HT(hipStreamSynchronize(e->stream));
#undef HT
  cleanup();
  e->herd_set = true;
  return rc;
}

extern "C" int kg_get_kangaroos(kg_engine *e, uint64_t *px, uint64_t *py, uint64_t *d, uint64_t n) {
  if (!e || !px || !py || !d) return fail(KG_ERR_ARG, "kg_get_kangaroos: bad argument");
  if (n != e->n) return fail(KG_ERR_ARG, "kg_get_kangaroos: n must equal kg_nb_kangaroo()");
  if (bind(e) != KG_OK || wait_idle(e) != KG_OK) return KG_ERR_HIP;
  const size_t chunk = e->stage_elems;
  u64 *sx = e->stage, *sy = e->stage + 4 * chunk, *sd = e->stage + 8 * chunk;
  for (u64 p0 = 0; p0 < n; p0 += chunk) {
    const u64 cnt = std::min<u64>(chunk, n - p0);
    for (int w = 0; w < 4; ++w) {
      HIPCHK(hipMemcpyAsync(sx + (size_t)w * chunk, e->X + (u64)w * e->cap + p0, cnt * 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(sy + (size_t)w * chunk, e->Y + (u64)w * e->cap + p0, cnt * 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(sd + (size_t)w * chunk, e->D + (u64)w * e->cap + p0, cnt * 8, hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    for (u64 i = 0; i < cnt; ++i) {
      const u64 p = p0 + i;
      u64 dd[4];
      for (int w = 0; w < 4; ++w) {
        px[4 * p + w] = sx[(size_t)w * chunk + i];
        py[4 * p + w] = sy[(size_t)w * chunk + i];
        dd[w] = sd[(size_t)w * chunk + i];
      }
      if (p & 1) order_sub(dd, dd, e->wild);  // engine.cu:371
      memcpy(d + 4 * p, dd, 32);
    }
  }
  return KG_OK;
}

extern "C" int kg_set_kangaroo(kg_engine *e, uint64_t k_idx, const uint64_t px[4], const uint64_t py[4],
                               const uint64_t d[4]) {
  if (!e || !px || !py || !d) return fail(KG_ERR_ARG, "kg_set_kangaroo: bad argument");
  if (k_idx >= e->n) return fail(KG_ERR_ARG, "kg_set_kangaroo: kIdx out of range");
  if (bind(e) != KG_OK) return KG_ERR_HIP;
  u64 dd[4];
  if (k_idx & 1) order_add(dd, d, e->wild);  // engine.cu:414-416
  else memcpy(dd, d, 32);
  // Streaming kernel: the poke goes behind the history kernel of the newest launch queued (stream2 is in order, and
  // that kernel starts only after its walk kernel): the launch's epilogue has written X / Y and its history sum has
  // been added to D by then.  The next walk kernel waits for the poke (launch_walk).
  hipStream_t st = e->stream2 ? e->stream2 : e->stream;
  hipLaunchKernelGGL(kg_poke_kernel, dim3(1), dim3(1), 0, st, e->X, e->Y, e->D, e->cap, (u64)k_idx, px[0],
                     px[1], px[2], px[3], py[0], py[1], py[2], py[3], dd[0], dd[1], dd[2], dd[3]);
  HIPCHK(hipGetLastError());
  if (e->stream2) e->pokes_queued = true;
  return KG_OK;
}

// ---- launch / collect ------------------------------------------------------------------------------------
// depth: launches that may already be in flight (0 for the public kg_kick, 1 for the kick inside kg_launch)
static int kick_checks(kg_engine *e, int depth) {
  if (!e) return fail(KG_ERR_ARG, "kg_kick: null engine");
  if (!e->params_set || !e->herd_set) return fail(KG_ERR_STATE, "GPUEngine: Kernel: SetParams/SetKangaroos not called");
  if (e->qn > depth) return fail(KG_ERR_STATE, "GPUEngine: Kernel: a launch is already in flight");
  if (e->pending >= 0) return fail(KG_ERR_STATE, "GPUEngine: Kernel: previous launch not collected");
  if (bind(e) != KG_OK) return KG_ERR_HIP;
  e->DW = pick_delta_words(e);
  return KG_OK;
}

// Queue one launch into launch buffer next_buf.  ws: workspace of the streaming kernel laid out for nbuf launch
// buffers (stream_ws_need bytes), unused by the register-resident kernels.  Streaming kernel: the walk kernel goes to
// `stream`, the history kernel and the launch's last copies to `stream2` behind an event, so that with two launch
// buffers they run underneath the NEXT walk kernel.  Allocates nothing.
static int launch_walk(kg_engine *e, void *ws, int nbuf) {
  const Variant *v = e->streaming ? nullptr : find_variant(e->K, e->DW);
  if (!v && !e->streaming) return fail(KG_ERR_STATE, "GPUEngine: Kernel: no kernel variant for this configuration");
  const int b = e->next_buf;
  WalkArgs a;
  a.X = e->X;
  a.Y = e->Y;
  a.D = e->D;
  a.cap = e->cap;
  a.n = e->n;
  a.threads = e->threads;
  a.nb_run = e->nb_run;
  a.dp_mask_hi = (u32)(e->dp_mask >> 32);
  a.dp_mask_lo = (u32)e->dp_mask;
  a.jt = e->jt;
  a.out = e->out_dev[b];
  a.out_count = e->count_dev + b;
  a.max_found = e->ring_cap[b];
  const size_t lds = e->streaming ? walk_stream_lds_bytes() : walk_rot_lds_bytes(e->K, e->DW);
  if (lds > 64 * 1024) {  // above the default dynamic-LDS limit (gfx950 has 160 KiB per CU)
    const void *fn = e->streaming ? (const void *)g_stream_fn : (const void *)v->fn;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipStream_t sA = e->stream;
  hipStream_t sB = e->streaming ? e->stream2 : e->stream;  // where the launch is finished
  if (e->pokes_queued) {
    // SetKangaroo pokes sit on stream2 behind the previous launch's history kernel: this walk must see them
    HIPCHK(hipEventRecord(e->ev_poke, e->stream2));
    HIPCHK(hipStreamWaitEvent(sA, e->ev_poke, 0));
    e->pokes_queued = false;
  }
  HIPCHK(hipMemsetAsync(e->count_dev + b, 0, sizeof(u32), sA));
  a.p_base = 0;
  if (e->streaming) {
    const StreamWs w = stream_ws_at(e, ws, nbuf, nbuf > 1 ? b : 0);
    HIPCHK(hipMemsetAsync(w.dp_head, 0xFF, stream_head_part(e), sA));  // DP_NONE
    const u32 sum_grid = (u32)(((u64)(e->K_seg / 2) * e->threads + WALK_BLOCK - 1) / WALK_BLOCK);  // a thread per pair of slots and lane
    const u64 seg_slots = (u64)e->K_seg * e->threads;  // kangaroo positions of one launch segment
    for (int sg = 0; sg < e->nseg; ++sg) {
      // launch segment sg: kangaroos sg * K_seg .. of every lane, i.e. positions [off, off + seg_slots)
      const u64 off = (u64)sg * seg_slots;
      StreamArgs sa;
      sa.a = a;
      sa.a.X = e->X + off;
      sa.a.Y = e->Y + off;
      sa.a.D = e->D + off;
      sa.a.n = e->n > off ? e->n - off : 0;
      sa.a.p_base = off;
      sa.ws = w.slots;
      sa.hist = w.hist + (size_t)sg * e->nb_run * seg_slots;
      sa.K = (u32)e->K_seg;
      sa.pace = w.pace;
      sa.dp_head = w.dp_head + off;
      sa.dp_link = w.dp_link;
      HIPCHK(hipMemsetAsync(sa.pace, 0, stream_pace_part(), sA));
      if (sg == 0) HIPCHK(hipEventRecord(e->ev_start[b], sA));
      hipLaunchKernelGGL(g_stream_fn, dim3(e->grid), dim3(WALK_BLOCK), lds, sA, sa);
      HIPCHK(hipGetLastError());
      const bool last = sg == e->nseg - 1;
      hipEvent_t walked = last ? e->ev_walk[b] : e->ev_seg[(size_t)b * (e->nseg - 1) + sg];
      if (last) {
        HIPCHK(hipEventRecord(e->ev_stop[b], sA));
        // clock stamps written by workgroup 0 behind the pacing table (the next launch clears the table: copy on sA)
        HIPCHK(hipMemcpyAsync(e->count_host + 2 + b * PACE_TAIL, w.pace + PACE_KEYS * 4, PACE_TAIL * sizeof(u32),
                              hipMemcpyDeviceToHost, sA));
      }
      HIPCHK(hipEventRecord(walked, sA));
      HIPCHK(hipStreamWaitEvent(sB, walked, 0));
      // distances: D += sum over the segment's jump-index history, and the distances of its DP items
      if (sg == 0) HIPCHK(hipEventRecord(e->ev_sum0[b], sB));
      if (e->DW == 3)
        hipLaunchKernelGGL(kg_hist_sum_kernel<3>, dim3(sum_grid), dim3(WALK_BLOCK), 0, sB, sa.hist, e->jt, sa.a.D,
                           e->cap, sa.a.n, e->threads, (u32)e->K_seg, e->nb_run, sa.dp_head, sa.dp_link, e->out_dev[b]);
      else
        hipLaunchKernelGGL(kg_hist_sum_kernel<8>, dim3(sum_grid), dim3(WALK_BLOCK), 0, sB, sa.hist, e->jt, sa.a.D,
                           e->cap, sa.a.n, e->threads, (u32)e->K_seg, e->nb_run, sa.dp_head, sa.dp_link, e->out_dev[b]);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(e->ev_sum1[b], sB));
  } else {
    HIPCHK(hipEventRecord(e->ev_start[b], sA));
    hipLaunchKernelGGL(v->fn, dim3(e->grid), dim3(WALK_BLOCK), lds, sA, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev_stop[b], sA));
  }
  HIPCHK(hipMemcpyAsync(e->count_host + b, e->count_dev + b, sizeof(u32), hipMemcpyDeviceToHost, sB));
  HIPCHK(hipEventRecord(e->ev_done[b], sB));
  e->q[e->qn++] = b;
  e->next_buf = b ^ 1;
  return KG_OK;
}

static int kick_depth(kg_engine *e, int depth) {
  const int rc = kick_checks(e, depth);
  if (rc != KG_OK) return rc;
  // the DP buffer this launch appends to follows the configuration (the other one may still hold a launch in flight
  // or the previous launch's items, which kg_launch unpacks after this call)
  const int rr = ring_ensure(e, e->next_buf);
  if (rr != KG_OK) return rr;
  if (e->streaming) {
    const size_t need = stream_ws_need(e, 2);  // grows with nb_run (kg_set_nb_run) and the DP ring; kept between launches
    if (need > e->ws_bytes) {
      if (sync_streams(e) != KG_OK) return KG_ERR_HIP;
      if (e->ws) (void)hipFree(e->ws);
      e->ws = nullptr;
      e->ws_bytes = 0;
      HIPCHK(hipMalloc((void **)&e->ws, need));
      e->ws_bytes = need;
    } else if (need != e->ws_laid_out && e->qn > 0) {
      // same allocation, other layout (nb_run or the ring changed while a launch is in flight): let it finish first
      if (sync_streams(e) != KG_OK) return KG_ERR_HIP;
    }
    e->ws_laid_out = need;
  }
  return launch_walk(e, e->ws, 2);
}

extern "C" int kg_kick(kg_engine *e) { return kick_depth(e, 0); }

// kg_kick for callers that cannot allocate: the streaming kernel's workspace is the library's static device
// array instead of a hipMalloc'd buffer.  No hipMalloc / hipFree / allocating routine on this path.
extern "C" int kg_kick_static(kg_engine *e) {
  const int rc = kick_checks(e, 0);
  if (rc != KG_OK) return rc;
  if (!e->streaming) return fail(KG_ERR_ARG, "kg_kick_static: the engine does not run the streaming kernel");
  if (e->ring_cap[e->next_buf] < ring_want(e))  // growing it would allocate: kg_set_params / kg_set_nb_run do that while idle
    return fail(KG_ERR_STATE, "kg_kick_static: the DP ring is smaller than this configuration needs; call kg_set_params / "
                              "kg_set_nb_run while no launch is in flight (they size it), or kg_kick once");
  const size_t need = stream_ws_need(e, 1);
  if (need > (size_t)KG_STATIC_WORKSPACE_BYTES) {
    char msg[200];
    snprintf(msg, sizeof msg, "kg_kick_static: this herd and step count need %zu bytes of workspace, the static workspace holds %zu",
             need, (size_t)KG_STATIC_WORKSPACE_BYTES);
    return fail(KG_ERR_ARG, msg);
  }
  void *ws = (e->gpu >= 0 && e->gpu < KG_MAX_DEVICES) ? g_scratch_ptr[e->gpu] : nullptr;  // looked up in kg_create
  if (!ws) return fail(KG_ERR_STATE, "kg_kick_static: static workspace address unknown for this device");
  return launch_walk(e, ws, 1);
}

// wait for the oldest launch in flight and take it off the queue; *buf = its launch buffer
static int wait_oldest(kg_engine *e, int *buf) {
  const int b = e->q[0];
  HIPCHK(hipEventSynchronize(e->ev_done[b]));
  e->q[0] = e->q[1];
  e->q[1] = -1;
  --e->qn;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, e->ev_start[b], e->ev_stop[b]) == hipSuccess) e->last_ms = ms;
  e->last_sum_ms = 0.f;
  if (e->streaming && hipEventElapsedTime(&ms, e->ev_sum0[b], e->ev_sum1[b]) == hipSuccess) e->last_sum_ms = ms;
  const u32 *st = e->count_host + 2 + b * PACE_TAIL;
  e->last_clk = (u64)st[0] | ((u64)st[1] << 32);
  e->last_ref = (u64)st[2] | ((u64)st[3] << 32);
  *buf = b;
  return KG_OK;
}

static int unpack(kg_engine *e, int b, kg_item *items, uint32_t cap, uint32_t *n, uint32_t *lost) {
  const u32 found = e->count_host[b];
  const u32 stored = std::min(found, e->ring_cap[b]);
  const u32 take = std::min(stored, cap);
  const DpItemDev *src = e->out_host[b];
  auto convert = [&](u32 lo, u32 hi) {
    for (u32 i = lo; i < hi; ++i) {
      kg_item &it = items[i];
      for (int k = 0; k < 4; ++k) {
        it.x[k] = (u64)src[i].x[2 * k] | ((u64)src[i].x[2 * k + 1] << 32);
        it.d[k] = (u64)src[i].d[2 * k] | ((u64)src[i].d[2 * k + 1] << 32);
      }
      it.kIdx = src[i].kidx;
      if (it.kIdx & 1) order_sub(it.d, it.d, e->wild);  // engine.cu:534
    }
  };
  if (take < (1u << 18)) {
    convert(0, take);
  } else {
    // low DP sizes: millions of items per launch (8.4 M at DP = 8 on the default grid), converted by a few threads
    const u32 nt = std::min<u32>(16, std::max<u32>(1, std::thread::hardware_concurrency()));
    const u32 per = (take + nt - 1) / nt;
    std::vector<std::thread> th;
    for (u32 t = 1; t < nt; ++t) th.emplace_back(convert, std::min(take, t * per), std::min(take, (t + 1) * per));
    convert(0, std::min(take, per));
    for (auto &t : th) t.join();
  }
  if (n) *n = take;
  const u32 dropped = found - take;
  if (lost) *lost = dropped;
  if (dropped) {
    char msg[160];
    snprintf(msg, sizeof msg, "Error: %u items lost\nHint: Search with less threads (-g) or increase dp (-d)", dropped);
    return fail(KG_ERR_LOST, msg);
  }
  return KG_OK;
}

extern "C" int kg_collect(kg_engine *e, kg_item *items, uint32_t cap, uint32_t *n, uint32_t *lost) {
  if (!e || (!items && cap)) return fail(KG_ERR_ARG, "kg_collect: bad argument");
  if (n) *n = 0;
  if (lost) *lost = 0;
  int b = e->pending;
  if (b < 0) {
    if (e->qn == 0) return fail(KG_ERR_STATE, "GPUEngine: Launch: no launch in flight");
    if (bind(e) != KG_OK || wait_oldest(e, &b) != KG_OK) return KG_ERR_HIP;
  }
  e->pending = -1;
  return unpack(e, b, items, cap, n, lost);
}

extern "C" int kg_launch(kg_engine *e, kg_item *items, uint32_t cap, uint32_t *n, uint32_t *lost) {
  if (!e || (!items && cap)) return fail(KG_ERR_ARG, "kg_launch: bad argument");
  if (n) *n = 0;
  if (lost) *lost = 0;
  int b = e->pending;
  int rk;
  if (b >= 0) {  // completed by kg_kick_and_wait
    e->pending = -1;
    rk = kick_depth(e, 0);
  } else {
    if (e->qn == 0) return fail(KG_ERR_STATE, "GPUEngine: Launch: no launch in flight");
    // Queue the next launch BEFORE waiting for the one in flight: the GPU goes from one walk kernel straight into the
    // next, and the history kernel of the launch awaited here runs underneath it on the second stream.  The next
    // launch uses the other launch buffer, whose items the previous call has already unpacked.
    rk = e->qn == 1 ? kick_depth(e, 1) : KG_OK;
    if (bind(e) != KG_OK || wait_oldest(e, &b) != KG_OK) return KG_ERR_HIP;
  }
  const int ru = unpack(e, b, items, cap, n, lost);
  return ru != KG_OK ? ru : rk;
}

extern "C" int kg_kick_and_wait(kg_engine *e) {
  const int r = kg_kick(e);
  if (r != KG_OK) return r;
  int b = -1;
  if (wait_oldest(e, &b) != KG_OK) return KG_ERR_HIP;
  e->pending = b;
  return KG_OK;
}

extern "C" int kg_dp_capacity(const kg_engine *e, uint32_t *items) {
  if (!e || !items) return fail(KG_ERR_ARG, "kg_dp_capacity: bad argument");
  *items = std::max(std::max(e->ring_cap[0], e->ring_cap[1]), ring_want(e));
  return KG_OK;
}

extern "C" int kg_last_sum_kernel_ms(const kg_engine *e, float *ms) {
  if (!e || !ms) return fail(KG_ERR_ARG, "kg_last_sum_kernel_ms: bad argument");
  *ms = e->last_sum_ms;
  return KG_OK;
}

extern "C" int kg_last_kernel_ms(const kg_engine *e, float *ms) {
  if (!e || !ms) return fail(KG_ERR_ARG, "kg_last_kernel_ms: bad argument");
  *ms = e->last_ms;
  return KG_OK;
}

extern "C" int kg_last_kernel_clock(const kg_engine *e, uint64_t *shader_cycles, uint64_t *ref_ticks_100mhz) {
  if (!e || !shader_cycles || !ref_ticks_100mhz) return fail(KG_ERR_ARG, "kg_last_kernel_clock: bad argument");
  *shader_cycles = e->last_clk;
  *ref_ticks_100mhz = e->last_ref;
  return KG_OK;
}

extern "C" int kg_kernel_name(const kg_engine *e, char *buf, size_t buflen) {
  if (!e || !buf || buflen == 0) return fail(KG_ERR_ARG, "kg_kernel_name: bad argument");
  const int dw = e->params_set ? pick_delta_words(e) : e->DW;
  if (e->streaming) {
    snprintf(buf, buflen, "kg_walk_kernel_stream<%d>", KG_STREAM_W);
    return KG_OK;
  }
  const Variant *v = find_variant(e->K, dw);
  if (!v) return fail(KG_ERR_STATE, "kg_kernel_name: no kernel variant for this configuration");
  snprintf(buf, buflen, "kg_walk_kernel_rot<%d,%d,%d,%d>", v->K, v->U, v->DW, v->W);
  return KG_OK;
}

extern "C" int kg_kernel_info(const kg_engine *e, int *k_per_lane, int *delta_words, int *lds_bytes, int *grid) {
  if (!e) return fail(KG_ERR_ARG, "kg_kernel_info: null engine");
  const int dw = e->params_set ? pick_delta_words(e) : e->DW;
  if (k_per_lane) *k_per_lane = e->K;
  if (delta_words) *delta_words = dw;
  if (lds_bytes) *lds_bytes = (int)(e->streaming ? walk_stream_lds_bytes() : walk_rot_lds_bytes(e->K, dw));
  if (grid) *grid = (int)e->grid;
  return KG_OK;
}

// ---- on-device known-answer test of the field arithmetic -----------------------------------------------------
extern "C" int kg_field_kat(int gpu_id, int op, const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *r,
                            uint32_t n) {
  if (!a || !b || !c || !r) return fail(KG_ERR_ARG, "kg_field_kat: null argument");
  if (op < 0 || op >= KAT_NOPS) return fail(KG_ERR_ARG, "kg_field_kat: unknown operation");
  if (n == 0) return KG_OK;
  int cnt = 0;
  if (kg_device_count(&cnt) != KG_OK) return KG_ERR_HIP;
  if (cnt == 0) return fail(KG_ERR_NODEVICE, "GPUEngine: There are no available device(s) that support HIP");
  if (gpu_id < 0 || gpu_id >= cnt) return fail(KG_ERR_ARG, "kg_field_kat: invalid gpuId");
  HIPCHK(hipSetDevice(gpu_id));
  const size_t bytes = (size_t)n * 8 * sizeof(u32);
  u32 *dev = nullptr;
  HIPCHK(hipMalloc((void **)&dev, 4 * bytes));
  int rc = KG_OK;
  hipError_t he = hipMemcpy(dev, a, bytes, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(dev + 8 * (size_t)n, b, bytes, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(dev + 16 * (size_t)n, c, bytes, hipMemcpyHostToDevice);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(kg_kat_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, dev, dev + 8 * (size_t)n,
                       dev + 16 * (size_t)n, dev + 24 * (size_t)n, n);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipMemcpy(r, dev + 24 * (size_t)n, bytes, hipMemcpyDeviceToHost);
  if (he != hipSuccess) rc = fail(KG_ERR_HIP, std::string("kg_field_kat: ") + hipGetErrorString(he));
  (void)hipFree(dev);
  return rc;
}
