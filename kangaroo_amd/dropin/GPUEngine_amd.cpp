// GPUEngine_amd.cpp — the reference's `class GPUEngine` implemented on the MI355X jump engine.
//
// This translation unit is compiled INSIDE the reference's source tree in place of
// math/gpu/engine.cu: it includes the reference's own, unmodified "engine.h"
// (math/gpu/engine.h:13-54) and defines every member declared there by forwarding to the C ABI of
// libkangaroo_amd.so (include/kangaroo_amd.h).  core/kangaroo.cpp and core/thread.cpp compile and
// link unchanged.  The class layout is the reference's; the only field this file uses for itself is
// the opaque `void *resultReady` (a CUDA event handle in the reference, engine.h:45), which holds the
// kg_engine handle here.  See INTEGRATION.md for the build line.
//
// Behaviour kept from the reference: no exceptions, bool results, "GPUEngine: ..." diagnostics on
// stdout, exit(1) when distinguished points were lost (engine.cu:506-510).
#include "engine.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kangaroo_amd.h"

namespace {

inline kg_engine *handle_of(void *p) { return static_cast<kg_engine *>(p); }

bool report(int rc, const char *what) {
  if (rc == KG_OK) return true;
  printf("GPUEngine: %s: %s\n", what, kg_last_error());
  return false;
}

// Int (math/cpu/int.h:165-168) carries five 64-bit limbs; the engine speaks four.
void pack4(uint64_t *dst, Int *src, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) memcpy(dst + 4 * i, src[i].bits64, 32);
}
void unpack4(Int *dst, const uint64_t *src, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) {
    memcpy(dst[i].bits64, src + 4 * i, 32);
    dst[i].bits64[4] = 0;
  }
}

}  // namespace

GPUEngine::GPUEngine(int nbThreadGroup, int nbThreadPerGroup, int gpuId, uint32_t maxFound) {
  initialised = false;
  lostWarning = false;
  resultReady = nullptr;
  inputKangaroo = inputKangarooPinned = jumpPinned = nullptr;
  outputItem = outputItemPinned = nullptr;
  nbThread = nbThreadGroup * nbThreadPerGroup;
  this->nbThreadPerGroup = nbThreadPerGroup;
  this->maxFound = maxFound;
  outputSize = kangarooSize = kangarooSizePinned = jumpSize = 0;
  dpMask = 0;
  wildOffset.SetInt32(0);

  kg_engine *h = nullptr;
  if (!report(kg_create(gpuId, nbThreadGroup, nbThreadPerGroup, maxFound, &h), "kg_create")) return;
  resultReady = h;
  char name[512];
  kg_device_name(h, name, sizeof(name));
  deviceName = name;
  initialised = true;
}

GPUEngine::~GPUEngine() {
  if (resultReady) kg_destroy(handle_of(resultReady));
  resultReady = nullptr;
}

int GPUEngine::GetMemory() {
  uint64_t b = 0;
  if (resultReady) kg_memory_bytes(handle_of(resultReady), &b);
  return b > 0x7FFFFFFFull ? 0x7FFFFFFF : (int)b;  // the reference's return type is int
}
int GPUEngine::GetNbThread() { return nbThread; }
int GPUEngine::GetGroupSize() { return KG_GROUP_SIZE; }

void GPUEngine::SetWildOffset(Int *offset) {
  wildOffset.Set(offset);
  if (resultReady) report(kg_set_wild_offset(handle_of(resultReady), offset->bits64), "SetWildOffset");
}

void GPUEngine::SetParams(uint64_t dpMask, Int *distance, Int *px, Int *py) {
  this->dpMask = dpMask;
  if (!resultReady) return;
  uint64_t jd[KG_NB_JUMP * 4], jx[KG_NB_JUMP * 4], jy[KG_NB_JUMP * 4];
  pack4(jd, distance, KG_NB_JUMP);
  pack4(jx, px, KG_NB_JUMP);
  pack4(jy, py, KG_NB_JUMP);
  report(kg_set_params(handle_of(resultReady), dpMask, jd, jx, jy), "SetParams");
}

void GPUEngine::SetKangaroos(Int *px, Int *py, Int *d) {
  if (!resultReady) return;
  const uint64_t n = kg_nb_kangaroo(handle_of(resultReady));
  uint64_t *buf = (uint64_t *)malloc(n * 32 * 3);
  if (!buf) {
    printf("GPUEngine: SetKangaroos: out of host memory\n");
    return;
  }
  pack4(buf, px, n);
  pack4(buf + 4 * n, py, n);
  pack4(buf + 8 * n, d, n);
  report(kg_set_kangaroos(handle_of(resultReady), buf, buf + 4 * n, buf + 8 * n, n), "SetKangaroos");
  free(buf);
}

void GPUEngine::GetKangaroos(Int *px, Int *py, Int *d) {
  if (!resultReady) return;
  const uint64_t n = kg_nb_kangaroo(handle_of(resultReady));
  uint64_t *buf = (uint64_t *)malloc(n * 32 * 3);
  if (!buf) {
    printf("GPUEngine: GetKangaroos: out of host memory\n");
    return;
  }
  if (report(kg_get_kangaroos(handle_of(resultReady), buf, buf + 4 * n, buf + 8 * n, n), "GetKangaroos")) {
    unpack4(px, buf, n);
    unpack4(py, buf + 4 * n, n);
    unpack4(d, buf + 8 * n, n);
  }
  free(buf);
}

void GPUEngine::SetKangaroo(uint64_t kIdx, Int *px, Int *py, Int *d) {
  if (!resultReady) return;
  report(kg_set_kangaroo(handle_of(resultReady), kIdx, px->bits64, py->bits64, d->bits64), "SetKangaroo");
}

bool GPUEngine::callKernel() {
  if (!resultReady) return false;
  return report(kg_kick(handle_of(resultReady)), "Kernel");
}

bool GPUEngine::callKernelAndWait() {
  if (!resultReady) return false;
  return report(kg_kick_and_wait(handle_of(resultReady)), "Kernel");
}

bool GPUEngine::Launch(std::vector<ITEM> &hashFound, bool spinWait) {
  // spinWait (engine.cu:486-500: poll the event in a loop instead of blocking on it) is accepted and has no effect:
  // the engine always blocks on its stream's event, which costs no CPU while the GPU walks
  (void)spinWait;
  hashFound.clear();
  if (!resultReady) return false;
  // staging for one launch's items, per host thread (one engine per thread, core/kangaroo.cpp:397-408); sized from
  // the engine's DP ring, which follows the DP mask (kg_dp_capacity), never below the caller's maxFound
  static thread_local std::vector<kg_item> items;
  uint32_t cap = maxFound;
  if (!report(kg_dp_capacity(handle_of(resultReady), &cap), "Launch")) return false;
  if (cap < maxFound) cap = maxFound;
  if (items.size() < cap) items.resize(cap);
  uint32_t n = 0, lost = 0;
  const int rc = kg_launch(handle_of(resultReady), items.data(), cap, &n, &lost);
  if (rc == KG_ERR_LOST) {
    // same consequence as engine.cu:506-510: tell the user how to avoid it and stop
    printf("GPUEngine: %u distinguished points lost, use a larger -d (dpSize) or a smaller grid\n", lost);
    exit(1);
  }
  if (!report(rc, "Launch")) return false;
  hashFound.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    ITEM &it = hashFound[i];
    memcpy(it.x.bits64, items[i].x, 32);
    it.x.bits64[4] = 0;
    memcpy(it.d.bits64, items[i].d, 32);
    it.d.bits64[4] = 0;
    it.kIdx = items[i].kIdx;
  }
  return true;
}

void *GPUEngine::AllocatePinnedMemory(size_t size) { return kg_alloc_pinned(size); }
void GPUEngine::FreePinnedMemory(void *buff) { kg_free_pinned(buff); }
void GPUEngine::PrintCudaInfo() { kg_print_device_info(); }

bool GPUEngine::GetGridSize(int gpuId, int *x, int *y) {
  return report(kg_default_grid(gpuId, x, y), "GetGridSize");
}
