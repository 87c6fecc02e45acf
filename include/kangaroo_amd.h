/*
 * kangaroo_amd.h — C ABI of the MI355X (gfx950) kangaroo jump engine.
 *
 * This is the drop-in boundary: one entry point per member of the reference's
 * `class GPUEngine` (math/gpu/engine.h:13-54 under /root/reference).  Plain C types only:
 * every 256-bit value is four little-endian uint64_t limbs (the reference's
 * Int::bits64[0..3]); kangaroos are indexed in the reference's host order
 *     idx = block * (128 * threadsPerGroup) + g * threadsPerGroup + t,   type = idx % 2
 * (math/gpu/engine.cu:281-326, math/gpu/compute.h:58).  How the engine lays the herd out in
 * HBM is private.  kangaroo_amd/dropin/GPUEngine_amd.cpp implements the C++ class on top of these
 * calls; INTEGRATION.md shows how the reference links against it.
 *
 * All functions return KG_OK (0) or a negative KG_ERR_* code; kg_last_error() gives the
 * message for the calling thread.  An engine is bound to one device and must be driven from
 * one host thread at a time (one engine per GPU thread, as core/kangaroo.cpp:397-508 does).
 */
#ifndef KANGAROO_AMD_H
#define KANGAROO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KG_OK 0
#define KG_ERR_ARG (-1)      /* bad argument */
#define KG_ERR_HIP (-2)      /* HIP runtime error (message has the HIP string) */
#define KG_ERR_STATE (-3)    /* call out of order (e.g. collect without kick) */
#define KG_ERR_LOST (-4)     /* more DPs than the output buffer holds; *lost tells how many */
#define KG_ERR_NODEVICE (-5) /* no usable GPU */

#define KG_GROUP_SIZE 128 /* GPU_GRP_SIZE, constants.h:15 */
#define KG_NB_JUMP 64     /* NB_JUMP, constants.h:12 */
#define KG_NB_RUN 256     /* NB_RUN, constants.h:21: default steps per launch */

typedef struct kg_engine kg_engine;

/* One distinguished point.  Replaces the 76-byte device item of math/gpu/kernel.h:154-173 and
 * the host ITEM of math/cpu/int.h:339-343 (x, d, kIdx). d already has the wild offset removed
 * (math/gpu/engine.cu:534). */
typedef struct {
  uint64_t x[4];
  uint64_t d[4];
  uint64_t kIdx;
} kg_item;

/* Number of usable devices.  Replaces the cudaGetDeviceCount checks of engine.cu:74-86. */
int kg_device_count(int *count);

/* Default launch geometry when *x <= 0 or *y <= 0.  Replaces GPUEngine::GetGridSize
 * (math/gpu/engine.cu:191-219); the defaults are CU-count based instead of SM/core tables.
 * Safe to call from any thread before an engine exists (core/kangaroo.cpp:712). */
int kg_default_grid(int gpu_id, int *x, int *y);

/* Replaces GPUEngine::GPUEngine (math/gpu/engine.cu:67-170): binds device gpu_id in the calling
 * thread and allocates for 128 * groups * per_group kangaroos.  max_found is the caller's DP
 * capacity hint (core/kangaroo.cpp:408 passes 131072); the engine never allocates less, and sizes its
 * DP ring from the configuration: twice the items one launch is expected to report
 * (kangaroos * nb_run / 2^dpBits), re-derived by kg_set_params / kg_set_nb_run, so that a default grid at
 * DP = 8 loses nothing where the reference exits with "items lost" (engine.cu:99,506-510). */
int kg_create(int gpu_id, int groups, int per_group, uint32_t max_found, kg_engine **out);
/* kg_create with explicit kernel tuning (the reference's counterpart is the compile-time choice of
 * GPU_GRP_SIZE / -maxrregcount, constants.h:15 and Makefile:11-14; there is no runtime knob there).
 * tune == NULL is kg_create.  Used by the parity tests and the throughput sweep to pin a kernel form. */
#define KG_MODE_AUTO 0   /* streaming kernel (herds beyond 44.7 M kangaroos, its 4 GiB workspace, run as several launch segments) */
#define KG_MODE_STREAM 1 /* the same, explicitly */
#define KG_MODE_REG 2    /* register-resident kernel (herd in VGPRs for the whole launch) */
typedef struct {
  int mode;       /* KG_MODE_* */
  int k_per_lane; /* kangaroos per lane; 0 = engine's choice.  Streaming: even, 2..1024; register-resident: 16 */
  uint32_t ring_max_items; /* upper bound of the DP ring per launch in items (80 bytes of pinned host memory each, two
                              buffers); 0 = the default, 2^25.  Never below the caller's max_found. */
  uint64_t ws_limit_bytes; /* streaming kernel: most workspace bytes (96 per kangaroo) one kernel launch may address; 0 = the
                              hardware's 4 GiB (one buffer descriptor).  A herd that needs more is walked in several launch
                              segments per Launch.  Tests set it low to run segments on small herds. */
} kg_tuning;
int kg_create_tuned(int gpu_id, int groups, int per_group, uint32_t max_found, const kg_tuning *tune, kg_engine **out);
/* Replaces GPUEngine::~GPUEngine (math/gpu/engine.cu:172-181). */
int kg_destroy(kg_engine *e);

/* "GPU #%d %s (%dx%d cores) Grid(%dx%d)" — GPUEngine::deviceName (math/gpu/engine.cu:101-107). */
int kg_device_name(const kg_engine *e, char *buf, size_t buflen);
/* Replaces GPUEngine::GetMemory (engine.cu:183-185), without the int overflow. */
int kg_memory_bytes(const kg_engine *e, uint64_t *bytes);
/* Replaces GPUEngine::GetNbThread (engine.cu:277-279): groups * per_group. */
int kg_nb_thread(const kg_engine *e);
/* Replaces GPUEngine::GetGroupSize (engine.cu:187-189): always 128. */
int kg_group_size(const kg_engine *e);
/* 128 * groups * per_group */
uint64_t kg_nb_kangaroo(const kg_engine *e);

/* Replaces GPUEngine::SetWildOffset (engine.cu:63-65): offset added (mod n) to the distance of
 * odd-index kangaroos on upload and removed on download / DP output. */
int kg_set_wild_offset(kg_engine *e, const uint64_t offset[4]);

/* Replaces GPUEngine::SetParams (engine.cu:442-472): DP mask on x.bits64[3] and the 64-entry
 * jump table, each array 64 x 4 limbs. */
int kg_set_params(kg_engine *e, uint64_t dp_mask, const uint64_t *jump_dist, const uint64_t *jump_px,
                  const uint64_t *jump_py);

/* Replaces GPUEngine::SetKangaroos (engine.cu:281-326): n = kg_nb_kangaroo(e) entries of 4 limbs
 * each, host index order; x, y must be canonical affine coordinates. */
int kg_set_kangaroos(kg_engine *e, const uint64_t *px, const uint64_t *py, const uint64_t *d, uint64_t n);
/* Herd creation on the GPU: P_i = d_i * G, plus the search key (key_x, key_y) for odd (wild) i; then as
 * kg_set_kangaroos with those points.  Replaces the CPU point computation of Kangaroo::CreateHerd
 * (core/kangaroo.cpp:549-571: ComputePublicKeys + AddDirect) in front of GPUEngine::SetKangaroos; d as
 * produced by kgh_herd_distances or the reference's own loop (core/kangaroo.cpp:531-548).  Pass an all-zero
 * key to get plain d_i * G for every i.  The points can be read back with kg_get_kangaroos. */
int kg_herd_from_distances(kg_engine *e, const uint64_t *d, uint64_t n, const uint64_t key_x[4],
                           const uint64_t key_y[4]);
/* Replaces GPUEngine::GetKangaroos (engine.cu:328-383).  Waits for the launch in flight. */
int kg_get_kangaroos(kg_engine *e, uint64_t *px, uint64_t *py, uint64_t *d, uint64_t n);
/* Replaces GPUEngine::SetKangaroo (engine.cu:385-425): overwrite one kangaroo.  Ordered on the
 * engine's stream: takes effect after the launch in flight and before the next one. */
int kg_set_kangaroo(kg_engine *e, uint64_t k_idx, const uint64_t px[4], const uint64_t py[4],
                    const uint64_t d[4]);

/* Steps per launch (default KG_NB_RUN).  No counterpart in the reference, where NB_RUN is a
 * compile-time constant (constants.h:21); used by the throughput sweep of BASELINE config 4. */
int kg_set_nb_run(kg_engine *e, uint32_t nb_run);

/* Replaces GPUEngine::callKernel (engine.cu:427-440): clear the DP counter and start one launch
 * of nb_run steps for every kangaroo, asynchronously. */
int kg_kick(kg_engine *e);
/* kg_kick for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same launch as kg_kick, but the streaming kernel's
 * workspace (kangaroo slots, jump-index history, pacing table) is a static device array of
 * KG_STATIC_WORKSPACE_BYTES inside the library instead of a buffer allocated at the first kg_kick; this call
 * performs no hipMalloc / hipFree.  Returns KG_ERR_ARG if the engine's herd and step count need more than that
 * (slots 96 * S + history nb_run * S + DP chain heads 4 * S bytes for S = kangaroos rounded up to 256 lanes of an
 * even number >= 4 per lane, + 64 KiB pacing table + 8 bytes per item of the DP ring: 65 536 kangaroos at NB_RUN
 * 256) or if the engine runs the register-resident kernel, and KG_ERR_STATE if the DP ring would have to grow first
 * (kg_set_params / kg_set_nb_run size it while no launch is in flight).  Pair it with kg_collect (kg_launch would queue an ordinary kg_kick).  All engines
 * of one GPU share the array: launches made through this call must not overlap on one GPU (they would race);
 * the array keeps the last launch's contents, what must start at zero is cleared per call. */
#define KG_STATIC_WORKSPACE_BYTES (32u << 20)
int kg_kick_static(kg_engine *e);
/* First half of GPUEngine::Launch (engine.cu:486-537): wait for the launch in flight and copy out
 * its DPs.  *n receives the item count stored (<= cap).  If the launch produced more than the
 * engine's buffer or cap can hold, returns KG_ERR_LOST with *lost = number dropped. */
int kg_collect(kg_engine *e, kg_item *items, uint32_t cap, uint32_t *n, uint32_t *lost);
/* GPUEngine::Launch (engine.cu:486-539): collect the launch in flight, start the next one.  The next launch is
 * queued BEFORE the call waits for the one in flight (two launch buffers), so the GPU goes from one walk kernel
 * straight into the next and host work overlaps it.  As in the reference, when the call returns the next launch is
 * already under way: a kg_set_kangaroo issued now takes effect for the launch after that one. */
int kg_launch(kg_engine *e, kg_item *items, uint32_t cap, uint32_t *n, uint32_t *lost);
/* Items one launch can return: the capacity kg_collect / kg_launch need in `items` to take everything the DP ring
 * holds under the current configuration (it follows kg_set_params / kg_set_nb_run).  The reference's counterpart is
 * the maxFound the caller itself passed (engine.cu:99). */
int kg_dp_capacity(const kg_engine *e, uint32_t *items);
/* Replaces GPUEngine::callKernelAndWait (engine.cu:474-484): kick + wait, DPs stay queued for the
 * next kg_collect. */
int kg_kick_and_wait(kg_engine *e);

/* Duration in milliseconds of the last completed launch's walk kernel, measured with HIP events on the
 * engine's stream (bench.py's roofline figure).  No counterpart in the reference. */
int kg_last_kernel_ms(const kg_engine *e, float *ms);
/* The same for the launch's history kernel (distance sums and the distances of the launch's DP items), which runs on
 * the engine's second stream underneath the next launch's walk kernel; 0 for the register-resident kernel. */
int kg_last_sum_kernel_ms(const kg_engine *e, float *ms);
/* Clock the last completed launch of the streaming kernel really ran at: shader clock cycles and ticks of the
 * constant 100 MHz reference counter between the start and the end of workgroup 0 (s_memtime / s_memrealtime);
 * MHz = 100 * cycles / ticks.  Zeros for the register-resident kernel.  No counterpart in the reference. */
int kg_last_kernel_clock(const kg_engine *e, uint64_t *shader_cycles, uint64_t *ref_ticks_100mhz);
/* Kernel variant in use: kangaroos per lane, distance-delta words, LDS bytes per workgroup,
 * workgroups launched.  No counterpart in the reference. */
int kg_kernel_info(const kg_engine *e, int *k_per_lane, int *delta_words, int *lds_bytes, int *grid);

/* Name of the kernel instantiation a launch will use, as it appears in a rocprofv3 kernel trace
 * (template arguments, register-resident kernel: kangaroos per lane, window, delta words, waves per SIMD;
 * streaming kernel: waves per SIMD).  No counterpart in
 * the reference. */
int kg_kernel_name(const kg_engine *e, char *buf, size_t buflen);

/* Known-answer test of the DEVICE field arithmetic: n operations `op` (0 mul a*b, 1 sqr a^2, 2 sub a-b,
 * 3 sub2 a-b-c, 4 (a-b)*c, 5 a mod p, 6 a^-1 (0 for 0), 7 ((a*b)*c)^2 through weak intermediates, 8 weak(a-b)*c) on
 * operands of 8 little-endian 32-bit words each (any 256-bit value, also >= p); results canonical.  Operations 9
 * and 10 take nine raw 29-bit limbs (a[0..7], b[0]) and run the settle + pack + unpack path of the streaming
 * kernel's workspace format (fe_settle<0,1> / fe_settle<2,2>).  The GPU-side
 * counterpart of the reference's Int::Check (math/cpu/int.cpp:1158-1511), which is CPU only. */
int kg_field_kat(int gpu_id, int op, const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *r, uint32_t n);

/* Replaces GPUEngine::AllocatePinnedMemory / FreePinnedMemory (engine.cu:221-233). */
void *kg_alloc_pinned(size_t size);
void kg_free_pinned(void *p);
/* PCI address of device gpu_id as "dddd:bb:dd.f" (what PrintCudaInfo's cudaDeviceProp carries, engine.cu:249-262):
 * lets a caller find the device's sysfs entry (power, clocks) without assuming that DRM card order is HIP order. */
int kg_device_pci_bus_id(int gpu_id, char *buf, size_t buflen);
/* Replaces GPUEngine::PrintCudaInfo (engine.cu:235-275): one line per device on stdout. */
int kg_print_device_info(void);

const char *kg_last_error(void);
const char *kg_version(void);

#ifdef __cplusplus
}
#endif
#endif
