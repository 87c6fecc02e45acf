// Micro-benchmark: sustained issue rate, shader clock and board power of single VALU instructions on gfx950, each
// run for about a second at 2 waves per SIMD on every CU — long enough for the power management to settle, which the
// millisecond runs of tools/ubench_instr_table.hip are not.  Tells whether a kernel made of these instructions is
// limited by issue slots or by the board's power cap, and what one instruction of each kind costs in energy.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ab/ubench_power tools/ubench_power.hip
// Power is read from the amdgpu hwmon file (read-only); the clock is s_memtime against the 100 MHz s_memrealtime.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
typedef uint32_t u32;
typedef uint64_t u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define REP8(L) L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
// eight independent chains per lane, 8 instructions per asm block, 16 blocks per loop iteration
#define KERNEL(NAME, LINE)                                                                                     \
  __global__ void __launch_bounds__(256, 2) NAME(u64 *out, u32 seed, u32 iters, u64 *clk) {                    \
    u64 c[8];                                                                                                  \
    for (int i = 0; i < 8; ++i) c[i] = ((u64)(threadIdx.x * 2654435761u + seed + i) << 20) | 12345u;           \
    u32 b = (threadIdx.x * 40503u + seed) | 1u, d = b ^ 0x5bd1e995u;                                           \
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                        \
    for (u32 it = 0; it < iters; ++it) {                                                                       \
      _Pragma("unroll") for (int rep = 0; rep < 16; ++rep)                                                     \
        asm volatile(REP8(LINE)                                                                                \
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) \
                     : "v"(b), "v"(d)                                                                          \
                     : "s40", "s41", "vcc");                                                                   \
    }                                                                                                          \
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                        \
    u64 x = 0;                                                                                                 \
    for (int i = 0; i < 8; ++i) x ^= c[i];                                                                     \
    out[blockIdx.x * 256 + threadIdx.x] = x;                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                           \
  }
// %N is a 64-bit register pair; %N's low half is written %L… via the sub-register syntax is not available in
// inline asm, so 32-bit instructions operate on the pair's low register through v_..._e32 on "%N" (the assembler
// takes the first register of the pair when the operand is given as a 64-bit value in a 32-bit slot is NOT
// legal) — hence 32-bit instructions get their own kernels with u32 chains below.
#define L_MAD(i) "v_mad_u64_u32 %" #i ", s[40:41], %8, %9, %" #i "\n\t"
#define L_SHR64(i) "v_lshrrev_b64 %" #i ", 1, %" #i "\n\t"
#define L_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 1, %" #i "\n\t"
KERNEL(k_mad_u64_u32, L_MAD)
KERNEL(k_lshrrev_b64, L_SHR64)
KERNEL(k_lshl_add_u64, L_LSHLADD64)

#define KERNEL32(NAME, LINE)                                                                                   \
  __global__ void __launch_bounds__(256, 2) NAME(u64 *out, u32 seed, u32 iters, u64 *clk) {                    \
    u32 c[8];                                                                                                  \
    for (int i = 0; i < 8; ++i) c[i] = threadIdx.x * 2654435761u + seed + i;                                   \
    u32 b = (threadIdx.x * 40503u + seed) | 1u, d = b ^ 0x5bd1e995u;                                           \
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                        \
    for (u32 it = 0; it < iters; ++it) {                                                                       \
      _Pragma("unroll") for (int rep = 0; rep < 16; ++rep)                                                     \
        asm volatile(REP8(LINE)                                                                                \
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) \
                     : "v"(b), "v"(d)                                                                          \
                     : "vcc");                                                                                 \
    }                                                                                                          \
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                        \
    u32 x = 0;                                                                                                 \
    for (int i = 0; i < 8; ++i) x ^= c[i];                                                                     \
    out[blockIdx.x * 256 + threadIdx.x] = x;                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                           \
  }
#define L_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define L_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define L_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define L_SHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n\t"
#define L_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 29\n\t"
#define L_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define L_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define L_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n\t"
#define L_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n\t"
#define L_MOV(i) "v_mov_b32 %" #i ", %8\n\t"
#define L_NOP(i) "s_nop 0\n\t"
KERNEL32(k_add_u32, L_ADD)
KERNEL32(k_and_b32, L_AND)
KERNEL32(k_xor_b32, L_XOR)
KERNEL32(k_lshrrev_b32, L_SHR)
KERNEL32(k_alignbit_b32, L_ALIGN)
KERNEL32(k_add3_u32, L_ADD3)
KERNEL32(k_mul_lo_u32, L_MULLO)
KERNEL32(k_mul_hi_u32, L_MULHI)
KERNEL32(k_mad_u32_u24, L_MAD24)
KERNEL32(k_mov_b32, L_MOV)
KERNEL32(k_s_nop, L_NOP)

// The rows above feed every instruction of a chain the SAME two multiplier operands (%8, %9): the multiplier array of a
// lane sees no change from one instruction to the next, only the adder does.  The rows below give the eight
// instructions of a block eight different operand pairs (random 32-bit values, random 52-bit mantissas), as a field
// multiplication does, for the question "what does a 29 x 29-bit integer multiply-add cost against a double-precision
// FMA that could carry a 52 x 52-bit product" (DESIGN.md section 3.1, FP64 multiplier).
#define MADV(c, x, y) "v_mad_u64_u32 %" #c ", s[40:41], %" #x ", %" #y ", %" #c "\n\t"
#define KERNEL_MADV(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                      \
  __global__ void __launch_bounds__(256, 2) NAME(u64 *out, u32 seed, u32 iters, u64 *clk) {                    \
    u64 c[8];                                                                                                  \
    u32 x[8];                                                                                                  \
    for (int i = 0; i < 8; ++i) {                                                                              \
      c[i] = ((u64)(threadIdx.x * 2654435761u + seed + i) << 20) | 12345u;                                     \
      x[i] = (threadIdx.x * 40503u + seed * 7919u + i * 0x9E3779B1u) * 2246822519u | 1u;                       \
    }                                                                                                          \
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                        \
    for (u32 it = 0; it < iters; ++it) {                                                                       \
      _Pragma("unroll") for (int rep = 0; rep < 16; ++rep)                                                     \
        asm volatile(I0 I1 I2 I3 I4 I5 I6 I7                                                                   \
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) \
                     : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7])  \
                     : "s40", "s41", "vcc");                                                                   \
    }                                                                                                          \
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                        \
    u64 xo = 0;                                                                                                \
    for (int i = 0; i < 8; ++i) xo ^= c[i];                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = xo;                                                                  \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                           \
  }
// both operands change from one instruction to the next (a column of a field multiplication)
KERNEL_MADV(k_mad_varied, MADV(0, 8, 9), MADV(1, 10, 11), MADV(2, 12, 13), MADV(3, 14, 15), MADV(4, 9, 12), MADV(5, 11, 14),
            MADV(6, 13, 8), MADV(7, 15, 10))
// neighbours share one operand, alternately the first and the second (a zig-zag through two columns at once)
KERNEL_MADV(k_mad_zigzag, MADV(0, 8, 9), MADV(1, 8, 10), MADV(2, 11, 10), MADV(3, 11, 12), MADV(4, 13, 12), MADV(5, 13, 14),
            MADV(6, 15, 14), MADV(7, 15, 9))
// the first operand stays, the second changes (a row of the product: operand scanning)
KERNEL_MADV(k_mad_fix0, MADV(0, 8, 9), MADV(1, 8, 10), MADV(2, 8, 11), MADV(3, 8, 12), MADV(4, 8, 13), MADV(5, 8, 14),
            MADV(6, 8, 15), MADV(7, 8, 9))
// the second operand stays, the first changes
KERNEL_MADV(k_mad_fix1, MADV(0, 9, 8), MADV(1, 10, 8), MADV(2, 11, 8), MADV(3, 12, 8), MADV(4, 13, 8), MADV(5, 14, 8),
            MADV(6, 15, 8), MADV(7, 9, 8))
// both operands change, ONE accumulator chain (what a column really is: eight dependent multiply-adds)
KERNEL_MADV(k_mad_chain, MADV(0, 8, 9), MADV(0, 10, 11), MADV(0, 12, 13), MADV(0, 14, 15), MADV(0, 9, 12), MADV(0, 11, 14),
            MADV(0, 13, 8), MADV(0, 15, 10))
// doubles with random 52-bit mantissas in [2^51, 2^52): products near 2^103, accumulators start at 2^104 (the
// alignment constant of the hi/lo product trick) and stay finite for the length of the run
#define KERNEL_F64(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                       \
  __global__ void __launch_bounds__(256, 2) NAME(u64 *out, u32 seed, u32 iters, u64 *clk) {                    \
    double c[8], x[8];                                                                                         \
    for (int i = 0; i < 8; ++i) {                                                                              \
      const u64 m = ((u64)((threadIdx.x * 40503u + seed * 7919u + i * 0x9E3779B1u) * 2246822519u) << 20) ^     \
                    ((u64)(threadIdx.x * 2654435761u + i) * 0x2545F4914F6CDD1Dull >> 12);                       \
      x[i] = (double)((m & ((1ull << 51) - 1)) | (1ull << 51));                                                \
      c[i] = 20282409603651670423947251286016.0 + (double)i; /* 2^104 */                                       \
    }                                                                                                          \
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                        \
    for (u32 it = 0; it < iters; ++it) {                                                                       \
      _Pragma("unroll") for (int rep = 0; rep < 16; ++rep)                                                     \
        asm volatile(I0 I1 I2 I3 I4 I5 I6 I7                                                                   \
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) \
                     : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));  \
    }                                                                                                          \
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                        \
    double xo = 0;                                                                                             \
    for (int i = 0; i < 8; ++i) xo += c[i];                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = (u64)__double_as_longlong(xo);                                       \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                           \
  }
KERNEL_F64(k_fma_f64, "v_fma_f64 %0, %8, %9, %0\n\t", "v_fma_f64 %1, %10, %11, %1\n\t", "v_fma_f64 %2, %12, %13, %2\n\t",
           "v_fma_f64 %3, %14, %15, %3\n\t", "v_fma_f64 %4, %9, %12, %4\n\t", "v_fma_f64 %5, %11, %14, %5\n\t",
           "v_fma_f64 %6, %13, %8, %6\n\t", "v_fma_f64 %7, %15, %10, %7\n\t")
KERNEL_F64(k_add_f64, "v_add_f64 %0, %0, %8\n\t", "v_add_f64 %1, %1, %9\n\t", "v_add_f64 %2, %2, %10\n\t",
           "v_add_f64 %3, %3, %11\n\t", "v_add_f64 %4, %4, %12\n\t", "v_add_f64 %5, %5, %13\n\t",
           "v_add_f64 %6, %6, %14\n\t", "v_add_f64 %7, %7, %15\n\t")
KERNEL_F64(k_mul_f64, "v_mul_f64 %0, %8, %9\n\t", "v_mul_f64 %1, %10, %11\n\t", "v_mul_f64 %2, %12, %13\n\t",
           "v_mul_f64 %3, %14, %15\n\t", "v_mul_f64 %4, %9, %12\n\t", "v_mul_f64 %5, %11, %14\n\t",
           "v_mul_f64 %6, %13, %8\n\t", "v_mul_f64 %7, %15, %10\n\t")

// mixed: the ratio of the walk kernel's hot loop (1 multiply-add : 1 simple op), interleaved
#define L_MIX(i) "v_mad_u64_u32 %" #i ", s[40:41], %8, %9, %" #i "\n\tv_and_b32 %8, %8, %9\n\t"
__global__ void __launch_bounds__(256, 2) k_mix(u64 *out, u32 seed, u32 iters, u64 *clk) {
  u64 c[8];
  for (int i = 0; i < 8; ++i) c[i] = ((u64)(threadIdx.x * 2654435761u + seed + i) << 20) | 12345u;
  u32 b = (threadIdx.x * 40503u + seed) | 1u, d = b ^ 0x5bd1e995u;
  const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (u32 it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)
      asm volatile(REP8(L_MIX)
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(b)
                   : "v"(d)
                   : "s40", "s41", "vcc");
  }
  const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  u64 x = b;
  for (int i = 0; i < 8; ++i) x ^= c[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// streaming copy: 16 bytes per lane per access, grid-stride, the HBM side alone
__global__ void __launch_bounds__(256, 2) k_copy(const uint4 *src, uint4 *dst, size_t n, u32 passes, u64 *clk) {
  const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (u32 p = 0; p < passes; ++p)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      uint4 v = src[i];
      v.x += p;
      dst[i] = v;
    }
  const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

static std::string find_power_file() {
  for (int card = 0; card < 16; ++card)
    for (int hw = 0; hw < 16; ++hw)
      for (const char *leaf : {"power1_average", "power1_input"}) {
        char p[256];
        snprintf(p, sizeof p, "/sys/class/drm/card%d/device/hwmon/hwmon%d/%s", card, hw, leaf);
        FILE *f = fopen(p, "r");
        if (f) {
          fclose(f);
          return p;
        }
      }
  return "";
}
static double read_watts(const std::string &path) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return 0;
  long long uw = 0;
  if (fscanf(f, "%lld", &uw) != 1) uw = 0;
  fclose(f);
  return uw * 1e-6;
}

struct Sampler {
  std::string path;
  std::atomic<bool> stop{false};
  std::vector<double> w;
  std::thread th;
  void start() {
    stop = false;
    w.clear();
    th = std::thread([this] {
      while (!stop) {
        if (!path.empty()) w.push_back(read_watts(path));
        std::this_thread::sleep_for(std::chrono::milliseconds(25));
      }
    });
  }
  double finish() {  // median of the second half (settled)
    stop = true;
    th.join();
    if (w.size() < 4) return 0;
    std::vector<double> h(w.begin() + w.size() / 2, w.end());
    std::sort(h.begin(), h.end());
    return h[h.size() / 2];
  }
};

typedef void (*kern_t)(u64 *, u32, u32, u64 *);

int main(int argc, char **argv) {
  const double target_s = argc > 1 ? atof(argv[1]) : 1.2;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 2;  // 2 workgroups of 4 waves per CU = 2 waves per SIMD
  u64 *out, *clk;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
  CHECK(hipMalloc(&clk, 16));
  Sampler smp;
  smp.path = find_power_file();
  printf("# %s, %d CUs, %d workgroups x 256 threads (2 waves per SIMD), about %.1f s per row; power file: %s\n", prop.gcnArchName, cus,
         blocks, target_s, smp.path.empty() ? "(none)" : smp.path.c_str());
  smp.start();
  std::this_thread::sleep_for(std::chrono::milliseconds(600));
  const double idle = smp.finish();
  printf("# idle power %.0f W\n", idle);
  printf("%-18s %9s %9s %12s %14s %9s %12s\n", "instruction", "clock MHz", "power W", "cyc/instr", "G wave-instr/s", "seconds", "nJ/wave-instr");
  struct Row { const char *name; kern_t k; int per_iter; } rows[] = {
      {"v_mad_u64_u32", k_mad_u64_u32, 128}, {"v_add_u32", k_add_u32, 128}, {"v_and_b32", k_and_b32, 128},
      {"v_xor_b32", k_xor_b32, 128}, {"v_lshrrev_b32", k_lshrrev_b32, 128}, {"v_mov_b32", k_mov_b32, 128},
      {"v_lshrrev_b64", k_lshrrev_b64, 128}, {"v_lshl_add_u64", k_lshl_add_u64, 128}, {"v_alignbit_b32", k_alignbit_b32, 128},
      {"v_add3_u32", k_add3_u32, 128}, {"v_mul_lo_u32", k_mul_lo_u32, 128}, {"v_mul_hi_u32", k_mul_hi_u32, 128},
      {"v_mad_u32_u24", k_mad_u32_u24, 128}, {"s_nop 0", k_s_nop, 128}, {"mad64 + and (1:1)", k_mix, 128},
      {"mad_u64_u32 varied", k_mad_varied, 128}, {"mad zig-zag operands", k_mad_zigzag, 128}, {"mad 1st operand fixed", k_mad_fix0, 128},
      {"mad 2nd operand fixed", k_mad_fix1, 128}, {"mad varied, 1 chain", k_mad_chain, 128}, {"v_fma_f64 varied", k_fma_f64, 128}, {"v_mul_f64 varied", k_mul_f64, 128},
      {"v_add_f64 varied", k_add_f64, 128},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (const Row &r : rows) {
    // calibrate
    u32 iters = 2000;
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, out, 1u, iters, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    iters = (u32)std::min<double>(4.0e9, iters * target_s * 1e3 / std::max(ms, 0.01f));
    smp.start();
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, out, 2u, iters, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    const double watts = smp.finish();
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    u64 h[2];
    CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0;
    const double instr_per_wave = (double)iters * r.per_iter;
    const double total = instr_per_wave * blocks * 4;  // wave-instructions
    const double rate = total / (ms * 1e-3);
    // issue cycles per wave-instruction and SIMD, derived from the whole machine's rate (round 2 timed workgroup 0
    // alone, which the arbiter favours: 2.61 "cycles" beside a rate that means 4.51)
    const double cyc = mhz * 1e6 * (double)cus * 4.0 / rate;
    printf("%-18s %9.0f %9.0f %12.2f %14.1f %9.2f %12.2f\n", r.name, mhz, watts, cyc, rate / 1e9, ms * 1e-3,
           (watts - idle) / rate * 1e9);
    fflush(stdout);
  }
  // HBM copy
  {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    uint4 *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    u32 passes = 2;
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_copy, dim3(blocks * 2), dim3(256), 0, 0, a, b, n, passes, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    passes = (u32)std::max(2.0, passes * target_s * 1e3 / ms);
    smp.start();
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_copy, dim3(blocks * 2), dim3(256), 0, 0, a, b, n, passes, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    const double watts = smp.finish();
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    u64 h[2];
    CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double tb = 2.0 * bytes * passes / (ms * 1e-3) / 1e12;
    printf("%-18s %9.0f %9.0f   copy %.2f TB/s (read + write), %.1f pJ per byte moved\n", "HBM copy 2 GiB", h[1] ? 100.0 * h[0] / h[1] : 0.0,
           watts, tb, (watts - idle) / (tb * 1e12) * 1e12);
  }
  return 0;
}
