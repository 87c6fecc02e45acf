// Micro-benchmark of the per-lane modular inverse (kangaroo_amd/csrc/modinv30.h) on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/ubench_inv tools/ubench_inv.hip && build/ubench_inv
// Every lane inverts a chain of values (a <- a^-1 + c); the result is checked with a * a^-1 == 1.
// LDS is requested to pin the occupancy at 1 or 2 waves per SIMD, as in the jump kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../kangaroo_amd/csrc/modinv30.h"
using namespace kg;

template <int MODE, int NITER>
__global__ void __launch_bounds__(256) kinv(u32 *out, u32 seed, int iters, u64 *clk, u32 *bad) {
  extern __shared__ u32 lds[];
  fe a;
  u32 s = seed + 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
  for (int i = 0; i < 9; ++i) {
    s = s * 1664525u + 1013904223u;
    a.l[i] = (s >> 3) & (i == 8 ? 0xFFFFFFu : M29);
  }
  if (threadIdx.x == 1000) lds[0] = a.l[0];  // keep the allocation alive
  u32 nbad = 0;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    fe r, chk;
    fe_inv_t<MODE, NITER>(r, a);
    fe_mul(chk, a, r);
    fe_canon(chk);
    u32 o = chk.l[0] ^ 1u;
    for (int i = 1; i < 9; ++i) o |= chk.l[i];
    nbad += (o != 0);
    a = r;
    a.l[0] += 12345u + it;  // stays weak
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  u32 x = 0;
  for (int i = 0; i < 9; ++i) x ^= a.l[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
  if (nbad) atomicAdd(bad, nbad);
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE, int NITER>
static void run(const char *name, int waves_per_simd, int iters) {
  const int blocks = 256 * waves_per_simd;
  const size_t lds = waves_per_simd == 1 ? 100 * 1024 : 60 * 1024;
  u32 *out, *bad;
  u64 *clk;
  hipMalloc(&out, blocks * 256 * 4);
  hipMalloc(&bad, 4);
  hipMalloc(&clk, 8);
  hipMemset(bad, 0, 4);
  hipFuncSetAttribute((const void *)kinv<MODE, NITER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kinv<MODE, NITER><<<blocks, 256, lds>>>(out, 1u, 4, clk, bad);  // warm-up
  hipMemset(bad, 0, 4);
  hipEventRecord(e0);
  kinv<MODE, NITER><<<blocks, 256, lds>>>(out, 7u, iters, clk, bad);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  u64 c = 0;
  u32 b = 0;
  hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
  // the chain includes one fe_mul + fe_canon check per inversion (~230 instructions, ~1.1 k cycles)
  printf("%-22s w/SIMD=%d  %8.3f ms  memtime ticks per inversion (wave 0) = %8.0f   wall us/inv = %.2f  bad=%u\n", name,
         waves_per_simd, ms, (double)c / iters, 1e3 * ms / iters, b);
  hipFree(out);
  hipFree(bad);
  hipFree(clk);
}

int main() {
  const int iters = 200;
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>("const 20x30", w, iters);
    run<1, 0>("var loop", w, iters);
    run<2, 7>("fix NITER=7", w, iters);
    run<2, 8>("fix NITER=8", w, iters);
    run<2, 9>("fix NITER=9", w, iters);
    run<2, 10>("fix NITER=10", w, iters);
    run<2, 12>("fix NITER=12", w, iters);
  }
  return 0;
}
