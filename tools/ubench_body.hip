// Micro-benchmark: cost of the arithmetic of one kangaroo jump (5 mul + 1 sqr + subtractions + canonicalisation,
// no inversion, no memory traffic) as a function of occupancy.  Upper bound for any kernel organisation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/ubench_body tools/ubench_body.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../kangaroo_amd/csrc/fe29.h"
using namespace kg;

template <int W>
__global__ void __launch_bounds__(256, W) kbody(u32 *out, u32 seed, int iters, u64 *clk) {
  extern __shared__ u32 lds[];
  fe x, y, jx, jy, u, q, acc;
  u32 s = seed + 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
  fe *all[7] = {&x, &y, &jx, &jy, &u, &q, &acc};
  for (int k = 0; k < 7; ++k)
    for (int i = 0; i < 9; ++i) {
      s = s * 1664525u + 1013904223u;
      all[k]->l[i] = (s >> 3) & (i == 8 ? 0xFFFFFFu : M29);
    }
  if (threadIdx.x == 1000) lds[0] = s;
  const u64 t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    fe dx, dxi, dy, sl, p2, rx, tt, ry, dx2;
    fe_sub_wide(dx, x, jx);
    fe_mul(dxi, q, u);
    fe_mul(u, dx, u);
    fe_sub_wide(dy, y, jy);
    fe_mul(sl, dy, dxi);
    fe_sqr(p2, sl);
    fe_sub2_wide(rx, p2, jx, x);
    fe_canon(rx);
    fe_sub_wide(tt, x, rx);
    fe_mul(ry, tt, sl);
    fe_sub_wide(ry, ry, y);
    fe_weak(ry);
    x = rx;
    y = ry;
    fe_sub_wide(dx2, rx, jy);
    fe_mul(acc, dx2, acc);
    q = acc;
    jx.l[0] = (jx.l[0] + rx.l[1]) & M29;
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  u32 r = 0;
  for (int i = 0; i < 9; ++i) r ^= x.l[i] ^ y.l[i] ^ u.l[i] ^ acc.l[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int W>
static void run(int iters) {
  const int blocks = 256 * W * 4;
  const size_t lds = (W == 1 ? 100 : W == 2 ? 60 : W == 4 ? 36 : 18) * 1024;
  u32 *out;
  u64 *clk;
  (void)hipMalloc(&out, blocks * 256 * 4);
  (void)hipMalloc(&clk, 8);
  (void)hipFuncSetAttribute((const void *)kbody<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  kbody<W><<<blocks, 256, lds>>>(out, 1u, 8, clk);
  (void)hipEventRecord(e0);
  kbody<W><<<blocks, 256, lds>>>(out, 7u, iters, clk);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  u64 c = 0;
  (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const double jumps = (double)blocks * 256 * iters;
  printf("waves/SIMD=%d  %8.3f ms  ticks per body (wave 0) = %7.0f   -> %.2f G bodies/s (whole GPU)\n", W, ms,
         (double)c / iters, jumps / ms / 1e6);
}

int main() {
  run<1>(2000);
  run<2>(2000);
  run<4>(2000);
  run<8>(2000);
  return 0;
}
