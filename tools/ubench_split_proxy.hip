// Timing-only proxy (results are NOT meaningful): compute waves write their results to LDS, one storer wave per
// workgroup moves them to the workspace, no synchronisation between the two.  Answers one question: how fast does
// the streaming step run when a compute wave never issues a wide store?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../kangaroo_amd/csrc/walk_kernel_stream.h"  // build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_split_proxy tools/ubench_split_proxy.hip
using namespace kg;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int NC, bool SPLIT>
__global__ void __launch_bounds__((NC + (SPLIT ? 1 : 0)) * 64, SPLIT ? 1 : 2) proxy_kernel(StreamArgs sa) {
  const WalkArgs &a = sa.a;
  extern __shared__ u32 lds[];
  u32 *jt = lds;
  u32 *obuf = lds + NB_JUMP * JT_STRIDE;  // [NC][15][3][64]
  const u32 tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  for (u32 i = tid; i < NB_JUMP * 18; i += blockDim.x) jt[(i / 18) * JT_STRIDE + i % 18] = a.jt->pxy[i / 18][i % 18];
  __syncthreads();
  const u32 K = sa.K, threads = a.threads;
  ws_ref wr;
  wr.rs = __builtin_amdgcn_make_buffer_rsrc(sa.ws, 0, (int)(WS_GROUPS * K * threads * 12u), 0x00020000);
  wr.row_bytes = threads * 12u;

  if (SPLIT && wave == NC) {  // ---- storer ----
    for (u32 run = 0; run < a.nb_run; ++run) {
      const bool asc = (run & 1) != 0;
      for (u32 s = 0; s < K; s += 2) {
        const u32 idA = asc ? s : K - 1 - s, idB = asc ? s + 1 : K - 2 - s;
#pragma unroll 1
        for (u32 w = 0; w < (u32)NC; ++w) {
          wr.off = (blockIdx.x * NC * 64 + w * 64 + lane) * 12u;
          u32 v[45];
#pragma unroll
          for (int i = 0; i < 45; ++i) v[i] = obuf[((w * 15 + i / 3) * 3 + i % 3) * 64 + lane];
          ws_store<0, 5>(wr, idA, v);
          ws_store<0, 8>(wr, idB, v + 18);
        }
      }
    }
    return;
  }

  const u32 Tl = SPLIT ? blockIdx.x * NC * 64 + tid : blockIdx.x * blockDim.x + tid;
  wr.off = Tl * 12u;
  auto load_rec = [&](StreamRec &r, u32 id) {
    u32 wv[18];
    ws_load<0, 18>(wv, wr, id);
#pragma unroll
    for (int i = 0; i < 9; ++i) { r.x.l[i] = wv[i]; r.y.l[i] = wv[9 + i]; }
  };
  auto load_p = [&](fe &q, u32 id) { ws_load<18, 9>(q.l, wr, id); };
  fe acc;
#pragma unroll
  for (int i = 0; i < 9; ++i) acc.l[i] = (i == 0) + lane;
  u32 dpacc = 0;
  auto jump = [&](const StreamRec &rec, const fe &dxi, const fe &jx, const u32 *row, u32 id, int second) {
    fe jy;
#pragma unroll
    for (int i = 0; i < 9; ++i) jy.l[i] = row[9 + i];
    fe dy, s, p2, rx, tt, ry;
    fe_sub_wide(dy, rec.y, jy);
    fe_mul(s, dy, dxi);
    fe_sqr(p2, s);
    fe_sub2_wide(rx, p2, jx, rec.x);
    fe_canon(rx);
    fe_sub_wide(tt, rec.x, rx);
    fe_mul(ry, tt, s);
    fe_sub_wide(ry, ry, rec.y);
    fe_weak(ry);
    {
      const u32 *row2 = jt + (rx.l[0] & (NB_JUMP - 1)) * JT_STRIDE;
      fe jx2, dx2;
#pragma unroll
      for (int i = 0; i < 9; ++i) jx2.l[i] = row2[i];
      fe_sub_wide(dx2, rx, jx2);
      fe_mul(acc, dx2, acc);
    }
    const u32 top_hi = (rx.l[8] << 8) | (rx.l[7] >> 21);
    dpacc += (top_hi & a.dp_mask_hi) == 0;
    if (SPLIT) {
      u32 *o = obuf + (wave * 15 + (second ? 6 : 0)) * 3 * 64 + lane;
#pragma unroll
      for (int i = 0; i < 9; ++i) { o[i * 64] = rx.l[i]; o[(9 + i) * 64] = ry.l[i]; }
      if (second) {
#pragma unroll
        for (int i = 0; i < 9; ++i) o[(18 + i) * 64] = acc.l[i];
      }
    } else {
      u32 wv[27];
#pragma unroll
      for (int i = 0; i < 9; ++i) { wv[i] = rx.l[i]; wv[9 + i] = ry.l[i]; wv[18 + i] = acc.l[i]; }
      if (second) ws_store<0, 8>(wr, id, wv); else ws_store<0, 5>(wr, id, wv);
    }
  };
  u32 run = 0;
#pragma unroll 1
  do {
    const bool asc = (run & 1) != 0;
    auto id_at = [&](u32 s) { return asc ? s : K - 1 - s; };
    StreamRec A, B;
    fe QB;
    load_rec(A, id_at(0));
    load_rec(B, id_at(1));
    load_p(QB, id_at(2));
    fe u;
    fe_inv(u, acc);
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int i = 0; i < 9; ++i) acc.l[i] = (i == 0);
#pragma unroll 1
    for (u32 s = 0; s < K; s += 2) {
      const u32 idA = id_at(s), idB = id_at(s + 1);
      const bool last = s + 2 >= K;
      StreamRec nA, nB;
      fe nQB;
      const u32 s2 = last ? s : s + 2;
      load_rec(nA, id_at(s2));
      __builtin_amdgcn_sched_barrier(0);
      const u32 jmpA = A.x.l[0] & (NB_JUMP - 1), jmpB = B.x.l[0] & (NB_JUMP - 1);
      const u32 *rowA = jt + jmpA * JT_STRIDE, *rowB = jt + jmpB * JT_STRIDE;
      fe jxA, jxB, dxA, dxB, dxiA, dxiB;
#pragma unroll
      for (int i = 0; i < 9; ++i) { jxA.l[i] = rowA[i]; jxB.l[i] = rowB[i]; }
      fe_sub_wide(dxA, A.x, jxA);
      fe_sub_wide(dxB, B.x, jxB);
      fe qA;
      fe_mul(qA, dxB, QB);
      fe_mul(dxiA, qA, u);
      fe_mul(u, dxA, u);
      fe_mul(dxiB, QB, u);
      fe_mul(u, dxB, u);
      jump(A, dxiA, jxA, rowA, idA, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_rec(nB, id_at(s2 + 1));
      load_p(nQB, s2 + 2 < K ? id_at(s2 + 2) : id_at(s2 + 1));
      __builtin_amdgcn_sched_barrier(0);
      jump(B, dxiB, jxB, rowB, idB, 1);
      A = nA; B = nB; QB = nQB;
    }
  } while (++run < a.nb_run);
  if (dpacc == 0xFFFFFFFFu) a.out_count[0] = 1;
}

__global__ void fill_kernel(u32 *p, size_t n, u32 seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    u32 x = (u32)i * 2654435761u + seed;
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
    p[i] = x & 0x1FFFFFFFu;
  }
}

template <int NC, bool SPLIT>
static void run_case(const char *name, u32 grid, u32 K, u32 nb_run, const JumpTableDev *jt, u32 *cnt) {
  const u32 block = (NC + (SPLIT ? 1 : 0)) * 64;
  const u32 threads = grid * NC * 64;
  const size_t ws_words = (size_t)WS_GROUPS * 3 * K * threads;
  u32 *ws;
  CK(hipMalloc((void **)&ws, ws_words * 4));
  fill_kernel<<<4096, 256>>>(ws, ws_words, 12345);
  StreamArgs sa{};
  sa.a.threads = threads; sa.a.nb_run = nb_run; sa.a.dp_mask_hi = 0xFFFFFFFFu; sa.a.dp_mask_lo = 0; sa.a.jt = jt;
  sa.a.out_count = cnt; sa.ws = ws; sa.hist = nullptr; sa.K = K;
  const size_t lds = (NB_JUMP * JT_STRIDE + (SPLIT ? NC * 15 * 3 * 64 : 0)) * 4;
  CK(hipFuncSetAttribute((const void *)proxy_kernel<NC, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((proxy_kernel<NC, SPLIT>), dim3(grid), dim3(block), lds, 0, sa);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double jumps = (double)threads * K * nb_run;
  printf("%-34s grid %u x %u threads, K %u, lds %zu B: %.2f ms -> %.2f GJ/s\n", name, grid, block, K, lds, best, jumps / best / 1e6);
  fflush(stdout);
  CK(hipFree(ws));
}

int main() {
  JumpTableDev h;
  for (int i = 0; i < NB_JUMP; ++i) {
    for (int j = 0; j < 18; ++j) h.pxy[i][j] = (u32)(rand() * 2654435761u) & 0x1FFFFFFFu;
    for (int j = 0; j < 8; ++j) h.dist[i][j] = rand();
  }
  JumpTableDev *jt; u32 *cnt;
  CK(hipMalloc((void **)&jt, sizeof h)); CK(hipMemcpy(jt, &h, sizeof h, hipMemcpyHostToDevice));
  CK(hipMalloc((void **)&cnt, 4)); CK(hipMemset(cnt, 0, 4));
  const u32 nb_run = 64;
  run_case<4, false>("baseline 4 waves/WG, own stores", 512, 64, nb_run, jt, cnt);
  run_case<7, true>("7 compute waves + storer wave", 256, 74, nb_run, jt, cnt);
  run_case<6, true>("6 compute waves + storer wave", 256, 86, nb_run, jt, cnt);
  run_case<4, false>("baseline again", 512, 64, nb_run, jt, cnt);
  return 0;
}
