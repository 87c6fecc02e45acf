// walk_kernel_stream.h — streaming form of the jump kernel: the herd lives in HBM during the launch and
// only the kangaroo being jumped (plus the prefetched next one) is in registers.
//
// Why: with the herd in registers (walk_kernel_rot.h) a lane can hold 16 kangaroos, the kernel runs one
// wave per SIMD, and the per-lane modular inverse is amortised over 16 jumps only (30 % of the cycles).
// Measured on MI355X (tools/ubench_body.hip) the arithmetic of one jump issues 28 % faster with two or
// more waves per SIMD than with one.  Here a lane owns K kangaroos with K set at run time (64 for the
// bench herd), the inverse is amortised over K, registers hold one body's working set, so 2-3 waves per
// SIMD fit, and the code is one rolled body (~12 KB, instruction-cache resident).  The price is HBM
// traffic: 72 + 18 B read, 72 + 18 + 1 B written per jump (kangaroos go in pairs and share one stored partial
// product), all of it lane-contiguous rows.
//
// Workspace (engine-private, HBM): a kangaroo slot is 27 words per lane
//   words 0..8 x limbs, 9..17 y limbs, 18..26 the 9 limbs of the stored partial product P[k],
// cut into 9 groups of three words; group g of slot k is a row of `threads` 12-byte entries at row index
// k * 9 + g, so that one buffer_load/store_dwordx3 moves 768 contiguous bytes per wavefront.
// Distances are not carried through the steps at all: every jump appends its jump index (one byte) to a
// history hist[step][slot][lane]; kg_hist_sum_kernel adds the jump distances of every kangaroo's history to D
// after the walk, and the few lanes that report a distinguished point sum their own history up to that step.
// 1 byte per jump instead of reading and writing the delta.  The prologue converts the X/Y planes into the
// workspace, the epilogue converts back, so the engine's resident format (walk_kernel.h) is unchanged.
//
// The step itself is the fused pass of walk_kernel_rot.h: same arithmetic, same results.  The four waves of a
// workgroup share one modular inversion per step through LDS.
#pragma once
#include "walk_kernel_rot.h"

namespace kg {

struct StreamArgs {
  WalkArgs a;
  u32 *ws;            // workspace, 9 * K rows of a.threads 12-byte entries
  unsigned char *hist;  // jump indices, [nb_run][K][a.threads]
  u32 K;              // kangaroos per lane
  u32 *pace;          // [PACE_KEYS][4]: per physical CU {claim counter, pass of workgroup 0, pass of workgroup 1, -}, zeroed per launch
};
constexpr u32 PACE_KEYS = 4096;

struct StreamRec {
  fe x, y;
};

// Workspace access through a buffer descriptor: an offset outside the workspace is dropped by the hardware
// bounds check instead of faulting, and there is no per-access address arithmetic on the VALU.
// cache policy of the workspace accesses (aux operand of the buffer instructions: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef KG_WS_AUX_LD
#define KG_WS_AUX_LD 0
#endif
#ifndef KG_WS_AUX_ST
#define KG_WS_AUX_ST 0
#endif
typedef __amdgpu_buffer_rsrc_t ws_rsrc;
typedef u32 u32x3 __attribute__((ext_vector_type(3)));
constexpr u32 WS_GROUPS = 9;
// One descriptor for the whole workspace (the host keeps it below 4 GiB); a group row is addressed by a
// uniform SGPR offset (slot * 9 + group) * row_bytes, the lane by a constant VGPR byte offset.
struct ws_ref {
  ws_rsrc rs;
  u32 row_bytes;  // threads * 12
  u32 off;        // lane * 12
};
// words [FIRST, FIRST + N) of slot id -> dst[0..N)
template <int FIRST, int N>
__device__ __forceinline__ void ws_load(u32 *dst, const ws_ref &w, u32 id) {
  constexpr int G0 = FIRST / 3, G1 = (FIRST + N - 1) / 3;
  const u32 base = id * WS_GROUPS;
#pragma unroll
  for (int g = G0; g <= G1; ++g) {
    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(w.rs, w.off, (base + (u32)g) * w.row_bytes, KG_WS_AUX_LD);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int i = 3 * g + c - FIRST;
      if (i >= 0 && i < N) dst[i] = v[c];
    }
  }
}
// Store-data hazard, found on MI355X with the full-size herd (tests/test_gpu_parity.py, 512 workgroups): a
// buffer_store_dwordx3 whose data VGPRs are overwritten by the VALU instruction right behind it can write the
// NEW values for lanes 12-15 of every 16 (the last quad of a row).  The compiler pads this hazard only for wide
// stores WITHOUT an SGPR offset (its model says a register soffset is safe; here it is not) and happily
// reuses the registers of the last limbs in the very next instruction.  ws_hold() is an s_nop that names the
// stored registers as inputs: nothing may overwrite them before it, and it is four wait states long (the compiler uses two where it knows the hazard).
__device__ __forceinline__ void ws_hold(const u32x3 &a, const u32x3 &b, const u32x3 &c) {
  asm volatile("s_nop 3" ::"v"(a), "v"(b), "v"(c));
}
// groups [G0, G1] of slot id <- src, where src[i] is word 3 * G0 + i of the slot
template <int G0, int G1>
__device__ __forceinline__ void ws_store(const ws_ref &w, u32 id, const u32 *src) {
  static_assert((G1 - G0 + 1) % 3 == 0, "ws_hold takes the groups three at a time");
  const u32 base = id * WS_GROUPS;
  u32x3 v[G1 - G0 + 1];
#pragma unroll
  for (int g = G0; g <= G1; ++g) {
#pragma unroll
    for (int c = 0; c < 3; ++c) v[g - G0][c] = src[3 * (g - G0) + c];
#ifdef KG_EXP_STORE_SLOT0  // timing experiment: all stores land in slot 0 (stay in L2), results wrong
    __builtin_amdgcn_raw_buffer_store_b96(v[g - G0], w.rs, w.off, (u32)g * w.row_bytes, KG_WS_AUX_ST);
#else
    __builtin_amdgcn_raw_buffer_store_b96(v[g - G0], w.rs, w.off, (base + (u32)g) * w.row_bytes, KG_WS_AUX_ST);
#endif
  }
#pragma unroll
  for (int g = G1 - G0 - 2; g >= 0; g -= 3) ws_hold(v[g], v[g + 1], v[g + 2]);
}

template <int DW, int W>
__global__ void __launch_bounds__(WALK_BLOCK, W) kg_walk_kernel_stream(StreamArgs sa) {
  const WalkArgs &a = sa.a;
  extern __shared__ u32 lds[];
  u32 *jt = lds;                           // [64][JT_STRIDE]
  u32 *jdist = lds + NB_JUMP * JT_STRIDE;  // [64][DW]
  u32 *inv_ex = jdist + NB_JUMP * DW;      // [8][9][64]: products in, inverses out (shared inversion)

  const u32 tid = threadIdx.x;
  for (u32 i = tid; i < NB_JUMP * 18; i += WALK_BLOCK) {
    u32 row = i / 18, col = i % 18;
    jt[row * JT_STRIDE + col] = a.jt->pxy[row][col];
  }
  for (u32 i = tid; i < NB_JUMP * DW; i += WALK_BLOCK) jdist[i] = a.jt->dist[i / DW][i % DW];
  // physical CU of this workgroup: HW_ID (cu_id 11:8, sh_id 12, se_id 15:13) and XCC_ID; first come, first slot
  const u32 pace_key = ((__builtin_amdgcn_s_getreg(0xF804) >> 8) & 0xFFu) | ((__builtin_amdgcn_s_getreg(0x1814) & 0xFu) << 8);
  if (tid == 0) inv_ex[8 * 9 * 64] = atomicAdd(sa.pace + pace_key * 4, 1u) & 1u;
  __syncthreads();
  const u32 pace_slot = inv_ex[8 * 9 * 64];

  const u32 Tl = blockIdx.x * WALK_BLOCK + tid;  // lane index < threads
  const u32 off = Tl * 12u;                      // byte offset inside a workspace row
  const u64 T = Tl;
  const u32 K = sa.K;
  const u32 threads = a.threads;
  u32 *ws = sa.ws;
  ws_ref wr;
  wr.rs = __builtin_amdgcn_make_buffer_rsrc(ws, 0, (int)(WS_GROUPS * K * threads * 12u), 0x00020000);
  wr.row_bytes = threads * 12u;
  wr.off = off;

  auto load_rec = [&](StreamRec &r, u32 id) {
    u32 wv[18];
    ws_load<0, 18>(wv, wr, id);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      r.x.l[i] = wv[i];
      r.y.l[i] = wv[9 + i];
    }
  };
  auto load_p = [&](fe &q, u32 id) { ws_load<18, 9>(q.l, wr, id); };
  auto store_all = [&](u32 id, const fe &x, const fe &y, const fe &pp) {
    u32 wv[27];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      wv[i] = x.l[i];
      wv[9 + i] = y.l[i];
      wv[18 + i] = pp.l[i];
    }
    ws_store<0, 8>(wr, id, wv);
  };
  auto store_xy = [&](u32 id, const fe &x, const fe &y) {
    u32 wv[18];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      wv[i] = x.l[i];
      wv[9 + i] = y.l[i];
    }
    ws_store<0, 5>(wr, id, wv);
  };
  // history row of (step, slot): K * threads bytes per step
  auto hist_row = [&](u32 step, u32 id) { return sa.hist + ((u64)step * K + id) * threads; };
  // distance travelled by slot id in steps [0, steps): sum of the jump distances in its history (used by the
  // lanes that report a distinguished point; the end-of-launch sums are kg_hist_sum_kernel's job)
  auto travelled = [&](dlt<DW> &sum, u32 steps, u32 id) {
#pragma unroll
    for (int i = 0; i < DW; ++i) sum.w[i] = 0u;
    constexpr u32 CH = 4;  // history bytes requested together (rare path: kept small, it costs SGPRs)
#pragma unroll 1
    for (u32 r0 = 0; r0 < steps; r0 += CH) {
      u32 j[CH];
#pragma unroll
      for (u32 i = 0; i < CH; ++i) {
        const u32 r = r0 + i < steps ? r0 + i : r0;  // clamped: a valid row, masked out below
        j[i] = hist_row(r, id)[Tl];
      }
#pragma unroll
      for (u32 i = 0; i < CH; ++i)
        if (r0 + i < steps) delta_add<DW>(sum.w, jdist + j[i] * DW);
    }
  };

  // ---- prologue: X/Y planes -> workspace, prefix products of the first step's dx (ascending) ----
  fe acc;
#pragma unroll
  for (int i = 0; i < 9; ++i) acc.l[i] = (i == 0);
#pragma unroll 1
  for (u32 k = 0; k < K; ++k) {
    const u64 p = T + (u64)k * threads;
    u32 w[8];
    fe x, y;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u64 v = a.X[(u64)i * a.cap + p];
      w[2 * i] = (u32)v;
      w[2 * i + 1] = (u32)(v >> 32);
    }
    fe_from_words(x, w);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u64 v = a.Y[(u64)i * a.cap + p];
      w[2 * i] = (u32)v;
      w[2 * i + 1] = (u32)(v >> 32);
    }
    fe_from_words(y, w);
    const u32 *row = jt + (x.l[0] & (NB_JUMP - 1)) * JT_STRIDE;
    fe jx, dx;
#pragma unroll
    for (int i = 0; i < 9; ++i) jx.l[i] = row[i];
    fe_sub_wide(dx, x, jx);
    fe_mul(acc, dx, acc);
    if (k & 1) store_all(k, x, y, acc);
    else store_xy(k, x, y);
  }

  // one jump of the kangaroo in `rec` (slot id) given dx^-1; multiplies its next dx into acc, stores the new
  // point (and, for the second kangaroo of a pair, the running product), reports a distinguished point
  auto jump = [&](const StreamRec &rec, const fe &dxi, const fe &jx, const u32 *row, u32 jmp, u32 id, u32 run, bool with_p) {
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
    hist_row(run, id)[Tl] = (unsigned char)jmp;
    {
      const u32 *row2 = jt + (rx.l[0] & (NB_JUMP - 1)) * JT_STRIDE;
      fe jx2, dx2;
#pragma unroll
      for (int i = 0; i < 9; ++i) jx2.l[i] = row2[i];
      fe_sub_wide(dx2, rx, jx2);
      fe_mul(acc, dx2, acc);
    }
    if (with_p) store_all(id, rx, ry, acc);
    else store_xy(id, rx, ry);

    // distinguished point: (x.bits64[3] & dpMask) == 0   (math/gpu/compute.h:54)
    const u32 top_hi = (rx.l[8] << 8) | (rx.l[7] >> 21);
    const u32 top_lo = (rx.l[7] << 11) | (rx.l[6] >> 18);
    const u64 p = T + (u64)id * threads;
    const bool is_dp = ((top_hi & a.dp_mask_hi) | (top_lo & a.dp_mask_lo)) == 0 && p < a.n;
    const unsigned long long m = __ballot(is_dp);
    if (m != 0) {
      const u32 lane = __lane_id();
      const u32 leader = (u32)__ffsll(m) - 1;
      u32 basep = 0;
      if (lane == leader) basep = atomicAdd(a.out_count, (u32)__popcll(m));
      basep = __shfl(basep, (int)leader);
      if (is_dp) {
        const u32 pos = basep + (u32)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < a.max_found) {
          DpItemDev *it = a.out + pos;
          u32 w[8];
          fe_to_words(w, rx);
#pragma unroll
          for (int i = 0; i < 8; ++i) it->x[i] = w[i];
          dlt<DW> dl;
          travelled(dl, run, id);  // steps before this one, from the history (this branch is rare)
          delta_add<DW>(dl.w, jdist + jmp * DW);
          add_dist_256<DW>(w, a.D, a.cap, p, dl);
#pragma unroll
          for (int i = 0; i < 8; ++i) it->d[i] = w[i];
          it->kidx = p;
        }
      }
    }
  };

  // ---- steps: pass `run` visits the ids descending when run is even (the prologue went up), ascending when odd.
  // Position s of a pass holds id(s) = asc ? s : K-1-s.  Kangaroos go in pairs (A at s, B at s+1, K is even) and
  // only B's running product is written: HBM writes, not reads, are what limits this kernel.  The product of
  // the dx still to come after B, q_B, was stored by the previous pass in slot id_B + dir (the second of its pair
  // then); q_A = q_B * dx_B costs one extra multiplication per pair.
  u32 run = 0;
#pragma unroll 1
  do {
    const bool asc = (run & 1) != 0;
    auto id_at = [&](u32 s) { return asc ? s : K - 1 - s; };
    // Pacing.  The two workgroups resident on a CU do not share it evenly: the issue arbiter prefers the older
    // wave, so the workgroup dispatched first used to finish at 70 % of the kernel time and its neighbour then ran
    // alone at one wave per SIMD.  Every pass each workgroup publishes its pass number in a table indexed by the
    // physical CU and raises its wave priority while it is behind its neighbour.
    {
      if (tid == 0) __hip_atomic_store(sa.pace + pace_key * 4 + 1 + pace_slot, run + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u32 other = __builtin_amdgcn_readfirstlane(
          __hip_atomic_load(sa.pace + pace_key * 4 + 1 + (pace_slot ^ 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#ifndef KG_NO_PACE  // (A/B switch)
      if (other > run + 1) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
#endif
    }
    // operands of the first pair: requested before the inversion, which hides the latency
    StreamRec A, B;
    fe QB;
    load_rec(A, id_at(0));
    load_rec(B, id_at(1));
    load_p(QB, K > 2 ? id_at(2) : id_at(1));
    fe u;
#ifdef KG_STREAM_PRIVATE_INV
    fe_inv(u, acc);
#else
    // One inversion for the four waves of the workgroup: lane l of every wave posts its product, the wave
    // whose turn it is inverts the product of the four and hands each wave the inverse of its own (3 + 6
    // multiplications against three inversions saved).  The other three waves wait at the barrier; their
    // SIMDs are not idle meanwhile, each holds a wave of another workgroup.
    {
      const u32 wave = tid >> 6, lane = tid & 63u;
#pragma unroll
      for (int i = 0; i < 9; ++i) inv_ex[(wave * 9 + i) * 64 + lane] = acc.l[i];
      __syncthreads();
      if (wave == (run & 3u)) {
        fe a0, a1, a2, a3, p01, p23, t, iv, i01, i23, r;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          a0.l[i] = inv_ex[(0 * 9 + i) * 64 + lane];
          a1.l[i] = inv_ex[(1 * 9 + i) * 64 + lane];
          a2.l[i] = inv_ex[(2 * 9 + i) * 64 + lane];
          a3.l[i] = inv_ex[(3 * 9 + i) * 64 + lane];
        }
        fe_mul(p01, a0, a1);
        fe_mul(p23, a2, a3);
        fe_mul(t, p01, p23);
        fe_inv(iv, t);
        fe_mul(i01, iv, p23);
        fe_mul(i23, iv, p01);
        fe_mul(r, i01, a1);
#pragma unroll
        for (int i = 0; i < 9; ++i) inv_ex[((4 + 0) * 9 + i) * 64 + lane] = r.l[i];
        fe_mul(r, i01, a0);
#pragma unroll
        for (int i = 0; i < 9; ++i) inv_ex[((4 + 1) * 9 + i) * 64 + lane] = r.l[i];
        fe_mul(r, i23, a3);
#pragma unroll
        for (int i = 0; i < 9; ++i) inv_ex[((4 + 2) * 9 + i) * 64 + lane] = r.l[i];
        fe_mul(r, i23, a2);
#pragma unroll
        for (int i = 0; i < 9; ++i) inv_ex[((4 + 3) * 9 + i) * 64 + lane] = r.l[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 9; ++i) u.l[i] = inv_ex[((4 + wave) * 9 + i) * 64 + lane];
    }
#endif
    // The first operands have long arrived.  Saying so matters: the compiler's wait-count bookkeeping merges
    // the loop entry with the back edge, and a load still "pending" at the entry turns into a wait on the
    // previous body's STORES at the top of every iteration (vmcnt counts loads and stores in order).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
    for (int i = 0; i < 9; ++i) acc.l[i] = (i == 0);

#pragma unroll 1
    for (u32 s = 0; s < K; s += 2) {
      const u32 idA = id_at(s), idB = id_at(s + 1);
      const bool last = s + 2 >= K;
      // prefetch the next pair (clamped at the end of the pass: valid slots, values unused)
      StreamRec nA, nB;
      fe nQB;
      // A's successor is requested here, B's after A has jumped: two short bursts instead of one long one
      // (the per-CU memory queue is what waves block on), and 25 fewer live registers
      const u32 s2 = last ? s : s + 2;
      load_rec(nA, id_at(s2));
      __builtin_amdgcn_sched_barrier(0);

      const u32 jmpA = A.x.l[0] & (NB_JUMP - 1), jmpB = B.x.l[0] & (NB_JUMP - 1);
      const u32 *rowA = jt + jmpA * JT_STRIDE, *rowB = jt + jmpB * JT_STRIDE;
      fe jxA, jxB, dxA, dxB, dxiA, dxiB;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        jxA.l[i] = rowA[i];
        jxB.l[i] = rowB[i];
      }
      fe_sub_wide(dxA, A.x, jxA);
      fe_sub_wide(dxB, B.x, jxB);
      if (!last) {
        fe qA;
        fe_mul(qA, dxB, QB);
        fe_mul(dxiA, qA, u);
        fe_mul(u, dxA, u);
        fe_mul(dxiB, QB, u);
        fe_mul(u, dxB, u);
      } else {  // nothing comes after B: q_B = 1
        fe_mul(dxiA, dxB, u);
        fe_mul(u, dxA, u);
        dxiB = u;
      }
      jump(A, dxiA, jxA, rowA, jmpA, idA, run, false);
      __builtin_amdgcn_sched_barrier(0);
      load_rec(nB, id_at(s2 + 1));
      load_p(nQB, s2 + 2 < K ? id_at(s2 + 2) : id_at(s2 + 1));
      __builtin_amdgcn_sched_barrier(0);
      jump(B, dxiB, jxB, rowB, jmpB, idB, run, true);
      A = nA;
      B = nB;
      QB = nQB;
    }
  } while (++run < a.nb_run);

  // ---- epilogue: workspace -> X/Y planes (canonical); the distances are summed by kg_hist_sum_kernel ----
#pragma unroll 1
  for (u32 k = 0; k < K; ++k) {
    StreamRec r;
    load_rec(r, k);
    const u64 p = T + (u64)k * threads;
    u32 w[8];
    fe_canon(r.x);
    fe_to_words(w, r.x);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.X[(u64)i * a.cap + p] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
    fe_canon(r.y);
    fe_to_words(w, r.y);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.Y[(u64)i * a.cap + p] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
  }
}

// End of launch: D[p] += sum of the jump distances in kangaroo p's history.  One thread per kangaroo, full
// occupancy, the history rows are lane-contiguous bytes (1 B per jump read once).
template <int DW>
__global__ void __launch_bounds__(WALK_BLOCK) kg_hist_sum_kernel(const unsigned char *hist, const JumpTableDev *jtab, u64 *D,
                                                                 u64 cap, u64 n, u32 threads, u32 K, u32 nb_run) {
  __shared__ u32 jdist[NB_JUMP * DW];
  for (u32 i = threadIdx.x; i < NB_JUMP * DW; i += WALK_BLOCK) jdist[i] = jtab->dist[i / DW][i % DW];
  __syncthreads();
  const u64 p = (u64)blockIdx.x * WALK_BLOCK + threadIdx.x;
  if (p >= cap) return;
  const u32 k = (u32)(p / threads), lane = (u32)(p % threads);
  const unsigned char *col = hist + (u64)k * threads + lane;  // + step * K * threads
  const u64 step_stride = (u64)K * threads;
  dlt<DW> sum;
#pragma unroll
  for (int i = 0; i < DW; ++i) sum.w[i] = 0u;
  u32 r = 0;
  for (; r + 8 <= nb_run; r += 8) {
    u32 j[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) j[i] = col[(u64)(r + i) * step_stride];
#pragma unroll
    for (int i = 0; i < 8; ++i) delta_add<DW>(sum.w, jdist + j[i] * DW);
  }
  for (; r < nb_run; ++r) delta_add<DW>(sum.w, jdist + col[(u64)r * step_stride] * DW);
  u32 w[8];
  add_dist_256<DW>(w, D, cap, p, sum);
#pragma unroll
  for (int i = 0; i < 4; ++i) D[(u64)i * cap + p] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
}

constexpr size_t walk_stream_lds_bytes(int DW) {
  return (size_t)(NB_JUMP * JT_STRIDE + NB_JUMP * DW + 8 * 9 * 64 + 4) * sizeof(u32);  // + the pacing slot
}

}  // namespace kg
