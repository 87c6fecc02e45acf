// walk_kernel_stream.h — streaming form of the jump kernel: the herd lives in HBM during the launch and
// only the pair of kangaroos being jumped (plus the prefetched next pair) is in registers.
//
// Why: with the herd in registers (walk_kernel_rot.h) a lane can hold 16 kangaroos, the kernel runs one
// wave per SIMD, and the per-lane modular inverse is amortised over 16 jumps only (30 % of the cycles).
// Measured on MI355X (tools/ubench_body.hip) the arithmetic of one jump issues 28 % faster with two or
// more waves per SIMD than with one.  Here a lane owns K kangaroos with K set at run time (64 for the
// bench herd), the inverse is amortised over K, registers hold one body's working set, so two waves per
// SIMD fit, and the code is one rolled body (instruction-cache resident).  The price is HBM traffic, all of it
// lane-contiguous 16-byte rows: 64 + 16 B read and 64 + 16 + 1 B written per jump (kangaroos go in pairs and
// share one stored partial product).
//
// Workspace (engine-private, HBM): a kangaroo slot is 24 words per lane in six rows of four words
//   rows 0,1 x   rows 2,3 y   rows 4,5 the stored partial product P[k]
// each a 256-bit value as eight fully carried 32-bit words (x canonical; y and P only < 2^256, see fe_settle);
// row r of slot k is a run of `threads` 16-byte entries at row index k * 6 + r, so that one
// buffer_load/store_dwordx4 moves 1024 contiguous, 16-byte aligned bytes per wavefront.  (Round 1 stored the nine
// 29-bit limbs as they are, in dwordx3 rows: 183 B per jump instead of 162, and stores that straddle 64-byte
// lines; both forms measure the same because the kernel runs at the board's power limit)
// Distances are not carried through the steps at all: every jump appends its jump index (one byte) to a
// history hist[step][pair][lane] (16 bits per pair of slots); kg_hist_sum_kernel adds the jump distances of every
// kangaroo's history to D after the walk, and the few lanes that report a distinguished point sum their own history
// up to that step.  The prologue copies the X/Y planes into the workspace, the epilogue copies back, so the
// engine's resident format (walk_common.h) is unchanged.
//
// The step itself is the fused pass of walk_kernel_rot.h: same arithmetic, same results.  The four waves of a
// workgroup share one modular inversion per step through LDS.
#pragma once
#include "walk_kernel_rot.h"

namespace kg {

// Distinguished points of the streaming kernel.  The walk does not carry distances (see below), so a reporting lane
// writes x and kIdx only and chains the item to its kangaroo (dp_head / dp_link, device memory); kg_hist_sum_kernel, which
// reads every kangaroo's jump history anyway, fills in the distances of the chained items afterwards.  Round 2 had the
// reporting lane re-sum its own history from HBM inside the walk kernel: a divergent loop of up to nb_run iterations
// that cost 27 % of the throughput at DP = 8 (one DP per kangaroo and launch).
struct DpLink {
  u32 prev;  // ring position of the same kangaroo's previous item of this launch, DP_NONE if this was its first
  u32 step;  // jump of the launch (0-based) that produced the distinguished point
};
constexpr u32 DP_NONE = 0xFFFFFFFFu;
struct StreamArgs {
  WalkArgs a;
  u32 *ws;            // workspace, WS_ROWS * K rows of a.threads 16-byte entries
  unsigned char *hist;  // jump indices, [nb_run][K][a.threads]
  u32 K;              // kangaroos per lane
  u32 *pace;          // [PACE_KEYS][4]: per physical CU {claim counter, pass of workgroup 0, pass of workgroup 1, -}, zeroed per launch;
                      // behind it PACE_TAIL words: the clock stamps of workgroup 0 (shader cycles, 100 MHz reference ticks)
  u32 *dp_head;       // [cap]: ring position of the kangaroo's most recent distinguished point of this launch, DP_NONE if none
  DpLink *dp_link;    // [max_found]: per reported item, the kangaroo's previous item and the step it was found at
};
constexpr u32 PACE_KEYS = 4096;
constexpr u32 PACE_TAIL = 4;

// Workspace access through a buffer descriptor: an offset outside the workspace is dropped by the hardware
// bounds check instead of faulting, and there is no per-access address arithmetic on the VALU.
// cache policy of the workspace accesses (aux operand of the buffer instructions: 1 = sc0, 2 = nt, 16 = sc1).
// Stores are non-temporal: a slot is read again only a whole pass (hundreds of MB) later; measured +0.6 % on one box,
// twice (nt loads: no gain, both: -0.5 %; gpurun_out/ab_r02c.txt summarised in profiles/r02_power_trace.txt).
#ifndef KG_WS_AUX_LD
#define KG_WS_AUX_LD 0
#endif
#ifndef KG_WS_AUX_ST
#define KG_WS_AUX_ST 2
#endif
// 1: kangaroos go in pairs and only the second of a pair stores the running product (one extra multiplication per
// pair, 32 B less traffic per jump).  0: every kangaroo stores it (5 multiplications per jump instead of 5.5).
#ifndef KG_WS_PAIRS
#define KG_WS_PAIRS 1
#endif
// 1 (needs KG_WS_PAIRS): kangaroos go in QUADS (A, B, C, D) and only D stores the running product: q_C = q_D * dx_D and
// q_B = q_C * dx_C are recomputed (11 multiplications per quad instead of 10), 8 B less read and 8 B less written per
// jump.  Stores are what this kernel pays most for (profiles/r03_levers.txt: skipping every other product store alone is
// worth 4 %).  The loop still runs pair by pair (one body, the instruction cache holds it); K is a multiple of 4.
#ifndef KG_WS_QUADS
#define KG_WS_QUADS 0
#endif
#ifndef KG_WS_WAVE_MAJOR
#define KG_WS_WAVE_MAJOR 0
#endif
#if KG_WS_QUADS && !KG_WS_PAIRS
#error "KG_WS_QUADS needs KG_WS_PAIRS"
#endif
constexpr u32 WS_GROUP = KG_WS_QUADS ? 4 : 2;  // kangaroos per lane come in multiples of this
// 1: the LDS copy of the jump table holds the subtrahends already lifted and negated — per row 2p' - jPx, 2p' - jPy and
// 4p' - jPx (p' = the limb-lifted multiples of p that fe_sub_wide / fe_sub2_wide add, fe29.h) — so that
// x - jPx, y - jPy and s^2 - jPx - x cost one limb-wise add (two for the last) instead of two (three): 36 VALU
// instructions fewer per jump, and jPx is not held in registers between the dx and the rx of a jump.
// 0: rows hold jPx, jPy as they are (round 1-2).
#ifndef KG_JT_NEG
#define KG_JT_NEG 1
#endif
// MEASUREMENT VARIANT, off in the library (profiles/r03_levers.txt): 1 = the three groups of a row start on 16-byte
// boundaries (row of 36 words) and are read with two ds_read_b128 and one ds_read_b32 each instead of ds_read2_b32 pairs.
#ifndef KG_JT_WIDE
#define KG_JT_WIDE 0
#endif
constexpr int SJT_GROUP = (KG_JT_NEG && KG_JT_WIDE) ? 12 : 9;  // words from one group of a row to the next
constexpr int SJT_STRIDE = KG_JT_NEG ? (KG_JT_WIDE ? 36 : 27) : JT_STRIDE;  // words of an LDS row (27: odd, spreads the banks)
// the nine limbs of one group of a row
__device__ __forceinline__ void jt_load9(fe &r, const u32 *g) {
#if KG_JT_NEG && KG_JT_WIDE
  typedef u32 v4 __attribute__((ext_vector_type(4)));
  const v4 *q = (const v4 *)__builtin_assume_aligned(g, 16);
  const v4 a = q[0], b = q[1];
  r.l[0] = a[0]; r.l[1] = a[1]; r.l[2] = a[2]; r.l[3] = a[3];
  r.l[4] = b[0]; r.l[5] = b[1]; r.l[6] = b[2]; r.l[7] = b[3];
  r.l[8] = g[8];
#else
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = g[i];
#endif
}
// MEASUREMENT VARIANT, off in the library (profiles/r03_levers.txt): 1 = ONE inversion per workgroup and step, computed
// on ONE lane.  The inverting wave first folds the 64 per-lane products of the workgroup into one with a product tree
// across its lanes (through LDS, KG_INV_TREE_LEVELS levels), inverts on the surviving lanes with every other lane
// switched off, and walks the tree back down (Montgomery's trick across lanes): 18 multiplications with 63 + 126
// active lanes in all, and an inversion that toggles one lane instead of 64.  Measured on MI355X: the shader clock
// rises 3 % (less energy per cycle, the kernel runs at the board's power cap) but the launch takes 1.7 % LONGER (6
// levels; 3 levels -0.9 %, 2 levels -0.4 %): the extra multiplications sit on the step's critical path, and cycles in
// which the SIMDs wait still pay for clocks and leakage.
// 0: the inverting wave inverts 64 values, one per lane (what ships).
#ifndef KG_INV_TREE
#define KG_INV_TREE 0
#endif
#ifndef KG_INV_TREE_LEVELS
#define KG_INV_TREE_LEVELS 6  // 6: down to one lane; 3: down to eight lanes (half the multiplications on the critical path)
#endif
constexpr u32 INV_TREE_WORDS = KG_INV_TREE ? KG_INV_TREE_LEVELS * 9 * 64 : 0;
// form of the single-lane inversion (modinv30.h): 2 = batches of branch-free iterations (what 64 lanes want),
// 1 = a per-lane loop with data-dependent branches (one lane has nobody to diverge from)
#ifndef KG_INV_TREE_MODE
#define KG_INV_TREE_MODE 2
#endif
typedef __amdgpu_buffer_rsrc_t ws_rsrc;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
constexpr u32 WS_ROWS = 6;                  // rows of a slot: 0,1 = x, 2,3 = y, 4,5 = stored partial product
constexpr u32 WS_SLOT_WORDS = WS_ROWS * 4;  // 32-bit words of one kangaroo slot per lane
// One descriptor for the whole workspace (the host keeps it below 4 GiB); a row is addressed by a uniform SGPR
// offset (slot * 6 + row) * row_bytes, the lane by a constant VGPR byte offset.
struct ws_ref {
  ws_rsrc rs;
  u32 row_bytes;  // threads * 16
  u32 off;        // lane * 16
};
// a 256-bit value as it lies in the workspace: eight 32-bit words, fully carried, < 2^256
struct pk256 {
  u32x4 lo, hi;
};
// rows [ROW, ROW + 2) of slot id
// KG_TIMING_NO_WS_LOADS / KG_TIMING_NO_WS_STORES: TIMING EXPERIMENTS ONLY (wrong results, never in the library): what the
// workspace traffic of the pass loop costs, loads and stores apart (profiles/r03_levers.txt)
template <int ROW>
__device__ __forceinline__ void ws_load(pk256 &v, const ws_ref &w, u32 id) {
#ifdef KG_TIMING_NO_WS_LOADS
  v.lo = u32x4{id, w.off, (u32)ROW, id * 3u};
  v.hi = u32x4{id ^ w.off, 7u, w.off, 1u};
  asm volatile("" : "+v"(v.lo), "+v"(v.hi));
  return;
#endif
  const u32 base = (id * WS_ROWS + (u32)ROW) * w.row_bytes;
  v.lo = __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.off, base, KG_WS_AUX_LD);
  v.hi = __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.off, base + w.row_bytes, KG_WS_AUX_LD);
}
template <int ROW>
__device__ __forceinline__ void ws_store(const ws_ref &w, u32 id, const pk256 &v) {
#ifdef KG_TIMING_NO_WS_STORES
  asm volatile("" ::"v"(v.lo), "v"(v.hi));
  return;
#endif
  const u32 base = (id * WS_ROWS + (u32)ROW) * w.row_bytes;
  __builtin_amdgcn_raw_buffer_store_b128(v.lo, w.rs, w.off, base, KG_WS_AUX_ST);
  __builtin_amdgcn_raw_buffer_store_b128(v.hi, w.rs, w.off, base + w.row_bytes, KG_WS_AUX_ST);
}
// Store-data hazard, found on MI355X with the full-size herd (tests/test_gpu_parity.py, 512 workgroups): a wide
// buffer store (dwordx3 / dwordx4) whose data VGPRs are overwritten by the VALU instruction right behind it can
// write the NEW values for the last lanes of a row.  The compiler pads this hazard only for wide stores WITHOUT an
// SGPR offset (its model says a register soffset is safe; here it is not) and happily reuses the data registers in
// the very next instruction.  ws_hold() is an s_nop that names the stored registers as inputs: nothing may
// overwrite them before it, and it is four wait states long (the compiler uses two where it knows the hazard).
// tools/check_store_hazard.py verifies the emitted code of every kernel of the library for this pattern.
__device__ __forceinline__ void ws_hold(const pk256 &a) { asm volatile("s_nop 3" ::"v"(a.lo), "v"(a.hi)); }
__device__ __forceinline__ void ws_hold(const pk256 &a, const pk256 &b) {
  asm volatile("s_nop 3" ::"v"(a.lo), "v"(a.hi), "v"(b.lo), "v"(b.hi));
}
__device__ __forceinline__ void ws_hold(const pk256 &a, const pk256 &b, const pk256 &c) {
  asm volatile("s_nop 3" ::"v"(a.lo), "v"(a.hi), "v"(b.lo), "v"(b.hi), "v"(c.lo), "v"(c.hi));
}

__device__ __forceinline__ void fe_unpack(fe &r, const pk256 &v) {
  const u32 w[8] = {v.lo[0], v.lo[1], v.lo[2], v.lo[3], v.hi[0], v.hi[1], v.hi[2], v.hi[3]};
  fe_from_words(r, w);
}
// a must be settled (fe_settle) or canonical
__device__ __forceinline__ void fe_pack(pk256 &v, const fe &a) {
  u32 w[8];
  fe_to_words(w, a);
  v.lo = u32x4{w[0], w[1], w[2], w[3]};
  v.hi = u32x4{w[4], w[5], w[6], w[7]};
}
__device__ __forceinline__ pk256 pk_from_planes(const u64 *P, u64 cap, u64 p) {
  pk256 v;
  const u64 a = P[p], b = P[cap + p], c = P[2 * cap + p], d = P[3 * cap + p];
  v.lo = u32x4{(u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32)};
  v.hi = u32x4{(u32)c, (u32)(c >> 32), (u32)d, (u32)(d >> 32)};
  return v;
}
__device__ __forceinline__ void pk_to_planes(u64 *P, u64 cap, u64 p, const pk256 &v) {
  P[p] = (u64)v.lo[0] | ((u64)v.lo[1] << 32);
  P[cap + p] = (u64)v.lo[2] | ((u64)v.lo[3] << 32);
  P[2 * cap + p] = (u64)v.hi[0] | ((u64)v.hi[1] << 32);
  P[3 * cap + p] = (u64)v.hi[2] | ((u64)v.hi[3] << 32);
}

struct StreamRec {  // one kangaroo as it travels through the registers between load and use
  pk256 x, y;
};

template <int W>
__global__ void __launch_bounds__(WALK_BLOCK, W) kg_walk_kernel_stream(StreamArgs sa) {
  const WalkArgs &a = sa.a;
  extern __shared__ u32 lds[];
  u32 *jt = lds;                            // [64][SJT_STRIDE]
  u32 *inv_ex = lds + NB_JUMP * SJT_STRIDE;  // [8][9][64]: products in, inverses out (shared inversion)

  const u32 tid = threadIdx.x;
  // clock stamps (workgroup 0 reports them at the end): the shader clock counter against the constant 100 MHz
  // reference gives the clock the kernel really ran at, which under this load is well below the 2.4 GHz peak
  const u64 stamp_clk = __builtin_amdgcn_s_memtime(), stamp_ref = __builtin_amdgcn_s_memrealtime();
  for (u32 i = tid; i < NB_JUMP * 18; i += WALK_BLOCK) {
    u32 row = i / 18, col = i % 18;
    const u32 v = a.jt->pxy[row][col];
#if KG_JT_NEG
    jt[row * SJT_STRIDE + (col / 9) * SJT_GROUP + col % 9] = fe_b2(col % 9) - v;  // 2p' - jPx | 2p' - jPy
    if (col < 9) jt[row * SJT_STRIDE + 2 * SJT_GROUP + col] = fe_b4(col) - v;       // 4p' - jPx
#else
    jt[row * SJT_STRIDE + col] = v;
#endif
  }
  // physical CU of this workgroup: HW_ID (cu_id 11:8, sh_id 12, se_id 15:13) and XCC_ID; first come, first slot
  const u32 pace_key = ((__builtin_amdgcn_s_getreg(0xF804) >> 8) & 0xFFu) | ((__builtin_amdgcn_s_getreg(0x1814) & 0xFu) << 8);
  if (tid == 0) inv_ex[8 * 9 * 64] = atomicAdd(sa.pace + pace_key * 4, 1u) & 1u;
  __syncthreads();
  const u32 pace_slot = inv_ex[8 * 9 * 64];

  const u32 Tl = blockIdx.x * WALK_BLOCK + tid;  // lane index < threads
  const u64 T = Tl;
  const u32 K = sa.K;
  const u32 threads = a.threads;
  ws_ref wr;
  wr.rs = __builtin_amdgcn_make_buffer_rsrc(sa.ws, 0, (int)(WS_SLOT_WORDS * 4u * K * threads), 0x00020000);
#if KG_WS_WAVE_MAJOR
  // MEASUREMENT VARIANT, off in the library (profiles/r03_levers.txt): a wave's slot record (six rows of 1 KB) is
  // contiguous and its K records follow each other, instead of every row being a run of all lanes' entries
  wr.row_bytes = 64u * 16u;
  wr.off = (Tl >> 6) * (K * WS_ROWS * 64u * 16u) + (Tl & 63u) * 16u;
#else
  wr.row_bytes = threads * 16u;
  wr.off = Tl * 16u;
#endif

  // jump-index history: one 16-bit entry per step, PAIR of slots (2j, 2j+1) and lane; low byte = the even slot's jump
  unsigned short *hist16 = (unsigned short *)sa.hist;
  const u32 half_k = K >> 1;
  // ---- prologue: X/Y planes -> workspace (the planes hold canonical values: eight words as they are), prefix
  // products of the first step's dx (ascending) ----
  fe acc;
#pragma unroll
  for (int i = 0; i < 9; ++i) acc.l[i] = (i == 0);
#pragma unroll 1
  for (u32 k = 0; k < K; ++k) {
    const u64 p = T + (u64)k * threads;
    const pk256 xw = pk_from_planes(a.X, a.cap, p), yw = pk_from_planes(a.Y, a.cap, p);
    fe x;
    fe_unpack(x, xw);
    const u32 *row = jt + (x.l[0] & (NB_JUMP - 1)) * SJT_STRIDE;
    fe jx, dx;
    jt_load9(jx, row);
#if KG_JT_NEG
    fe_add_lifted(dx, x, jx);
#else
    fe_sub_wide(dx, x, jx);
#endif
    fe_mul(acc, dx, acc);
    ws_store<0>(wr, k, xw);
    ws_store<2>(wr, k, yw);
    if (!KG_WS_PAIRS || (k & (WS_GROUP - 1)) == WS_GROUP - 1) {
      pk256 qw;
      fe_settle<2, 2>(acc);
      fe_pack(qw, acc);
      ws_store<4>(wr, k, qw);
      ws_hold(xw, yw, qw);
    } else {
      ws_hold(xw, yw);
    }
  }

  // one jump of the kangaroo (x, y) in slot id given dx^-1; multiplies its next dx into acc, stores the new
  // point (and, for the second kangaroo of a pair, the running product), reports a distinguished point
  auto jump = [&](const fe &x, const fe &y, const fe &dxi, const fe &jx, const u32 *row, u32 id, u32 run, bool with_p) {
    fe jy;
    jt_load9(jy, row + SJT_GROUP);
    fe dy, s, p2, rx, tt, ry;
#if KG_JT_NEG
    fe_add_lifted(dy, y, jy);
    fe_mul(s, dy, dxi);
    fe_sqr(p2, s);
    {
      fe c4;  // 4p' - jPx, fetched where it is needed: nothing of the jump point stays in registers across the multiplications
      jt_load9(c4, row + 2 * SJT_GROUP);
      fe_add_lifted_sub(rx, p2, c4, x);
    }
#else
    fe_sub_wide(dy, y, jy);
    fe_mul(s, dy, dxi);
    fe_sqr(p2, s);
    fe_sub2_wide(rx, p2, jx, x);
#endif
    fe_canon(rx);
    fe_sub_wide(tt, x, rx);
    fe_mul(ry, tt, s);
    fe_sub_wide(ry, ry, y);
    fe_weak(ry);
    fe_settle<0, 1>(ry);
    pk256 xw, yw;
    fe_pack(xw, rx);
    fe_pack(yw, ry);
    {
      const u32 *row2 = jt + (rx.l[0] & (NB_JUMP - 1)) * SJT_STRIDE;
      fe jx2, dx2;
      jt_load9(jx2, row2);
#if KG_JT_NEG
      fe_add_lifted(dx2, rx, jx2);
#else
      fe_sub_wide(dx2, rx, jx2);
#endif
      fe_mul(acc, dx2, acc);
    }
    ws_store<0>(wr, id, xw);
    ws_store<2>(wr, id, yw);
    if (with_p) {
      pk256 qw;
      fe_settle<2, 2>(acc);
      fe_pack(qw, acc);
      ws_store<4>(wr, id, qw);
      ws_hold(xw, yw, qw);
    } else {
      ws_hold(xw, yw);
    }

    // distinguished point: (x.bits64[3] & dpMask) == 0   (math/gpu/compute.h:54)
    const u64 p = T + (u64)id * threads;
    const bool is_dp = ((xw.hi[3] & a.dp_mask_hi) | (xw.hi[2] & a.dp_mask_lo)) == 0 && p < a.n;
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
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            it->x[i] = xw.lo[i];
            it->x[4 + i] = xw.hi[i];
          }
          it->kidx = a.p_base + p;
          // the distance is filled in by kg_hist_sum_kernel: chain the item to its kangaroo.  Only this lane ever
          // touches the head of kangaroo p; the exchange is an atomic so that it is served by the L2 and never by
          // a stale line of this CU's vector cache
          const u32 prev = atomicExch(sa.dp_head + p, pos);
          sa.dp_link[pos] = DpLink{prev, run};
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
      if (other > run + 1) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
    }
    // operands of the first pair: requested before the inversion, which hides the latency
    StreamRec A, B;
    pk256 QB;
    ws_load<0>(A.x, wr, id_at(0));
    ws_load<2>(A.y, wr, id_at(0));
    ws_load<0>(B.x, wr, id_at(1));
    ws_load<2>(B.y, wr, id_at(1));
#if KG_WS_QUADS
    // QB holds the quad's stored product q_D (in slot of the next quad's first kangaroo); R, S the x of C and D, which
    // the quad's first pair needs for q_B = dx_C * dx_D * q_D
    pk256 R, S;
    ws_load<0>(R, wr, id_at(2));
    ws_load<0>(S, wr, id_at(3));
    ws_load<4>(QB, wr, K > 4 ? id_at(4) : id_at(3));
    fe qC, qD;  // live from the first pair of a quad to its second
#pragma unroll
    for (int i = 0; i < 9; ++i) qC.l[i] = qD.l[i] = 0;
#else
    ws_load<4>(QB, wr, K > 2 ? id_at(2) : id_at(1));
#endif
#if !KG_WS_PAIRS
    pk256 QA;  // product of the dx still to come after A: what B's slot holds from the previous pass
    ws_load<4>(QA, wr, id_at(1));
#endif
    fe u;
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
#if KG_INV_TREE
        {
          // Product tree over the lanes of this wave.  Level j pairs lane l (l a multiple of 2^(j+1)) with lane
          // l + 2^j; tree[j] keeps what every lane held when the level started.  Only this wave is here (the other
          // three wait at the barrier below), its LDS operations execute in program order, and the accesses are
          // volatile so that the compiler keeps that order: no further synchronisation is needed inside a wave.
          volatile u32 *tree = inv_ex + 8 * 9 * 64 + 4;  // [6][9][64]
          fe cur = t;
#pragma unroll 1
          for (u32 j = 0; j < KG_INV_TREE_LEVELS; ++j) {
            volatile u32 *lvl = tree + j * (9 * 64);
#pragma unroll
            for (int i = 0; i < 9; ++i) lvl[i * 64 + lane] = cur.l[i];
            __builtin_amdgcn_wave_barrier();
            if ((lane & ((2u << j) - 1u)) == 0u) {
              fe o;
#pragma unroll
              for (int i = 0; i < 9; ++i) o.l[i] = lvl[i * 64 + lane + (1u << j)];
              fe_mul(cur, cur, o);
            }
          }
          if ((lane & ((1u << KG_INV_TREE_LEVELS) - 1u)) == 0u) fe_inv_t<KG_INV_TREE_MODE>(iv, cur);  // every other lane is off: the 12 000 instructions of the inversion toggle one lane
#pragma unroll 1
          for (int j = KG_INV_TREE_LEVELS - 1; j >= 0; --j) {
            volatile u32 *lvl = tree + (u32)j * (9 * 64);
            const u32 span = 2u << j, half = 1u << j;
            if ((lane & (span - 1u)) == 0u) {
              // iv = (mine * theirs)^-1 as the level held them: theirs^-1 = iv * mine goes to the partner, mine^-1 = iv * theirs stays
              fe mine, theirs, r2;
#pragma unroll
              for (int i = 0; i < 9; ++i) {
                mine.l[i] = lvl[i * 64 + lane];
                theirs.l[i] = lvl[i * 64 + lane + half];
              }
              fe_mul(r2, iv, mine);
#pragma unroll
              for (int i = 0; i < 9; ++i) lvl[i * 64 + lane + half] = r2.l[i];
              fe_mul(iv, iv, theirs);
            }
            __builtin_amdgcn_wave_barrier();
            if ((lane & (span - 1u)) == half) {
#pragma unroll
              for (int i = 0; i < 9; ++i) iv.l[i] = lvl[i * 64 + lane];
            }
          }
        }
#elif defined(KG_TIMING_NO_INVERSION)
        iv = t;  // TIMING EXPERIMENT ONLY (wrong results): what the inversion's place on the critical path costs
#else
        fe_inv(iv, t);
#endif
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
    // The first operands have long arrived.  Saying so matters: the compiler's wait-count bookkeeping merges
    // the loop entry with the back edge, and a load still "pending" at the entry turns into a wait on the
    // previous body's STORES at the top of every iteration (vmcnt counts loads and stores in order).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
    for (int i = 0; i < 9; ++i) acc.l[i] = (i == 0);

    unsigned short *hist_run = hist16 + (u64)run * half_k * threads + Tl;  // + pair * threads
#pragma unroll 1
    for (u32 s = 0; s < K; s += 2) {
      const u32 idA = id_at(s), idB = id_at(s + 1);
      const bool last = s + 2 >= K;
      // The pair's packed operands were requested one iteration ago (the first pair's before the inversion).
      // Each is unpacked where it is needed and its registers are then reloaded IN PLACE with the next pair's
      // (clamped at the end of the pass: valid slots, values unused): no copies at the back edge, and the loads
      // go out in three short bursts (x and product, y of A, y of B) instead of one long one — the per-CU
      // memory queue is what waves block on.
      const u32 s2 = last ? s : s + 2;
      fe xA, xB, qB;
      fe_unpack(xA, A.x);
      fe_unpack(xB, B.x);
#if KG_WS_QUADS
      const bool first = (s & 2u) == 0;       // first pair (A, B) of its quad; the second is (C, D)
      const u32 q0 = s & ~2u;                 // position of the quad's A
      const bool lastq = q0 + 4 >= K;         // nothing comes after this quad's D: q_D = 1
      fe xC, xD;
      if (first) {
        fe_unpack(xC, R);
        fe_unpack(xD, S);
        if (!lastq) fe_unpack(qD, QB);
        A.x = R;  // C and D are the next pair: their x is already here
        B.x = S;
        __builtin_amdgcn_sched_barrier(0);
        // the next quad's C, D and stored product (clamped at the end of the pass: valid slots, values unused)
        const u32 n0 = lastq ? q0 : q0 + 4;
        ws_load<0>(R, wr, id_at(n0 + 2));
        ws_load<0>(S, wr, id_at(n0 + 3));
        ws_load<4>(QB, wr, n0 + 4 < K ? id_at(n0 + 4) : id_at(n0 + 3));
        __builtin_amdgcn_sched_barrier(0);
      } else {
        __builtin_amdgcn_sched_barrier(0);
        ws_load<0>(A.x, wr, id_at(s2));
        ws_load<0>(B.x, wr, id_at(s2 + 1));
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      fe_unpack(qB, QB);
#if !KG_WS_PAIRS
      fe qA;
      fe_unpack(qA, QA);
#endif
      __builtin_amdgcn_sched_barrier(0);
      ws_load<0>(A.x, wr, id_at(s2));
      ws_load<0>(B.x, wr, id_at(s2 + 1));
#ifdef KG_TIMING_QUADS
      // TIMING EXPERIMENT ONLY (wrong results, never in the library): what a stored product per FOUR kangaroos would buy —
      // every other pair neither loads nor stores a product and pays one more multiplication (profiles/r03_levers.txt)
      // KG_TIMING_QUADS: 1 = no load, no store, one more multiplication; 2 = store skipped only; 3 = load skipped only;
      // 4 = as 1 without the multiplication; 5 = as 1, the multiplication's result unused
      const bool quad_skip = ((s >> 1) & 1u) != 0;
      if (!quad_skip || KG_TIMING_QUADS == 2)
#endif
      ws_load<4>(QB, wr, s2 + 2 < K ? id_at(s2 + 2) : id_at(s2 + 1));
#if !KG_WS_PAIRS
      ws_load<4>(QA, wr, id_at(s2 + 1));
#endif
      __builtin_amdgcn_sched_barrier(0);
#endif  // KG_WS_QUADS

      const u32 jmpA = xA.l[0] & (NB_JUMP - 1), jmpB = xB.l[0] & (NB_JUMP - 1);
      hist_run[(u64)(idB >> 1) * threads] = (unsigned short)(asc ? (jmpA | (jmpB << 8)) : (jmpB | (jmpA << 8)));
      const u32 *rowA = jt + jmpA * SJT_STRIDE, *rowB = jt + jmpB * SJT_STRIDE;
      fe jxA, jxB, dxA, dxB, dxiA, dxiB;
      jt_load9(jxA, rowA);
      jt_load9(jxB, rowB);
#if KG_JT_NEG
      fe_add_lifted(dxA, xA, jxA);
      fe_add_lifted(dxB, xB, jxB);
#else
      fe_sub_wide(dxA, xA, jxA);
      fe_sub_wide(dxB, xB, jxB);
#endif
#if KG_WS_QUADS
      {
        // first pair of a quad: q_C = dx_D * q_D, q_B = dx_C * q_C, q_A = dx_B * q_B (q_C and q_D stay for the second pair);
        // either pair then:     dxi_1st = q_1st * u, u *= dx_1st, dxi_2nd = q_2nd * u, u *= dx_2nd
        fe qF, qS;
        if (first) {
          const u32 *rowC = jt + (xC.l[0] & (NB_JUMP - 1)) * SJT_STRIDE, *rowD = jt + (xD.l[0] & (NB_JUMP - 1)) * SJT_STRIDE;
          fe jxC, jxD, dxC, dxD;
          jt_load9(jxC, rowC);
          jt_load9(jxD, rowD);
#if KG_JT_NEG
          fe_add_lifted(dxC, xC, jxC);
          fe_add_lifted(dxD, xD, jxD);
#else
          fe_sub_wide(dxC, xC, jxC);
          fe_sub_wide(dxD, xD, jxD);
#endif
          if (!lastq) {
            fe_mul(qC, dxD, qD);
          } else {  // nothing comes after D: q_D = 1, q_C = dx_D — as a weak value: it is fe_mul's second operand below
            qC = dxD;
            fe_weak(qC);
          }
          fe_mul(qS, dxC, qC);
          fe_mul(qF, dxB, qS);
        } else {
          qF = qC;
          qS = qD;
        }
        fe_mul(dxiA, qF, u);
        fe_mul(u, dxA, u);
        if (first || !lastq) {
          fe_mul(dxiB, qS, u);
          fe_mul(u, dxB, u);
        } else {  // D of the last quad
          dxiB = u;
        }
      }
#elif KG_WS_PAIRS
      if (!last) {
        fe qA;
        fe_mul(qA, dxB, qB);
        fe_mul(dxiA, qA, u);
        fe_mul(u, dxA, u);
        fe_mul(dxiB, qB, u);
        fe_mul(u, dxB, u);
      } else {  // nothing comes after B: q_B = 1
        fe_mul(dxiA, dxB, u);
        fe_mul(u, dxA, u);
        dxiB = u;
      }
#else
      fe_mul(dxiA, qA, u);
      fe_mul(u, dxA, u);
      if (!last) {
        fe_mul(dxiB, qB, u);
        fe_mul(u, dxB, u);
      } else {  // nothing comes after B
        dxiB = u;
      }
#endif
      {
        fe yA;
        fe_unpack(yA, A.y);
        __builtin_amdgcn_sched_barrier(0);
        ws_load<2>(A.y, wr, id_at(s2));
        __builtin_amdgcn_sched_barrier(0);
        jump(xA, yA, dxiA, jxA, rowA, idA, run, !KG_WS_PAIRS);
      }
      {
        fe yB;
        fe_unpack(yB, B.y);
        __builtin_amdgcn_sched_barrier(0);
        ws_load<2>(B.y, wr, id_at(s2 + 1));
        __builtin_amdgcn_sched_barrier(0);
#ifdef KG_TIMING_QUADS
        if (quad_skip && KG_TIMING_QUADS == 1) fe_mul(acc, acc, dxiB);  // the multiplication a quad would add (any operands: timing only)
        if (quad_skip && KG_TIMING_QUADS == 5) {  // the same multiplication with its result thrown away (the data stay what they are)
          fe t;
          fe_mul(t, dxiA, dxiB);
          asm volatile("" ::"v"(t.l[0]), "v"(t.l[1]), "v"(t.l[2]), "v"(t.l[3]), "v"(t.l[4]), "v"(t.l[5]), "v"(t.l[6]), "v"(t.l[7]), "v"(t.l[8]));
        }
        jump(xB, yB, dxiB, jxB, rowB, idB, run, !quad_skip || KG_TIMING_QUADS == 3);
#elif KG_WS_QUADS
        jump(xB, yB, dxiB, jxB, rowB, idB, run, !first);  // only D stores the running product
#else
        jump(xB, yB, dxiB, jxB, rowB, idB, run, true);
#endif
      }
    }
  } while (++run < a.nb_run);

  // ---- epilogue: workspace -> X/Y planes (canonical); the distances are summed by kg_hist_sum_kernel ----
#pragma unroll 1
  for (u32 k = 0; k < K; ++k) {
    pk256 xw, yw;
    ws_load<0>(xw, wr, k);
    ws_load<2>(yw, wr, k);
    const u64 p = T + (u64)k * threads;
    pk_to_planes(a.X, a.cap, p, xw);  // x is kept canonical throughout
    fe y;
    fe_unpack(y, yw);
    fe_canon(y);
    fe_pack(yw, y);
    pk_to_planes(a.Y, a.cap, p, yw);
  }
  if (blockIdx.x == 0 && tid == 0) {
    const u64 dc = __builtin_amdgcn_s_memtime() - stamp_clk, dr = __builtin_amdgcn_s_memrealtime() - stamp_ref;
    u32 *tail = sa.pace + PACE_KEYS * 4;
    tail[0] = (u32)dc;
    tail[1] = (u32)(dc >> 32);
    tail[2] = (u32)dr;
    tail[3] = (u32)(dr >> 32);
  }
}

// 256-bit value minus a DW-word one (the distance at the end of the launch minus what was travelled after a step)
template <int DW>
__device__ __forceinline__ void sub_dist_256(u32 (&out)[8], const u32 (&end)[8], const dlt<DW> &sfx) {
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 t = (u64)end[i] - (i < DW ? sfx.w[i < DW ? i : 0] : 0u) - br;
    out[i] = (u32)t;
    br = (u32)(t >> 63);
  }
}

// End of launch: D[p] += sum of the jump distances in kangaroo p's history, and the distances of the launch's
// distinguished points.  One thread per PAIR of slots and lane (the history holds the two jump indices of a pair in
// one 16-bit entry), full occupancy, lane-contiguous reads (1 B per jump read once).  A kangaroo that reported
// distinguished points has them chained from dp_head[p], most recent first: its thread walks the history a second
// time, downwards from the last step, and writes  d(item) = D_end - (distance travelled after the item's step)
// into the pinned ring (math/gpu/compute.h:48,57: the reference adds the jump distance before it reports).
template <int DW>
__global__ void __launch_bounds__(WALK_BLOCK) kg_hist_sum_kernel(const unsigned char *hist, const JumpTableDev *jtab, u64 *D,
                                                                 u64 cap, u64 n, u32 threads, u32 K, u32 nb_run,
                                                                 const u32 *dp_head, const DpLink *dp_link, DpItemDev *out) {
  __shared__ u32 jdist[NB_JUMP * DW];
  for (u32 i = threadIdx.x; i < NB_JUMP * DW; i += WALK_BLOCK) jdist[i] = jtab->dist[i / DW][i % DW];
  __syncthreads();
  const u64 q = (u64)blockIdx.x * WALK_BLOCK + threadIdx.x;  // pair * threads + lane
  if (q >= (u64)(K >> 1) * threads) return;
  const u32 pair = (u32)(q / threads), lane = (u32)(q % threads);
  const unsigned short *col = (const unsigned short *)hist + q;  // + step * (K / 2) * threads
  const u64 step_stride = (u64)(K >> 1) * threads;
  dlt<DW> s0, s1;
#pragma unroll
  for (int i = 0; i < DW; ++i) s0.w[i] = s1.w[i] = 0u;
  u32 r = 0;
  for (; r + 8 <= nb_run; r += 8) {
    u32 j[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) j[i] = col[(u64)(r + i) * step_stride];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      delta_add<DW>(s0.w, jdist + (j[i] & 0xFFu) * DW);
      delta_add<DW>(s1.w, jdist + (j[i] >> 8) * DW);
    }
  }
  for (; r < nb_run; ++r) {
    const u32 j = col[(u64)r * step_stride];
    delta_add<DW>(s0.w, jdist + (j & 0xFFu) * DW);
    delta_add<DW>(s1.w, jdist + (j >> 8) * DW);
  }
  u32 e0[8], e1[8];
  const u64 p0 = (u64)(2 * pair) * threads + lane, p1 = p0 + threads;
  add_dist_256<DW>(e0, D, cap, p0, s0);
#pragma unroll
  for (int i = 0; i < 4; ++i) D[(u64)i * cap + p0] = (u64)e0[2 * i] | ((u64)e0[2 * i + 1] << 32);
  add_dist_256<DW>(e1, D, cap, p1, s1);
#pragma unroll
  for (int i = 0; i < 4; ++i) D[(u64)i * cap + p1] = (u64)e1[2 * i] | ((u64)e1[2 * i + 1] << 32);

  // distinguished points of this launch (rare unless the DP size is very low)
  u32 h0 = dp_head[p0], h1 = dp_head[p1];
  if (h0 == DP_NONE && h1 == DP_NONE) return;
#pragma unroll
  for (int i = 0; i < DW; ++i) s0.w[i] = s1.w[i] = 0u;  // now: distance travelled in the steps (r, nb_run)
  DpLink l0 = h0 != DP_NONE ? dp_link[h0] : DpLink{DP_NONE, 0u}, l1 = h1 != DP_NONE ? dp_link[h1] : DpLink{DP_NONE, 0u};
  r = nb_run;  // steps [r, nb_run) are in the suffix sums
  while (h0 != DP_NONE || h1 != DP_NONE) {
    // the later of the two chains' next steps (a chain's steps strictly decrease along prev)
    const u32 t0 = h0 != DP_NONE ? l0.step : 0u, t1 = h1 != DP_NONE ? l1.step : 0u;
    const u32 t = h0 == DP_NONE ? t1 : h1 == DP_NONE ? t0 : (t0 > t1 ? t0 : t1);
    while (r > t + 1) {
      --r;
      const u32 j = col[(u64)r * step_stride];
      delta_add<DW>(s0.w, jdist + (j & 0xFFu) * DW);
      delta_add<DW>(s1.w, jdist + (j >> 8) * DW);
    }
    if (h0 != DP_NONE && l0.step == t) {
      u32 w[8];
      sub_dist_256<DW>(w, e0, s0);
#pragma unroll
      for (int i = 0; i < 8; ++i) out[h0].d[i] = w[i];
      h0 = l0.prev;
      if (h0 != DP_NONE) l0 = dp_link[h0];
    }
    if (h1 != DP_NONE && l1.step == t) {
      u32 w[8];
      sub_dist_256<DW>(w, e1, s1);
#pragma unroll
      for (int i = 0; i < 8; ++i) out[h1].d[i] = w[i];
      h1 = l1.prev;
      if (h1 != DP_NONE) l1 = dp_link[h1];
    }
  }
}

constexpr size_t walk_stream_lds_bytes() {
  return (size_t)(NB_JUMP * SJT_STRIDE + 8 * 9 * 64 + 4 + INV_TREE_WORDS) * sizeof(u32);  // + the pacing slot, + the lane tree
}

}  // namespace kg
