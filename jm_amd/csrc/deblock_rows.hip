// deblock_rows.hip -- K9+K10 as a row pipeline: the whole frame in two launches (gfx950).
//
// Same result, bit for bit, as JM's raster-order in-place DeblockFrame (lencod/src/loopFilter.c:63-297,
// loop_filter_normal.c) and as k_deblock_diag (deblock.hip).
//
// Why a pipeline.  Filtering macroblock (x,y) reads samples its left, top and top-right neighbours have
// finished writing, so the only parallel schedule is the 2:1 wavefront x + 2y: W/16 + 2(H/16-1) dependent
// steps per frame (254 at 1080p).  The path is bound by the latency of one step, not by HBM; a kernel
// launch per step costs far more than the step.  Here the steps of one macroblock ROW run inside one
// single-wave workgroup, and rows hand their bottom samples to the row below through memory:
//
//   k_deblock_prep  (fully parallel) boundary strengths of all 32 edge segments of every macroblock with
//                   DeblockMb's skip rules folded in (strength 0 = leave the segment alone), and the
//                   alpha / beta / indexA of the left edge, the top edge and the internal edges for Y, U, V:
//                   one 96-byte record per macroblock.  Also zeroes the pipeline's progress words.
//   k_deblock_rows  one workgroup (one wave) per macroblock row and plane kind (luma | both chroma planes).
//                   It walks the row left to right.  The tile (macroblock + 8 left columns + 4 top rows) lives
//                   in LDS; lane = sample row for the vertical edges, lane = sample column for the
//                   horizontal edges, the four edges of a direction are filtered in registers.  The left
//                   neighbour is carried in LDS.  The top neighbour's bottom rows come from the row above:
//                   a row publishes "macroblocks 0..k-1 are final in memory" in a progress word after
//                   storing its samples write-through (sc1) and draining its stores; the row below polls
//                   that word (relaxed, agent scope) until it covers the top-right neighbour and reads the
//                   handed-over rows with sc1 loads (they bypass the CU's L1).  No fences, no grid barrier.
//                   Rows draw their index from an atomic ticket, so a row only ever waits for a row whose
//                   workgroup has already started: no assumption about dispatch order.  Every spin is
//                   bounded; a timeout sets an error word that all rows watch.
//
// What is stored when (luma; chroma is the same with 8-sample macroblocks and a 2-row top halo):
//   after step x: rows 0..15 x columns 16x-8 .. 16x+7 (final: nothing later in this row touches them) and the
//   top neighbour's rows -4..-1 x columns 16x .. 16x+15; after the last step also columns 16x+8 .. 16x+15.
//   The row below may start macroblock x once progress >= min(x+2, W/16).  All stores are 8-byte aligned.
#include "deblock_common.h"

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct __attribute__((aligned(16))) DbPrep {
  uint8_t bsY[2][4][4];      // [dir][edge][segment] luma strengths, 0 = nothing to do
  uint8_t bsC[2][4][4];      // the same for chroma (indexed by the luma edge number)
  uint8_t prm[3][3][3];      // [Y,U,V][left edge, top edge, internal][alpha, beta, indexA]
  uint8_t pad_[5];
};                           // 96 bytes

__global__ __launch_bounds__(256) void k_deblock_prep(const jmhip_db_mb *__restrict__ mbs, const jmhip_db_motion *__restrict__ motion,
                                                      int mb_w, int mb_h, int fmt, int direct8x8, DbPrep *__restrict__ prep,
                                                      unsigned *__restrict__ sync, int nsync)
{
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) for (int k = tid; k < nsync; k += 256) sync[k] = 0;
  const int addr = blockIdx.x * 8 + (tid >> 5), l = tid & 31;
  if (addr >= mb_w * mb_h) return;
  const jmhip_db_mb *q = &mbs[addr];
  const int mbx = addr % mb_w, mby = addr / mb_w;
  DbPrep *o = &prep[addr];
  const int dir = l >> 4, edge = (l >> 2) & 3, idx = l & 3;

  int left_ok = mbx != 0, top_ok = mby != 0;                                  // DeblockMb, loopFilter.c:150-165
  if (q->df_disable_idc == 2) {
    left_ok = mbx != 0 && mbs[addr - 1].slice_nr == q->slice_nr;
    top_ok  = mby != 0 && mbs[addr - mb_w].slice_nr == q->slice_nr;
  }
  const int t8 = q->transform8x8, cbp = q->cbp, mbt = q->mb_type, st = q->slice_type;
  const int non8x8 = (edge & 1) ? !t8 : 1;
  bool do_edge = q->df_disable_idc != 1;
  if (cbp == 0) {                                                              // loopFilter.c:173-184 / :222-233
    const int skip8 = dir == 0 ? (fmt != 3) : (fmt == 1);
    if (!non8x8 && skip8) do_edge = false;
    if (edge > 0 && (st == 0 || st == 1)) {
      if ((mbt == 0 && st == 0) || mbt == 1 || mbt == (dir == 0 ? 2 : 3)) do_edge = false;
      if ((edge & 1) && (mbt == (dir == 0 ? 3 : 2) || (mbt == 0 && st == 1 && direct8x8))) do_edge = false;
    }
  }
  if (!(edge || (dir == 0 ? left_ok : top_ok))) do_edge = false;
  const int S = do_edge ? strength_of(dir, edge, idx, addr, mb_w, mbs, motion) : 0;
  const int ecr = (fmt == 1 || fmt == 2) ? c_chroma_edge[dir][edge][fmt] : -4;
  o->bsY[dir][edge][idx] = (uint8_t)(non8x8 ? S : 0);
  o->bsC[dir][edge][idx] = (uint8_t)(ecr >= 0 ? S : 0);
  if (l < 9) {
    const int comp = l / 3, which = l - comp * 3;
    const jmhip_db_mb *p = which == 0 ? (mbx ? &mbs[addr - 1] : q) : (which == 1 ? (mby ? &mbs[addr - mb_w] : q) : q);
    const int qq = comp ? q->qpc[comp - 1] : q->qp, qp = comp ? p->qpc[comp - 1] : p->qp;
    const int QP = (qp + qq + 1) >> 1;
    const int iA = clip3(0, 51, QP + q->df_alpha_c0), iB = clip3(0, 51, QP + q->df_beta);
    o->prm[comp][which][0] = c_alpha[iA]; o->prm[comp][which][1] = c_beta[iB]; o->prm[comp][which][2] = (uint8_t)iA;
  }
  if (l < 5) o->pad_[l] = 0;
}

// ---------------------------------------------------------------------------------------------------
// edge filters on registers: p[O..O+7] = L3 L2 L1 L0 | R0 R1 R2 R3   (EdgeLoopLumaVer/Hor, loop_filter_normal.c:301-581)
template <int O, int N>
__device__ __forceinline__ void luma_edge(int (&p)[N], int bS, int alpha, int beta, int c0)
{
  const int L3 = p[O], L2 = p[O + 1], L1 = p[O + 2], L0 = p[O + 3], R0 = p[O + 4], R1 = p[O + 5], R2 = p[O + 6], R3 = p[O + 7];
  const int diff = R0 - L0, ad = iabs_(diff);
  if (ad < alpha && iabs_(R0 - R1) < beta && iabs_(L0 - L1) < beta) {
    const int aqb = iabs_(R0 - R2) < beta, apb = iabs_(L0 - L2) < beta;
    if (bS == 4) {
      const int small_gap = ad < ((alpha >> 2) + 2);
      const int aq = aqb & small_gap, ap = apb & small_gap, RL0 = L0 + R0;
      if (ap) {
        p[O + 3] = (R1 + ((L1 + RL0) << 1) + L2 + 4) >> 3;
        p[O + 2] = (L2 + L1 + RL0 + 2) >> 2;
        p[O + 1] = (((L3 + L2) << 1) + L2 + L1 + RL0 + 4) >> 3;
      } else p[O + 3] = ((L1 << 1) + L0 + R1 + 2) >> 2;
      if (aq) {
        p[O + 4] = (L1 + ((R1 + RL0) << 1) + R2 + 4) >> 3;
        p[O + 5] = (R2 + R0 + L0 + R1 + 2) >> 2;
        p[O + 6] = (((R3 + R2) << 1) + R2 + R1 + RL0 + 4) >> 3;
      } else p[O + 4] = ((R1 << 1) + R0 + L1 + 2) >> 2;
    } else {
      const int RL0 = (L0 + R0 + 1) >> 1;
      const int tc = c0 + apb + aqb;
      const int dif = clip3(-tc, tc, ((diff << 2) + (L1 - R1) + 4) >> 3);
      if (apb) p[O + 2] = L1 + clip3(-c0, c0, (L2 + RL0 - (L1 << 1)) >> 1);
      p[O + 3] = clip3(0, 255, L0 + dif);
      p[O + 4] = clip3(0, 255, R0 - dif);
      if (aqb) p[O + 5] = R1 + clip3(-c0, c0, (R2 + RL0 - (R1 << 1)) >> 1);
    }
  }
}
// p[O..O+3] = L1 L0 | R0 R1   (EdgeLoopChromaVer/Hor, loop_filter_normal.c:590-757)
__device__ __forceinline__ void chroma_edge4(int &L1r, int &L0r, int &R0r, int &R1r, int bS, int alpha, int beta, int c0)
{
  const int L1 = L1r, L0 = L0r, R0 = R0r, R1 = R1r, diff = R0 - L0;
  if (iabs_(diff) < alpha && iabs_(R0 - R1) < beta && iabs_(L0 - L1) < beta) {
    if (bS == 4) {
      L0r = ((L1 << 1) + L0 + R1 + 2) >> 2;
      R0r = ((R1 << 1) + R0 + L1 + 2) >> 2;
    } else {
      const int tc = c0 + 1, dif = clip3(-tc, tc, ((diff << 2) + (L1 - R1) + 4) >> 3);
      L0r = clip3(0, 255, L0 + dif);
      R0r = clip3(0, 255, R0 - dif);
    }
  }
}

struct RowArgs {
  uint8_t *Y, *U, *V; int pitchY, pitchC;
  const DbPrep *prep; unsigned *sync;          // sync[0] ticket, sync[1] error, sync[2 + kind*mb_h + row] progress
  int mb_w, mb_h, fmt, nkinds;
};

#define DB_SPIN_LIMIT (1u << 21)

// wave-uniform: wait until *flag >= need; false on timeout / pipeline error
__device__ __forceinline__ bool wait_progress(gu32 *flag, unsigned need, gu32 *err)
{
  for (unsigned spins = 0;; spins++) {
    if (__hip_atomic_load(flag, RLX_AGENT) >= need) return true;
    if ((spins & 31u) == 31u && __hip_atomic_load(err, RLX_AGENT) != 0) return false;
    if (spins > DB_SPIN_LIMIT) { __hip_atomic_store(err, 1u, RLX_AGENT); return false; }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ void publish(gu32 *flag, unsigned value, int lane)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the write-through stores of this wave have landed
  if (lane == 0) __hip_atomic_store(flag, value, RLX_AGENT);
}
__device__ __forceinline__ void st8(uint8_t *p, uint32_t lo, uint32_t hi)
{
  __hip_atomic_store((gu64 *)p, ((unsigned long long)hi << 32) | lo, RLX_AGENT);        // global_store_dwordx2 ... sc1
}
__device__ __forceinline__ uint2 ld8_sc1(const uint8_t *p)
{
  const unsigned long long v = __hip_atomic_load((gu64 *)p, RLX_AGENT);
  return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) { return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24); }

// ---- luma row: tile rows -4..15 (index +4), columns -8..15 (byte index +8), pitch 24 bytes
#define YP 24
__device__ void luma_row(const RowArgs &A, int row, uint8_t *s_tile, const uint8_t *s_tc0, uint8_t *s_prep)
{
  const int lane = threadIdx.x, mb_w = A.mb_w;
  uint32_t *tile = (uint32_t *)s_tile;
  gu32 *err = (gu32 *)(A.sync + 1), *prog = (gu32 *)(A.sync + 2 + row), *up = (gu32 *)(A.sync + 2 + row - 1);
  uint8_t *rowp = A.Y + (long)(16 * row) * A.pitchY;
  const DbPrep *prow = A.prep + (long)row * mb_w;
  bool alive = true;

  uint4 own = make_uint4(0, 0, 0, 0), pre = make_uint4(0, 0, 0, 0);            // prefetched: own sample row / 16 bytes of the prep record
  if (lane < 16) own = *(const uint4 *)(rowp + (long)lane * A.pitchY);
  else if (lane >= 32 && lane < 38) pre = ((const uint4 *)prow)[lane - 32];

  for (int x = 0; x < mb_w && alive; x++) {
    // ---- top neighbour rows: wait for the row above to have finished macroblock x+1
    uint2 t0 = make_uint2(0, 0), t1 = make_uint2(0, 0);
    if (row > 0) {
      alive = wait_progress(up, (unsigned)min(x + 2, mb_w), err);
      if (!alive) break;
      if (lane >= 16 && lane < 20) {
        const uint8_t *tp = rowp + (long)(lane - 20) * A.pitchY + 16 * x;
        t0 = ld8_sc1(tp); t1 = ld8_sc1(tp + 8);
      }
    }
    // ---- tile: own rows, top rows, prep record
    if (lane < 16) { uint32_t *d = tile + (lane + 4) * 6 + 2; d[0] = own.x; d[1] = own.y; d[2] = own.z; d[3] = own.w; }
    else if (lane < 20) { uint32_t *d = tile + (lane - 16) * 6 + 2; d[0] = t0.x; d[1] = t0.y; d[2] = t1.x; d[3] = t1.y; }
    else if (lane >= 32 && lane < 38) ((uint4 *)s_prep)[lane - 32] = pre;
    __syncthreads();
    // prefetch the next macroblock's own rows and prep record (nobody else writes them before this row does)
    if (x + 1 < mb_w) {
      if (lane < 16) own = *(const uint4 *)(rowp + (long)lane * A.pitchY + 16 * (x + 1));
      else if (lane >= 32 && lane < 38) pre = ((const uint4 *)(prow + x + 1))[lane - 32];
    }
    const DbPrep *P = (const DbPrep *)s_prep;
    // ---- vertical edges: lane = sample row, columns -4..15 in registers
    if (lane < 16) {
      uint32_t *t = tile + (lane + 4) * 6 + 1;
      const uint32_t w0 = t[0], w1 = t[1], w2 = t[2], w3 = t[3], w4 = t[4];
      const uint32_t bs4 = *(const uint32_t *)&P->bsY[0][0][0] | *(const uint32_t *)&P->bsY[0][1][0] | *(const uint32_t *)&P->bsY[0][2][0] | *(const uint32_t *)&P->bsY[0][3][0];
      if (bs4) {
        int p[20];
        const uint32_t w[5] = {w0, w1, w2, w3, w4};
#pragma unroll
        for (int k = 0; k < 20; k++) p[k] = (w[k >> 2] >> (8 * (k & 3))) & 255;
        const int seg = lane >> 2;
#define VEDGE(E) { const int bS = P->bsY[0][E][seg]; if (bS) { const uint8_t *q = P->prm[0][(E) ? 2 : 0]; \
                     luma_edge<4 * (E), 20>(p, bS, q[0], q[1], s_tc0[q[2] * 4 + (bS > 3 ? 3 : bS)]); } }
        VEDGE(0) VEDGE(1) VEDGE(2) VEDGE(3)
#undef VEDGE
#pragma unroll
        for (int k = 0; k < 5; k++) t[k] = pack4(p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3]);
      }
    }
    __syncthreads();
    // ---- horizontal edges: lane = sample column, rows -4..15 in registers
    if (lane < 16) {
      const uint32_t bs4 = *(const uint32_t *)&P->bsY[1][0][0] | *(const uint32_t *)&P->bsY[1][1][0] | *(const uint32_t *)&P->bsY[1][2][0] | *(const uint32_t *)&P->bsY[1][3][0];
      if (bs4) {
        uint8_t *c = s_tile + 8 + lane;
        int p[20];
#pragma unroll
        for (int k = 0; k < 20; k++) p[k] = c[k * YP];
        const int seg = lane >> 2;
#define HEDGE(E) { const int bS = P->bsY[1][E][seg]; if (bS) { const uint8_t *q = P->prm[0][(E) ? 2 : 1]; \
                     luma_edge<4 * (E), 20>(p, bS, q[0], q[1], s_tc0[q[2] * 4 + (bS > 3 ? 3 : bS)]); } }
        HEDGE(0) HEDGE(1) HEDGE(2) HEDGE(3)
#undef HEDGE
#pragma unroll
        for (int k = 1; k < 19; k++) c[k * YP] = (uint8_t)p[k];
      }
    }
    __syncthreads();
    // ---- write-through stores, carry, publish
    if (lane < 16) {
      uint32_t *t = tile + (lane + 4) * 6;
      uint8_t *g = rowp + (long)lane * A.pitchY + 16 * x;
      if (x > 0) st8(g - 8, t[0], t[1]);
      st8(g, t[2], t[3]);
      if (x == mb_w - 1) st8(g + 8, t[4], t[5]);
      t[0] = t[4]; t[1] = t[5];                                               // columns 8..15 become the next macroblock's -8..-1
    } else if (lane < 20 && row > 0) {
      const uint32_t *t = tile + (lane - 16) * 6 + 2;
      uint8_t *g = rowp + (long)(lane - 20) * A.pitchY + 16 * x;
      st8(g, t[0], t[1]); st8(g + 8, t[2], t[3]);
    }
    publish(prog, (unsigned)(x + 1), lane);
    __syncthreads();
  }
}

// ---- chroma row: both planes; per plane tile rows -2..RH-1 (index +2), columns -8..7 (byte index +8), pitch 16 bytes
#define CPB 16
__device__ void chroma_row(const RowArgs &A, int row, uint8_t *s_tile /* 2 planes x 18 rows x 16 */, const uint8_t *s_tc0, uint8_t *s_prep)
{
  const int lane = threadIdx.x, mb_w = A.mb_w, fmt = A.fmt, RH = fmt == 2 ? 16 : 8;
  gu32 *err = (gu32 *)(A.sync + 1), *prog = (gu32 *)(A.sync + 2 + A.mb_h + row), *up = (gu32 *)(A.sync + 2 + A.mb_h + row - 1);
  const DbPrep *prow = A.prep + (long)row * mb_w;
  // roles: lanes [0, 2RH): (plane, sample row) for loads / vertical edges / stores; lanes [0,16): (plane, column) for horizontal edges
  const int uvr = lane / RH, rr = lane - uvr * RH;               // valid for lane < 2*RH
  const bool is_row = lane < 2 * RH;
  uint8_t *plane_r = (uvr ? A.V : A.U) + (long)(RH * row) * A.pitchC;
  const int uvc = (lane >> 3) & 1, cc = lane & 7;                // valid for lane < 16
  const bool is_top = lane >= 32 && lane < 36;                   // lanes 32..35: (plane, top row -2 / -1)
  const int uvt = (lane - 32) >> 1, tr = (lane - 32) & 1;
  uint8_t *plane_t = (uvt ? A.V : A.U) + (long)(RH * row) * A.pitchC;
  bool alive = true;

  uint2 own = make_uint2(0, 0); uint4 pre = make_uint4(0, 0, 0, 0);
  if (is_row) own = *(const uint2 *)(plane_r + (long)rr * A.pitchC);
  if (lane >= 40 && lane < 46) pre = ((const uint4 *)prow)[lane - 40];

  for (int x = 0; x < mb_w && alive; x++) {
    uint2 t0 = make_uint2(0, 0);
    if (row > 0) {
      alive = wait_progress(up, (unsigned)min(x + 2, mb_w), err);
      if (!alive) break;
      if (is_top) t0 = ld8_sc1(plane_t + (long)(tr - 2) * A.pitchC + 8 * x);
    }
    if (is_row) { uint32_t *d = (uint32_t *)(s_tile + uvr * 18 * CPB + (rr + 2) * CPB + 8); d[0] = own.x; d[1] = own.y; }
    if (is_top) { uint32_t *d = (uint32_t *)(s_tile + uvt * 18 * CPB + tr * CPB + 8); d[0] = t0.x; d[1] = t0.y; }
    if (lane >= 40 && lane < 46) ((uint4 *)s_prep)[lane - 40] = pre;
    __syncthreads();
    if (x + 1 < mb_w) {
      if (is_row) own = *(const uint2 *)(plane_r + (long)rr * A.pitchC + 8 * (x + 1));
      if (lane >= 40 && lane < 46) pre = ((const uint4 *)(prow + x + 1))[lane - 40];
    }
    const DbPrep *P = (const DbPrep *)s_prep;
    // ---- vertical edges (luma edges 0 and 2 -> chroma columns 0 and 4): lane = (plane, row), columns -4..7
    if (is_row) {
      const uint32_t bs4 = *(const uint32_t *)&P->bsC[0][0][0] | *(const uint32_t *)&P->bsC[0][2][0];
      if (bs4) {
        uint32_t *t = (uint32_t *)(s_tile + uvr * 18 * CPB + (rr + 2) * CPB + 4);
        const uint32_t w[3] = {t[0], t[1], t[2]};
        int p[12];
#pragma unroll
        for (int k = 0; k < 12; k++) p[k] = (w[k >> 2] >> (8 * (k & 3))) & 255;
        const int seg = RH == 8 ? (rr >> 1) : (rr >> 2);
        { const int bS = P->bsC[0][0][seg]; if (bS) { const uint8_t *q = P->prm[1 + uvr][0];
            chroma_edge4(p[2], p[3], p[4], p[5], bS, q[0], q[1], s_tc0[q[2] * 4 + (bS > 3 ? 3 : bS)]); } }
        { const int bS = P->bsC[0][2][seg]; if (bS) { const uint8_t *q = P->prm[1 + uvr][2];
            chroma_edge4(p[6], p[7], p[8], p[9], bS, q[0], q[1], s_tc0[q[2] * 4 + (bS > 3 ? 3 : bS)]); } }
#pragma unroll
        for (int k = 0; k < 3; k++) t[k] = pack4(p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3]);
      }
    }
    __syncthreads();
    // ---- horizontal edges: lane = (plane, column), rows -2..RH-1
    if (lane < 16) {
      const uint32_t bs4 = *(const uint32_t *)&P->bsC[1][0][0] | *(const uint32_t *)&P->bsC[1][1][0] | *(const uint32_t *)&P->bsC[1][2][0] | *(const uint32_t *)&P->bsC[1][3][0];
      if (bs4) {
        uint8_t *c = s_tile + uvc * 18 * CPB + 8 + cc;
        int p[18];
#pragma unroll
        for (int k = 0; k < 18; k++) p[k] = (k < RH + 2) ? c[k * CPB] : 0;
        const int seg = cc >> 1;
        // chroma_edge[1][e][fmt]: 4:2:0 -> rows 0 (e=0), 4 (e=2); 4:2:2 -> rows 0, 4, 8, 12 (e = 0..3)
#define CHEDGE(E, ROW) { const int bS = P->bsC[1][E][seg]; if (bS) { const uint8_t *q = P->prm[1 + uvc][(E) ? 2 : 1]; \
                           chroma_edge4(p[ROW], p[(ROW) + 1], p[(ROW) + 2], p[(ROW) + 3], bS, q[0], q[1], s_tc0[q[2] * 4 + (bS > 3 ? 3 : bS)]); } }
        CHEDGE(0, 0)
        if (fmt == 1) { CHEDGE(2, 4) }
        else { CHEDGE(1, 4) CHEDGE(2, 8) CHEDGE(3, 12) }
#undef CHEDGE
#pragma unroll
        for (int k = 1; k < 17; k++) if (k < RH + 1) c[k * CPB] = (uint8_t)p[k];
      }
    }
    __syncthreads();
    if (is_row) {
      uint32_t *t = (uint32_t *)(s_tile + uvr * 18 * CPB + (rr + 2) * CPB);
      uint8_t *g = plane_r + (long)rr * A.pitchC + 8 * x;
      if (x > 0) st8(g - 8, t[0], t[1]);
      if (x == mb_w - 1) st8(g, t[2], t[3]);
      t[0] = t[2]; t[1] = t[3];
    }
    if (is_top && tr == 1 && row > 0) {
      const uint32_t *t = (const uint32_t *)(s_tile + uvt * 18 * CPB + CPB + 8);
      st8(plane_t - (long)A.pitchC + 8 * x, t[0], t[1]);
    }
    publish(prog, (unsigned)(x + 1), lane);
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void k_deblock_rows(RowArgs A)
{
  __shared__ __attribute__((aligned(16))) uint8_t s_tile[2 * 18 * CPB > 20 * YP ? 2 * 18 * CPB : 20 * YP];
  __shared__ __attribute__((aligned(16))) uint8_t s_prep[sizeof(DbPrep)];
  __shared__ uint8_t s_tc0[52 * 4];
  __shared__ unsigned s_ticket;
  const int lane = threadIdx.x;
  if (lane == 0) s_ticket = __hip_atomic_fetch_add((gu32 *)A.sync, 1u, RLX_AGENT);
  for (int k = lane; k < 52 * 4; k += 64) s_tc0[k] = c_tc0[k >> 2][k & 3];
  for (int k = lane; k < (int)sizeof(s_tile) / 4; k += 64) ((uint32_t *)s_tile)[k] = 0;
  __syncthreads();
  const int t = (int)s_ticket, row = t / A.nkinds, kind = t - row * A.nkinds;
  if (row >= A.mb_h) return;
  if (kind == 0) luma_row(A, row, s_tile, s_tc0, s_prep);
  else chroma_row(A, row, s_tile, s_tc0, s_prep);
}

// prep + rows on the context's stream; the caller has checked alignment (8-byte planes and pitches)
int jmhip_launch_deblock_rows(jmhip_ctx *ctx, uint8_t *d_Y, int pitchY, uint8_t *d_U, uint8_t *d_V, int pitchC,
                              const jmhip_db_mb *d_mbs, const jmhip_db_motion *d_motion, int direct8x8)
{
  const int mb_w = ctx->W / 16, mb_h = ctx->H / 16, nmb = mb_w * mb_h, fmt = ctx->cfg.yuv_format;
  const int nkinds = fmt ? 2 : 1, nsync = 2 + 2 * mb_h;
  hipLaunchKernelGGL(k_deblock_prep, dim3((nmb + 7) / 8), dim3(256), 0, ctx->stream, d_mbs, d_motion, mb_w, mb_h, fmt, direct8x8,
                     (DbPrep *)ctx->d_db_prep, ctx->d_db_sync, nsync);
  RowArgs A;
  A.Y = d_Y; A.U = d_U; A.V = d_V; A.pitchY = pitchY; A.pitchC = pitchC; A.prep = (const DbPrep *)ctx->d_db_prep; A.sync = ctx->d_db_sync;
  A.mb_w = mb_w; A.mb_h = mb_h; A.fmt = fmt; A.nkinds = nkinds;
  hipLaunchKernelGGL(k_deblock_rows, dim3(nkinds * mb_h), dim3(64), 0, ctx->stream, A);
  return JMHIP_OK;
}
