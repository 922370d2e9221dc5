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
//                   one 192-byte record per macroblock (tc0 included).  Also zeroes the pipeline's progress words.
//   k_deblock_rows  one workgroup (one wave) per macroblock row and plane kind (luma | both chroma planes).
//                   It walks the row left to right.  The tile (macroblock + 8 left columns + 4 top rows) lives
//                   in LDS; lane = sample row for the vertical edges, lane = sample column for the
//                   horizontal edges, the four edges of a direction are filtered in registers.  The left
//                   neighbour is carried in LDS.  The top neighbour's bottom rows come from the row above as
//                   8-byte "granules" {tag, four samples}, each written by ONE write-through (sc1) store the
//                   moment the four samples are final (columns 0..11 of a macroblock right after its horizontal
//                   edges, columns 12..15 right after the vertical edges of the next macroblock), and read with
//                   sc1 loads (they bypass the CU's L1) until the tag shows: the data is the flag, so there is no
//                   fence, no drain, no progress counter and no grid barrier, and a row trails the row above by
//                   about one and a half macroblocks.  Every image sample is stored by exactly one row (a row
//                   leaves its bottom four rows to the row below, which filters them across its top edge).
//                   Rows draw their index from an atomic ticket, so a row only ever waits for a row whose
//                   workgroup has already started: no assumption about dispatch order.  Every spin is
//                   bounded; a timeout sets an error word that all rows watch.
//
// Image stores per step (luma; chroma is the same with 8-sample macroblocks and a 2-row top halo):
//   rows 0..11 (0..15 in the bottom macroblock row) x columns 16x-8 .. 16x+7 -- final, nothing later in this row
//   touches them -- and the top neighbour's rows -4..-1 x columns 16x .. 16x+15; after the last step also columns
//   16x+8 .. 16x+15.  All stores are 8-byte aligned.
#include "deblock_common.h"

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(3))) int lds_int;   // the abort flag is polled twice per step: through a generic pointer that is a flat load plus a wait for every
                                                         // outstanding memory operation of the wave; through an LDS pointer it is a ds_read
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct __attribute__((aligned(16))) DbPrep {
  uint8_t bsY[2][4][4];      // [dir][segment][edge] luma strengths, 0 = nothing to do; a lane reads its four edges as one dword
  uint8_t c0Y[2][4][4];      // CLIP_TAB[indexA][bS] of the same segments (loop_filter.h:39-45)
  uint8_t bsC[2][4][4];      // chroma strengths (indexed by the luma edge number)
  uint8_t c0C[2][2][4][4];   // [U,V][dir][segment][edge]
  uint8_t ab[3][3][2];       // [Y,U,V][left edge, top edge, internal][alpha, beta]
  uint8_t pad_[14];
};                           // 192 bytes = 12 x 16
#define PREP_VEC 12

#ifndef DB_SPARSE
__global__ __launch_bounds__(256) void k_deblock_prep(const jmhip_db_mb *__restrict__ mbs, const jmhip_db_motion *__restrict__ motion,
                                                      int mb_w, int mb_h, int fmt, int direct8x8, DbPrep *__restrict__ prep,
                                                      unsigned *__restrict__ sync, int nsync, unsigned long long *__restrict__ hand,
                                                      const uint8_t *__restrict__ Y, int pitchY, const uint8_t *__restrict__ U,
                                                      const uint8_t *__restrict__ V, int pitchC, int lr, int cr, uint8_t *__restrict__ flags)
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
  // alpha / beta / indexA of this edge: the neighbour across edge 0, the macroblock itself inside (DeblockMb / EdgeLoop*: QP average)
  const jmhip_db_mb *p = edge ? q : (dir == 0 ? (mbx ? &mbs[addr - 1] : q) : (mby ? &mbs[addr - mb_w] : q));
  const int which = edge ? 2 : dir;
  const int sc = S > 3 ? 3 : S;
#pragma unroll
  for (int comp = 0; comp < 3; comp++) {
    const int qq = comp ? q->qpc[comp - 1] : q->qp, qp = comp ? p->qpc[comp - 1] : p->qp;
    const int QP = (qp + qq + 1) >> 1;
    const int iA = clip3(0, 51, QP + q->df_alpha_c0), iB = clip3(0, 51, QP + q->df_beta);
    if (comp == 0) o->c0Y[dir][idx][edge] = c_tc0[iA][sc];
    else o->c0C[comp - 1][dir][idx][edge] = c_tc0[iA][sc];
    if (idx == 0 && edge < 2) { o->ab[comp][which][0] = c_alpha[iA]; o->ab[comp][which][1] = c_beta[iB]; }
  }
  const int vY = non8x8 ? S : 0, vC = ecr >= 0 ? S : 0;
  o->bsY[dir][idx][edge] = (uint8_t)vY;
  o->bsC[dir][idx][edge] = (uint8_t)vC;
  if (l < 14) o->pad_[l] = 0;
  // Hand-over granules of this macroblock (its bottom rows, for the band below): tag 0 = not there yet.  Where the band that owns the
  // macroblock cannot change those rows -- no edge segment of the macroblock is active and neither is the left edge of its right
  // neighbour, which would reach into its last three columns -- they are what the picture holds now, so the granule is issued right
  // here and the band below does not wait for the band above at this column.  In a P picture of mostly skipped macroblocks that
  // removes the wavefront start-up between bands; an intra picture (every edge active) is untouched.  Conservative: an active
  // segment need not change a sample (alpha / beta tests), and only segments covering the bottom rows could.
  int S2 = 0;                                          // left edge of the right neighbour, segment idx (lanes l < 4: dir 0, edge 0)
  if (l < 4 && mbx + 1 < mb_w) {
    const jmhip_db_mb *q2 = q + 1;
    const bool left_ok2 = q2->df_disable_idc == 2 ? q2->slice_nr == q->slice_nr : true;
    if (q2->df_disable_idc != 1 && left_ok2) S2 = strength_of(0, 0, idx, addr + 1, mb_w, mbs, motion);
  }
  const int half = (tid >> 5) & 1;
  const unsigned ownY = (unsigned)(__ballot(vY != 0) >> (32 * half)), ownC = (unsigned)(__ballot(vC != 0) >> (32 * half)), nbr = (unsigned)(__ballot(S2 != 0) >> (32 * half));
  const unsigned anyY = ownY | nbr, anyC = ownC | nbr;
  // for the segment walks (deblock_sparse.hip): bit 0 = some segment of this macroblock is active, bit 1 = its left edge is (luma or chroma:
  // lanes 0..3 hold dir 0, edge 0) -- where it is not, nothing connects this macroblock to its left neighbour and a row can be cut
  if (flags && l == 0) flags[addr] = (uint8_t)(((ownY | ownC) != 0) | ((((ownY | ownC) & 0xfu) != 0) << 1));
  if (l < 16) {
    const bool pre = lr > 0 && mby % lr == lr - 1 && mby + 1 < mb_h && anyY == 0;
    unsigned long long v = 0;
    if (pre) v = (1ull << 32) | *(const uint32_t *)(Y + (long)(16 * mby + 12 + (l >> 2)) * pitchY + 16 * mbx + 4 * (l & 3));
    hand[(long)addr * 24 + l] = v;
  } else if (l < 24) {
    const int qg = l - 16, RH = fmt == 2 ? 16 : 8;
    const bool pre = cr > 0 && (fmt == 1 || fmt == 2) && mby % cr == cr - 1 && mby + 1 < mb_h && anyC == 0;
    unsigned long long v = 0;
    if (pre) v = (1ull << 32) | *(const uint32_t *)(((qg >> 2) ? V : U) + (long)(RH * mby + RH - 2 + ((qg >> 1) & 1)) * pitchC + 8 * mbx + 4 * (qg & 1));
    hand[(long)addr * 24 + l] = v;
  }
}

#endif  // !DB_SPARSE

// ---------------------------------------------------------------------------------------------------
// Edge filters on registers, branch-free: a step of the pipeline is a chain of eight dependent edge filters executed by a
// single wave, so its duration is the number of instructions issued; exec-mask branches around four-instruction bodies cost
// more than the bodies.  Every lane computes the normal (bS < 4) filter, the strong (bS = 4) filter is added only when some
// lane of the wave needs it (wave-uniform branch), and v_cndmask selects per sample.
__device__ __forceinline__ int med3i(int a, int b, int c)
{
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ int absdiff(int a, int b) { return (int)__builtin_amdgcn_sad_u16((unsigned)a, (unsigned)b, 0u); }   // samples are 0..255

// p[O..O+7] = L3 L2 L1 L0 | R0 R1 R2 R3   (EdgeLoopLumaVer/Hor, loop_filter_normal.c:301-581)
template <int O, int N>
__device__ __forceinline__ void luma_edge(int (&p)[N], int bS, int alpha, int beta, int c0, bool any_strong)
{
  const int L3 = p[O], L2 = p[O + 1], L1 = p[O + 2], L0 = p[O + 3], R0 = p[O + 4], R1 = p[O + 5], R2 = p[O + 6], R3 = p[O + 7];
  const int diff = R0 - L0, ad = absdiff(R0, L0);
  const bool on = ((int)(bS != 0) & (int)(ad < alpha) & (int)(absdiff(R0, R1) < beta) & (int)(absdiff(L0, L1) < beta)) != 0;
  const bool apb = absdiff(L0, L2) < beta, aqb = absdiff(R0, R2) < beta;
  // bS < 4
  const int RL0 = (L0 + R0 + 1) >> 1;
  const int tc = c0 + (int)apb + (int)aqb;
  const int dif = med3i(-tc, tc, ((diff << 2) + (L1 - R1) + 4) >> 3);
  int nL2 = L2, nR2 = R2;
  // both arms of every selection are computed and pinned (FORCE): left alone the compiler turns "cond ? five instructions : x" into an
  // exec-mask branch, which costs this single wave more issue slots (s_and_saveexec, s_cbranch, s_or, a copy) than the arm itself
#define FORCE(v) asm volatile("" : "+v"(v))
  int tL1 = L1 + med3i(-c0, c0, (L2 + RL0 - (L1 << 1)) >> 1), tR1 = R1 + med3i(-c0, c0, (R2 + RL0 - (R1 << 1)) >> 1);
  FORCE(tL1); FORCE(tR1);
  int nL1 = apb ? tL1 : L1, nR1 = aqb ? tR1 : R1;
  int nL0 = med3i(0, 255, L0 + dif), nR0 = med3i(0, 255, R0 - dif);
  if (any_strong) {                                                           // bS == 4 somewhere in the wave
    const bool st = bS == 4, sg = ad < ((alpha >> 2) + 2);
    const bool ap = apb && sg, aq = aqb && sg;
    const int S = L0 + R0;
    int aL0 = (R1 + ((L1 + S) << 1) + L2 + 4) >> 3, bL0 = ((L1 << 1) + L0 + R1 + 2) >> 2, aL1 = (L2 + L1 + S + 2) >> 2, aL2 = (((L3 + L2) << 1) + L2 + L1 + S + 4) >> 3;
    int aR0 = (L1 + ((R1 + S) << 1) + R2 + 4) >> 3, bR0 = ((R1 << 1) + R0 + L1 + 2) >> 2, aR1 = (R2 + R0 + L0 + R1 + 2) >> 2, aR2 = (((R3 + R2) << 1) + R2 + R1 + S + 4) >> 3;
    FORCE(aL0); FORCE(bL0); FORCE(aL1); FORCE(aL2); FORCE(aR0); FORCE(bR0); FORCE(aR1); FORCE(aR2);
    const int sL0 = ap ? aL0 : bL0, sL1 = ap ? aL1 : L1, sL2 = ap ? aL2 : L2;
    const int sR0 = aq ? aR0 : bR0, sR1 = aq ? aR1 : R1, sR2 = aq ? aR2 : R2;
#undef FORCE
    nL0 = st ? sL0 : nL0; nL1 = st ? sL1 : nL1; nL2 = st ? sL2 : nL2;
    nR0 = st ? sR0 : nR0; nR1 = st ? sR1 : nR1; nR2 = st ? sR2 : nR2;
    p[O + 1] = on ? nL2 : L2; p[O + 6] = on ? nR2 : R2;
  }
  p[O + 2] = on ? nL1 : L1; p[O + 3] = on ? nL0 : L0; p[O + 4] = on ? nR0 : R0; p[O + 5] = on ? nR1 : R1;
}
// L1 L0 | R0 R1   (EdgeLoopChromaVer/Hor, loop_filter_normal.c:590-757)
__device__ __forceinline__ void chroma_edge4(int &L1r, int &L0r, int &R0r, int &R1r, int bS, int alpha, int beta, int c0)
{
  const int L1 = L1r, L0 = L0r, R0 = R0r, R1 = R1r, diff = R0 - L0;
  const bool on = ((int)(bS != 0) & (int)(absdiff(R0, L0) < alpha) & (int)(absdiff(R0, R1) < beta) & (int)(absdiff(L0, L1) < beta)) != 0;
  const int tc = c0 + 1, dif = med3i(-tc, tc, ((diff << 2) + (L1 - R1) + 4) >> 3);
  const bool st = bS == 4;
  const int nL0 = st ? ((L1 << 1) + L0 + R1 + 2) >> 2 : med3i(0, 255, L0 + dif);
  const int nR0 = st ? ((R1 << 1) + R0 + L1 + 2) >> 2 : med3i(0, 255, R0 - dif);
  L0r = on ? nL0 : L0; R0r = on ? nR0 : R0;
}

struct RowArgs {
  uint8_t *Y, *U, *V; int pitchY, pitchC;
  const DbPrep *prep; unsigned *sync;          // sync[0] ticket, sync[1] error
  unsigned long long *hand;                    // hand-over granules: per macroblock 16 luma + 8 chroma, zeroed by k_deblock_prep
  int mb_w, mb_h, fmt, nkinds;
  int stride;                                  // macroblocks per picture row: row stride of prep / hand / store_bottom (mb_w is the width of the walk)
  const uint8_t *store_bottom;                 // segment walks (deblock_sparse.hip): per macroblock, nobody below will store its bottom rows; NULL otherwise
  unsigned *ctl;                               // ctl[0] mode (1 = segment walks do the frame), ctl[1] number of segment tasks, ctl[2] their ticket
  const int2 *tasks;                           // segment tasks {row, first column | end column << 16}
};
#define DB_SYNC_ROWINFO 8                      // sync[8 ..]: one 64-bit word per macroblock row for k_deblock_tasks (deblock_sparse.hip)
#define HAND_PER_MB 24                         // 8-byte granules per macroblock: [0,16) luma rows 12..15 x 4 dwords, [16,24) chroma

#define DB_SPIN_LIMIT (1u << 21)

__device__ __forceinline__ void st8(uint8_t *p, uint32_t lo, uint32_t hi) { *(uint2 *)p = make_uint2(lo, hi); }
// a granule = {tag 1, four samples}: ONE 8-byte write-through store, so a reader that sees the tag sees the samples
__device__ __forceinline__ void put_granule(unsigned long long *g, uint32_t samples)
{
  __hip_atomic_store((gu64 *)g, (1ull << 32) | samples, RLX_AGENT);
}
__device__ __forceinline__ unsigned long long get_granule(const unsigned long long *g) { return __hip_atomic_load((gu64 *)g, RLX_AGENT); }
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) { return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24); }

// wave-uniform: re-read this lane's granule (lanes with `mine`) until every tag is set; false on timeout / pipeline error
__device__ __forceinline__ bool await_granules(const unsigned long long *g, bool mine, unsigned long long &v, gu32 *err)
{
  for (unsigned spins = 0;; spins++) {
    if (__all(!mine || (v >> 32) != 0)) return true;
    if ((spins & 31u) == 31u && __hip_atomic_load(err, RLX_AGENT) != 0) return false;
    if (spins > DB_SPIN_LIMIT) { __hip_atomic_store(err, 1u, RLX_AGENT); return false; }
    if (mine) v = get_granule(g);
  }
}

// A row is walked by TWO waves of one workgroup (warp specialisation):
//   the FILTER wave only runs the edge filters on LDS tiles -- per macroblock x: vertical edges V(x), barrier, horizontal
//   edges H(x), barrier -- its instruction stream is the dependent chain that bounds the whole frame, so nothing else is in it;
//   the MOVER wave does every memory operation around it, overlapped with the filters: it prefetches the own samples and the
//   strength record two macroblocks ahead and puts them into the tile ring, polls the top neighbour's hand-over granules and
//   puts the top rows in place (while V(x) runs: the vertical edges need no top rows), stores finished macroblock x-1 to the
//   image as 16-byte rows and hands its bottom rows to the row below (while H(x) runs).
// LDS: a ring of four tiles (macroblock x-1: being stored; x: being filtered; x+1: being filled), each rows -4..15 x 16
// columns; the left neighbour's columns 12..15 are read and written by V(x) directly in tile x-1, so nothing is copied.
// What is final when: after V(x) every sample of macroblock x-1 is final for this row (rows 12..15 still get the row below's
// top edge: they are handed over, not stored); after H(x) rows -4..-1 of tile x hold the top neighbour's final bottom rows,
// which THIS row stores.  Every image sample is stored by exactly one row.
// A failed wait (timeout / pipeline error) raises s_abort; both waves leave at the next barrier.

// ---- luma: FOUR macroblock rows per workgroup.  Only 16 lanes of the filter wave are busy with one row, and a VALU instruction
// costs the same four cycles whatever the lane count, so the same instruction stream filters four rows at once: lanes
// [16g, 16g+16) work on row g of the band, one macroblock behind row g-1 (step s: row g does macroblock s-g; H(x) of row g runs
// in the step in which row g-1 runs V(x+1), exactly the order the 2:1 wavefront needs).  A ring slot is a tall tile: the four
// rows' macroblock of one column stacked (4 + 64 rows x 16 bytes), so row g's "top rows" simply ARE row g-1's bottom rows: inside
// a band nothing is handed over at all; granules connect the last row of a band to the first row of the next band only.
#ifndef LR
#define LR 4                                   // rows per luma workgroup (deblock_sparse.hip compiles this file with LR = 1)
#endif
// tall-tile row R lives at YROW(R): 16 bytes per row plus 16 bytes of padding after every 16 rows, so that the four rows' lanes of a
// column access (same k, rows 16 apart) fall into different LDS banks
#define YROW(R) ((R) * 16 + ((R) >> 4) * 16)
#define YT_BYTES (YROW(4 + 16 * LR) + 16)      // one ring slot
// Phase profiler (profiles/prof_deblock.py --phases; built with -DDB_PROF by profiles/build_dbprof.sh, not part of the product
// library): per luma band and wave, the s_memtime ticks spent waiting at the two barriers of a step and working between them, and
// four 100 MHz wall-clock stamps (loop entry, step 1, step 64, end) for the bands' time line.
#ifdef DB_PROF
__device__ unsigned long long g_db_prof[64 * 2 * 8];
extern "C" void jmhip_debug_read_db_prof(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_db_prof), sizeof(g_db_prof)); }
#define DBP_SLOT ((band * 2 + (threadIdx.x >> 6)) * 8)
#define DBP_ON ((threadIdx.x & 63) == 0 && band < 64)
#define DBP_DECL unsigned long long tp_ = clock64(), ta_[4] = {0, 0, 0, 0}; if (DBP_ON) g_db_prof[DBP_SLOT + 4] = wall_clock64()
#define DBP_STEP(s_) do { if (DBP_ON && ((s_) == 1 || (s_) == 64)) g_db_prof[DBP_SLOT + ((s_) == 1 ? 5 : 6)] = wall_clock64(); } while (0)
#define DBP(i) do { const unsigned long long t_ = clock64(); ta_[i] += t_ - tp_; tp_ = t_; } while (0)
#define DBP_END do { if (DBP_ON) { for (int i_ = 0; i_ < 4; i_++) g_db_prof[DBP_SLOT + i_] = ta_[i_]; g_db_prof[DBP_SLOT + 7] = wall_clock64(); } } while (0)
#else
#define DBP_DECL
#define DBP(i)
#define DBP_STEP(s_)
#define DBP_END
#endif
__device__ void luma_rows(const RowArgs &A, int band, uint8_t *s_tiles /* 4 x YT_BYTES */, uint8_t *s_preps /* LR x 2 x sizeof(DbPrep) */, volatile lds_int *s_abort)
{
  const int tid = threadIdx.x, lane = tid & 63, mb_w = A.mb_w;
  const bool filter_wave = tid < 64, loader_wave = tid >= 128;
  gu32 *err = (gu32 *)(A.sync + 1);
  const int row0 = band * LR, nrows = min(LR, A.mb_h - row0);                  // rows of this band
  const bool has_up = row0 > 0, has_down = row0 + nrows < A.mb_h;
  const int g = lane >> 4, l = lane & 15;                                     // row of the band, lane inside it
  const bool row_ok = g < nrows;
  uint8_t *band_p = A.Y + (long)(16 * row0) * A.pitchY;
  // mover roles
  const uint8_t *own_p = band_p + (long)lane * A.pitchY;                      // lane = sample row of the band (row g, line l)
  const int pg = lane / PREP_VEC, pv = lane - pg * PREP_VEC;                  // record loads: lanes [0, LR*PREP_VEC): row pg, vector pv
  const bool is_pre = lane < LR * PREP_VEC && pg < nrows;
  const DbPrep *pre_p = A.prep + (long)(row0 + pg) * A.stride;
  // polling the band above (loader wave): sixteen granules per column, granule `lane` = top row lane>>2, dword lane&3
  const bool is_gran = lane < 16 && has_up;
  const unsigned long long *hand_up = A.hand + (long)(row0 - 1) * A.stride * HAND_PER_MB + lane;
  unsigned long long *hand_me = A.hand + (long)(row0 + nrows - 1) * A.stride * HAND_PER_MB;
  // image stores: lane L = tall-tile row L = picture row 16*row0 - 4 + L; row g's stores cover its rows -4..11 (the top neighbour's
  // bottom rows it filtered, and its own rows it will not hand on); the frame's last four rows ride on lanes 0..3 of a second store
  uint8_t *store_p = band_p + (long)(lane - 4) * A.pitchY;
  const bool store_ok = row_ok && (lane >= 4 || has_up);
  // rows 12..15 of the band's last row: stored here when no row below will (the frame's last row; in a segment walk the columns whose
  // lower neighbour no walk covers)
  const bool tail_lane = lane < 4 && (!has_down || A.store_bottom);
  const uint8_t *sb_p = (has_down && A.store_bottom) ? A.store_bottom + (long)(row0 + nrows - 1) * A.stride : nullptr;
#define TAIL_OK(X) (tail_lane && (!sb_p || sb_p[X]))
  uint8_t *tail_p = band_p + (long)(16 * nrows - 4 + lane) * A.pitchY;

  const int nsteps = mb_w + nrows - 1;
  // ---- the LOADER wave: own samples and strength records, global -> registers -> tile ring, two columns ahead.  It has its own loop,
  // unrolled by two so that the two register sets alternate by NAME: a register is written to LDS two steps after its load was issued
  // and is never copied -- a copy (or a rotation through a third register) would make the wave wait for a load it issued the step
  // before, and with nothing to filter that wait (the L2/HBM latency) was the length of a step.  Same barriers as the other two waves.
  if (loader_wave) {
    uint4 ownA = make_uint4(0, 0, 0, 0), ownB = ownA, preA = ownA, preB = ownA;
    if (row_ok) *(uint4 *)(s_tiles + YROW(4 + lane)) = *(const uint4 *)own_p;               // column 0 of every row
    if (is_pre) ((uint4 *)(s_preps + pg * 2 * sizeof(DbPrep)))[pv] = ((const uint4 *)pre_p)[pv];
    // column c >= 1 of row g is consumed in step c + g - 1, by the register set of that step's parity
    { const int ca = (g & 1) ? 2 : 1, cb = 3 - ca;
      if (row_ok && ca < mb_w) ownA = *(const uint4 *)(own_p + 16 * ca);
      if (row_ok && cb < mb_w) ownB = *(const uint4 *)(own_p + 16 * cb); }
    { const int ca = (pg & 1) ? 2 : 1, cb = 3 - ca;
      if (is_pre && ca < mb_w) preA = ((const uint4 *)(pre_p + ca))[pv];
      if (is_pre && cb < mb_w) preB = ((const uint4 *)(pre_p + cb))[pv]; }
    // the band above is polled here too: this wave issues loads only, so a wait for an old load never waits for a store (the wave
    // that stores would sit out the write acknowledge of its previous step on every poll).  Four granule registers, re-armed right
    // after use, i.e. four steps before they are needed again; the first look at a register is outside the re-poll loop so that the
    // compiler's wait counts stay exact (inside the loop it must assume the register was just re-loaded)
    unsigned long long gr0 = 0, gr1 = 0, gr2 = 0, gr3 = 0;
    if (is_gran) {
      gr0 = get_granule(hand_up);
      if (mb_w > 1) gr1 = get_granule(hand_up + 1 * HAND_PER_MB);
      if (mb_w > 2) gr2 = get_granule(hand_up + 2 * HAND_PER_MB);
      if (mb_w > 3) gr3 = get_granule(hand_up + 3 * HAND_PER_MB);
    }
#define LSTEP(S, OWN, PRE, GR) { \
    __syncthreads(); if (*s_abort) return; \
    if (has_up && (S) < mb_w) {                              /* the first row's top rows of column S */ \
      if (!__all(!is_gran || (GR >> 32) != 0)) { if (!await_granules(hand_up + (long)(S) * HAND_PER_MB, is_gran, GR, err)) *s_abort = 1; } \
      if (is_gran) { ((uint32_t *)(s_tiles + ((S) & 3) * YT_BYTES + YROW(lane >> 2)))[lane & 3] = (uint32_t)GR; \
                     if ((S) + 4 < mb_w) GR = get_granule(hand_up + (long)((S) + 4) * HAND_PER_MB); } \
    } \
    __syncthreads(); if (*s_abort) return; \
    { const int xo = (S) - g + 1; \
      if (row_ok && xo >= 1 && xo < mb_w) *(uint4 *)(s_tiles + (xo & 3) * YT_BYTES + YROW(4 + lane)) = OWN; \
      if (row_ok && xo >= 1 && xo + 2 < mb_w) OWN = *(const uint4 *)(own_p + 16 * (xo + 2)); } \
    { const int xq = (S) - pg + 1; \
      if (is_pre && xq >= 1 && xq < mb_w) ((uint4 *)(s_preps + (pg * 2 + (xq & 1)) * sizeof(DbPrep)))[pv] = PRE; \
      if (is_pre && xq >= 1 && xq + 2 < mb_w) PRE = ((const uint4 *)(pre_p + xq + 2))[pv]; } }
    for (int s = 0; s < nsteps; s += 4) {
      LSTEP(s, ownA, preA, gr0)
      if (s + 1 < nsteps) LSTEP(s + 1, ownB, preB, gr1)
      if (s + 2 < nsteps) LSTEP(s + 2, ownA, preA, gr2)
      if (s + 3 < nsteps) LSTEP(s + 3, ownB, preB, gr3)
    }
#undef LSTEP
    __syncthreads();
    return;
  }
  DBP_DECL;
  for (int s = 0; s < nsteps; s++) {
    const int x = s - g;                                   // this lane group's macroblock (filter wave); row g is busy iff 0 <= x < mb_w
    const bool busy = row_ok && x >= 0 && x < mb_w;
    uint8_t *tc = s_tiles + (x & 3) * YT_BYTES, *tp = s_tiles + ((x + 3) & 3) * YT_BYTES;
    const DbPrep *P = (const DbPrep *)(s_preps + (g * 2 + (x & 1)) * sizeof(DbPrep));
    DBP(3);
    DBP_STEP(s);
    __syncthreads();                                       // columns s-g are in place for every row; the horizontal edges of step s-1 are done
    DBP(0);
    if (*s_abort) return;
    if (filter_wave) {
      // ---- V(x): lane = sample row, columns -4..15 in registers (-4..-1 live in the slot of x-1)
      if (busy) {
        const int seg = l >> 2;
        const uint32_t bs = *(const uint32_t *)&P->bsY[0][seg][0];
        if (bs) {
          uint32_t *tl = (uint32_t *)(tp + YROW(4 + lane)) + 3;
          uint4 *tr = (uint4 *)(tc + YROW(4 + lane));
          const uint32_t c0 = *(const uint32_t *)&P->c0Y[0][seg][0];
          const uint32_t abE = *(const uint16_t *)&P->ab[0][0][0], abI = *(const uint16_t *)&P->ab[0][2][0];
          const uint4 v = *tr;
          const uint32_t w[5] = {*tl, v.x, v.y, v.z, v.w};
          int p[20];
#pragma unroll
          for (int k = 0; k < 20; k++) p[k] = (w[k >> 2] >> (8 * (k & 3))) & 255;
#define VEDGE(E) { const int bS = (bs >> (8 * (E))) & 255; const uint32_t ab = (E) ? abI : abE; \
                   if (__any(bS)) luma_edge<4 * (E), 20>(p, bS, ab & 255, ab >> 8, (c0 >> (8 * (E))) & 255, __any(bS == 4)); }
          VEDGE(0) VEDGE(1) VEDGE(2) VEDGE(3)
#undef VEDGE
          *tl = pack4(p[0], p[1], p[2], p[3]);
          *tr = make_uint4(pack4(p[4], p[5], p[6], p[7]), pack4(p[8], p[9], p[10], p[11]), pack4(p[12], p[13], p[14], p[15]), pack4(p[16], p[17], p[18], p[19]));
        }
      }
    } else {
      // ---- mover, while the vertical edges run.  Its own view of the step: row r of the band works on column s - r.
      const int xl = s - (nrows - 1);                      // column of the band's last row
      if (has_down && xl >= 1 && xl <= mb_w && lane >= 48) {                  // its bottom rows of column xl-1: columns 0..11 (V(xl) leaves them alone)
        const int r = (lane - 48) >> 2, c4 = (lane - 48) & 3;
        if (c4 < 3) put_granule(hand_me + (long)(xl - 1) * HAND_PER_MB + r * 4 + c4, ((const uint32_t *)(s_tiles + ((xl - 1) & 3) * YT_BYTES + YROW(16 * nrows + r)))[c4]);
      }
    }
    DBP(1);
    __syncthreads();                                       // V done; the band's top rows are in the slot
    DBP(2);
    if (*s_abort) return;
    if (filter_wave) {
      // ---- H(x): lane = sample column; rows -4..15 of row g are tall-tile rows 16g .. 16g+19 of the slot of x
      if (busy) {
        const int seg = l >> 2;
        const uint32_t bs = *(const uint32_t *)&P->bsY[1][seg][0];
        if (bs) {
          const uint32_t c0 = *(const uint32_t *)&P->c0Y[1][seg][0];
          const uint32_t abE = *(const uint16_t *)&P->ab[0][1][0], abI = *(const uint16_t *)&P->ab[0][2][0];
          uint8_t *c = tc + g * (16 * 16 + 16) + l;          // YROW(16 g + k) = g * 272 + k * 16 + (k >> 4) * 16
          int p[20];
#pragma unroll
          for (int k = 0; k < 20; k++) p[k] = c[k * 16 + (k >> 4) * 16];
#define HEDGE(E) { const int bS = (bs >> (8 * (E))) & 255; const uint32_t ab = (E) ? abI : abE; \
                   if (__any(bS)) luma_edge<4 * (E), 20>(p, bS, ab & 255, ab >> 8, (c0 >> (8 * (E))) & 255, __any(bS == 4)); }
          HEDGE(0) HEDGE(1) HEDGE(2) HEDGE(3)
#undef HEDGE
#pragma unroll
          for (int k = 1; k < 19; k++) c[k * 16 + (k >> 4) * 16] = (uint8_t)p[k];
        }
      }
    } else {
      // ---- mover, while the horizontal edges run: column s-g-1 of row g is final for this band (V(s-g) is through its last columns)
      const int xl = s - (nrows - 1);
      if (has_down && xl >= 1 && xl <= mb_w && lane >= 48 && lane < 52)
        put_granule(hand_me + (long)(xl - 1) * HAND_PER_MB + (lane - 48) * 4 + 3, ((const uint32_t *)(s_tiles + ((xl - 1) & 3) * YT_BYTES + YROW(16 * nrows + lane - 48)))[3]);
      {
        const int xs = s - g - 1;                          // store: tall-tile rows 16g .. 16g+15 of the slot of xs = picture rows -4..11 of row g
        if (store_ok && xs >= 0 && xs < mb_w) *(uint4 *)(store_p + 16 * xs) = *(const uint4 *)(s_tiles + (xs & 3) * YT_BYTES + YROW(lane));
        const int xt = s - (nrows - 1) - 1;                // the frame's very last rows
        if (xt >= 0 && xt < mb_w && TAIL_OK(xt)) *(uint4 *)(tail_p + 16 * xt) = *(const uint4 *)(s_tiles + (xt & 3) * YT_BYTES + YROW(16 * nrows + lane));
      }
    }
  }
  DBP_END;
  __syncthreads();                                         // the last horizontal edges are done
  if (!filter_wave) {                                      // every row's last column; the last row's hand-over of it
    const int xe = mb_w - 1;
    const uint8_t *tl = s_tiles + (xe & 3) * YT_BYTES;
    // rows 0..nrows-2 stored their last column in the loop (their step mb_w-1+g+1 <= nsteps-1) except the band's last row
    if (has_down && lane >= 48) put_granule(hand_me + (long)xe * HAND_PER_MB + (lane - 48), ((const uint32_t *)(tl + YROW(16 * nrows + ((lane - 48) >> 2))))[(lane - 48) & 3]);
    if (store_ok && g == nrows - 1) *(uint4 *)(store_p + 16 * xe) = *(const uint4 *)(tl + YROW(lane));
    if (TAIL_OK(xe)) *(uint4 *)(tail_p + 16 * xe) = *(const uint4 *)(tl + YROW(16 * nrows + lane));
  }
#undef TAIL_OK
}

// ---- chroma: both planes, the same band scheme: 64 / (2 RH) rows per workgroup (4 rows at 4:2:0, 2 at 4:2:2); per plane a tall
// tile of 2 + RH * rows sample rows x 8 bytes
#define CT_PLANE (34 * 8)
#define CT_BYTES (2 * CT_PLANE)
__device__ void chroma_rows(const RowArgs &A, int band, uint8_t *s_tiles /* 4 x CT_BYTES */, uint8_t *s_preps, volatile lds_int *s_abort)
{
  const int tid = threadIdx.x, lane = tid & 63, mb_w = A.mb_w, fmt = A.fmt, RH = fmt == 2 ? 16 : 8, CR = LR == 1 ? 1 : 32 / RH;
  const bool filter_wave = tid < 64;
  gu32 *err = (gu32 *)(A.sync + 1);
  const int row0 = band * CR, nrows = min(CR, A.mb_h - row0);
  const bool has_up = row0 > 0, has_down = row0 + nrows < A.mb_h;
  // vertical-edge / own-row lanes: lane -> (row gv of the band, plane uvr, sample row rr)
  const int gv = lane / (2 * RH), uvr = (lane / RH) & 1, rr = lane % RH;
  const bool vrow_ok = gv < nrows;
  const int vt = 2 + RH * gv + rr;                                            // tall-tile row of that sample row
  // horizontal-edge lanes: lane -> (row gh, plane uvc, column cc)
  const int gh = lane >> 4, uvc = (lane >> 3) & 1, cc = lane & 7;
  const bool hcol_ok = gh < nrows && gh < CR;
  // mover roles
  const uint8_t *own_p = (uvr ? A.V : A.U) + (long)(RH * (row0 + gv) + rr) * A.pitchC;
  const int pg = lane / PREP_VEC, pv = lane - pg * PREP_VEC;
  const bool is_pre = lane < CR * PREP_VEC && pg < nrows;
  const DbPrep *pre_p = A.prep + (long)(row0 + pg) * A.stride;
  // polling the band above (loader wave, as in luma_rows): eight granules per column, granule gq = plane (bit 2), top row (bit 1), dword (bit 0)
  const int gq = lane & 7;
  const bool is_gran = lane < 8 && has_up;
  const int goff = ((gq >> 2) & 1) * CT_PLANE + ((gq >> 1) & 1) * 8 + (gq & 1) * 4;
  const unsigned long long *hand_up = A.hand + (long)(row0 - 1) * A.stride * HAND_PER_MB + 16 + gq;
  unsigned long long *hand_me = A.hand + (long)(row0 + nrows - 1) * A.stride * HAND_PER_MB + 16;
  // image stores: lane -> plane su = lane >> 5, tall-tile row st = lane & 31 (row gs = st / RH stores its rows -2 .. RH-3)
  const int su = lane >> 5, st = lane & 31, gs = st / RH;
  const bool store_ok = gs < nrows && (st >= 2 || has_up);
  uint8_t *store_p = (su ? A.V : A.U) + (long)(RH * row0 - 2 + st) * A.pitchC;
  const bool tail_lane = lane < 4 && (!has_down || A.store_bottom);          // the last two rows of each plane of the band's last row (see luma_rows)
  const uint8_t *sb_p = (has_down && A.store_bottom) ? A.store_bottom + (long)(row0 + nrows - 1) * A.stride : nullptr;
#define TAIL_OK(X) (tail_lane && (!sb_p || sb_p[X]))
  uint8_t *tail_p = ((lane >> 1) ? A.V : A.U) + (long)(RH * (row0 + nrows) - 2 + (lane & 1)) * A.pitchC;
  const int tail_off = (lane >> 1) * CT_PLANE + (RH * nrows + (lane & 1)) * 8;

  const int nsteps = mb_w + nrows - 1;
  // ---- the LOADER wave (see luma_rows): own samples of both planes and the strength records, two columns ahead, two register sets
  if (tid >= 128) {
    uint2 ownA = make_uint2(0, 0), ownB = ownA;
    uint4 preA = make_uint4(0, 0, 0, 0), preB = preA;
    if (vrow_ok) *(uint2 *)(s_tiles + uvr * CT_PLANE + vt * 8) = *(const uint2 *)own_p;
    if (is_pre) ((uint4 *)(s_preps + pg * 2 * sizeof(DbPrep)))[pv] = ((const uint4 *)pre_p)[pv];
    { const int ca = (gv & 1) ? 2 : 1, cb = 3 - ca;
      if (vrow_ok && ca < mb_w) ownA = *(const uint2 *)(own_p + 8 * ca);
      if (vrow_ok && cb < mb_w) ownB = *(const uint2 *)(own_p + 8 * cb); }
    { const int ca = (pg & 1) ? 2 : 1, cb = 3 - ca;
      if (is_pre && ca < mb_w) preA = ((const uint4 *)(pre_p + ca))[pv];
      if (is_pre && cb < mb_w) preB = ((const uint4 *)(pre_p + cb))[pv]; }
    unsigned long long gr0 = 0, gr1 = 0, gr2 = 0, gr3 = 0;
    if (is_gran) {
      gr0 = get_granule(hand_up);
      if (mb_w > 1) gr1 = get_granule(hand_up + 1 * HAND_PER_MB);
      if (mb_w > 2) gr2 = get_granule(hand_up + 2 * HAND_PER_MB);
      if (mb_w > 3) gr3 = get_granule(hand_up + 3 * HAND_PER_MB);
    }
#define LSTEP(S, OWN, PRE, GR) { \
    __syncthreads(); if (*s_abort) return; \
    if (has_up && (S) < mb_w) { \
      if (!__all(!is_gran || (GR >> 32) != 0)) { if (!await_granules(hand_up + (long)(S) * HAND_PER_MB, is_gran, GR, err)) *s_abort = 1; } \
      if (is_gran) { *(uint32_t *)(s_tiles + ((S) & 3) * CT_BYTES + goff) = (uint32_t)GR; \
                     if ((S) + 4 < mb_w) GR = get_granule(hand_up + (long)((S) + 4) * HAND_PER_MB); } \
    } \
    __syncthreads(); if (*s_abort) return; \
    { const int xo = (S) - gv + 1; \
      if (vrow_ok && xo >= 1 && xo < mb_w) *(uint2 *)(s_tiles + (xo & 3) * CT_BYTES + uvr * CT_PLANE + vt * 8) = OWN; \
      if (vrow_ok && xo >= 1 && xo + 2 < mb_w) OWN = *(const uint2 *)(own_p + 8 * (xo + 2)); } \
    { const int xq = (S) - pg + 1; \
      if (is_pre && xq >= 1 && xq < mb_w) ((uint4 *)(s_preps + (pg * 2 + (xq & 1)) * sizeof(DbPrep)))[pv] = PRE; \
      if (is_pre && xq >= 1 && xq + 2 < mb_w) PRE = ((const uint4 *)(pre_p + xq + 2))[pv]; } }
    for (int s = 0; s < nsteps; s += 4) {
      LSTEP(s, ownA, preA, gr0)
      if (s + 1 < nsteps) LSTEP(s + 1, ownB, preB, gr1)
      if (s + 2 < nsteps) LSTEP(s + 2, ownA, preA, gr2)
      if (s + 3 < nsteps) LSTEP(s + 3, ownB, preB, gr3)
    }
#undef LSTEP
    __syncthreads();
    return;
  }
  for (int s = 0; s < nsteps; s++) {
    __syncthreads();
    if (*s_abort) return;
    if (filter_wave) {
      // ---- vertical edges (luma edges 0 and 2 -> chroma columns 0 and 4): lane = (row, plane, sample row), columns -4..7
      const int x = s - gv;
      if (vrow_ok && x >= 0 && x < mb_w) {
        const DbPrep *P = (const DbPrep *)(s_preps + (gv * 2 + (x & 1)) * sizeof(DbPrep));
        const int seg = RH == 8 ? (rr >> 1) : (rr >> 2);
        const uint32_t bs = *(const uint32_t *)&P->bsC[0][seg][0] & 0x00ff00ffu;
        if (bs) {
          uint32_t *tl = (uint32_t *)(s_tiles + ((x + 3) & 3) * CT_BYTES + uvr * CT_PLANE + vt * 8) + 1;
          uint2 *tr = (uint2 *)(s_tiles + (x & 3) * CT_BYTES + uvr * CT_PLANE + vt * 8);
          const uint32_t c0 = *(const uint32_t *)&P->c0C[uvr][0][seg][0];
          const uint32_t abE = *(const uint16_t *)&P->ab[1 + uvr][0][0], abI = *(const uint16_t *)&P->ab[1 + uvr][2][0];
          const uint2 v = *tr;
          const uint32_t w[3] = {*tl, v.x, v.y};
          int p[12];
#pragma unroll
          for (int k = 0; k < 12; k++) p[k] = (w[k >> 2] >> (8 * (k & 3))) & 255;
          chroma_edge4(p[2], p[3], p[4], p[5], bs & 255, abE & 255, abE >> 8, c0 & 255);
          chroma_edge4(p[6], p[7], p[8], p[9], (bs >> 16) & 255, abI & 255, abI >> 8, (c0 >> 16) & 255);
          *tl = pack4(p[0], p[1], p[2], p[3]);
          *tr = make_uint2(pack4(p[4], p[5], p[6], p[7]), pack4(p[8], p[9], p[10], p[11]));
        }
      }
    } else {
      const int xl = s - (nrows - 1);
      if (has_down && xl >= 1 && xl <= mb_w && lane >= 48 && lane < 56 && !(lane & 1)) {          // last row's bottom two rows of column xl-1, columns 0..3
        const int q = lane - 48, uv = q >> 2, r = (q >> 1) & 1;
        put_granule(hand_me + (long)(xl - 1) * HAND_PER_MB + q, *(const uint32_t *)(s_tiles + ((xl - 1) & 3) * CT_BYTES + uv * CT_PLANE + (RH * nrows + r) * 8));
      }
    }
    __syncthreads();
    if (*s_abort) return;
    if (filter_wave) {
      // ---- horizontal edges: lane = (row, plane, column); rows -2..RH-1 of row gh are tall-tile rows RH*gh .. RH*gh+RH+1
      const int x = s - gh;
      if (hcol_ok && x >= 0 && x < mb_w) {
        const DbPrep *P = (const DbPrep *)(s_preps + (gh * 2 + (x & 1)) * sizeof(DbPrep));
        const int seg = cc >> 1;
        const uint32_t bs = *(const uint32_t *)&P->bsC[1][seg][0];
        if (bs) {
          const uint32_t c0 = *(const uint32_t *)&P->c0C[uvc][1][seg][0];
          const uint32_t abE = *(const uint16_t *)&P->ab[1 + uvc][1][0], abI = *(const uint16_t *)&P->ab[1 + uvc][2][0];
          uint8_t *c = s_tiles + (x & 3) * CT_BYTES + uvc * CT_PLANE + (RH * gh) * 8 + cc;
          int p[18];
#pragma unroll
          for (int k = 0; k < 18; k++) p[k] = (k < RH + 2) ? c[k * 8] : 0;
          // chroma_edge[1][e][fmt]: 4:2:0 -> rows 0 (e=0), 4 (e=2); 4:2:2 -> rows 0, 4, 8, 12 (e = 0..3)
#define CHEDGE(E, ROW) { const uint32_t ab = (E) ? abI : abE; \
                         chroma_edge4(p[ROW], p[(ROW) + 1], p[(ROW) + 2], p[(ROW) + 3], (bs >> (8 * (E))) & 255, ab & 255, ab >> 8, (c0 >> (8 * (E))) & 255); }
          CHEDGE(0, 0)
          if (fmt == 1) { CHEDGE(2, 4) }
          else { CHEDGE(1, 4) CHEDGE(2, 8) CHEDGE(3, 12) }
#undef CHEDGE
#pragma unroll
          for (int k = 1; k < 17; k++) if (k < RH + 1) c[k * 8] = (uint8_t)p[k];
        }
      }
    } else {
      const int xl = s - (nrows - 1);
      if (has_down && xl >= 1 && xl <= mb_w && lane >= 48 && lane < 56 && (lane & 1)) {           // columns 4..7 after V(xl)
        const int q = lane - 48, uv = q >> 2, r = (q >> 1) & 1;
        put_granule(hand_me + (long)(xl - 1) * HAND_PER_MB + q, *((const uint32_t *)(s_tiles + ((xl - 1) & 3) * CT_BYTES + uv * CT_PLANE + (RH * nrows + r) * 8) + 1));
      }
      {
        const int xs = s - gs - 1;
        if (store_ok && xs >= 0 && xs < mb_w) *(uint2 *)(store_p + 8 * xs) = *(const uint2 *)(s_tiles + (xs & 3) * CT_BYTES + su * CT_PLANE + st * 8);
        const int xt = s - (nrows - 1) - 1;
        if (xt >= 0 && xt < mb_w && TAIL_OK(xt)) *(uint2 *)(tail_p + 8 * xt) = *(const uint2 *)(s_tiles + (xt & 3) * CT_BYTES + tail_off);
      }
    }
  }
  __syncthreads();
  if (!filter_wave) {
    const int xe = mb_w - 1;
    const uint8_t *tl = s_tiles + (xe & 3) * CT_BYTES;
    if (has_down && lane >= 48 && lane < 56) {
      const int q = lane - 48, uv = q >> 2, r = (q >> 1) & 1, c4 = q & 1;
      put_granule(hand_me + (long)xe * HAND_PER_MB + q, *((const uint32_t *)(tl + uv * CT_PLANE + (RH * nrows + r) * 8) + c4));
    }
    if (store_ok && gs == nrows - 1) *(uint2 *)(store_p + 8 * xe) = *(const uint2 *)(tl + su * CT_PLANE + st * 8);
    if (TAIL_OK(xe)) *(uint2 *)(tail_p + 8 * xe) = *(const uint2 *)(tl + tail_off);
  }
#undef TAIL_OK
}

#ifndef DB_SPARSE
__global__ __launch_bounds__(192) void k_deblock_rows(RowArgs A)
{
  __shared__ __attribute__((aligned(16))) uint8_t s_tiles[4 * (YT_BYTES > CT_BYTES ? YT_BYTES : CT_BYTES)];
  __shared__ __attribute__((aligned(16))) uint8_t s_preps[LR * 2 * sizeof(DbPrep)];
  __shared__ unsigned s_ticket;
  __shared__ int s_abort;
  const int tid = threadIdx.x;
  if (A.ctl && A.ctl[0] == 1u) return;               // the segment walks (deblock_sparse.hip) do this frame
  if (tid == 0) { s_ticket = __hip_atomic_fetch_add((gu32 *)A.sync, 1u, RLX_AGENT); s_abort = 0; }
  for (int k = tid; k < (int)sizeof(s_tiles) / 4; k += 192) ((uint32_t *)s_tiles)[k] = 0;
  __syncthreads();
  // tickets: luma and chroma bands alternate, top to bottom -- a workgroup only ever waits for one with a smaller ticket,
  // i.e. one that has already started
  const int t = (int)s_ticket, CR = A.fmt == 2 ? 2 : 4;
  const int nl = (A.mb_h + LR - 1) / LR, nc = A.nkinds > 1 ? (A.mb_h + CR - 1) / CR : 0;
  // order: l0 c0 l1 c1 ... while both last, then the rest of the longer list
  const int both = min(nl, nc);
  if (t < 2 * both) { if (t & 1) chroma_rows(A, t >> 1, s_tiles, s_preps, (lds_int *)&s_abort); else luma_rows(A, t >> 1, s_tiles, s_preps, (lds_int *)&s_abort); }
  else if (nl > both) { if (t - both < nl) luma_rows(A, t - both, s_tiles, s_preps, (lds_int *)&s_abort); }
  else if (t - both < nc) chroma_rows(A, t - both, s_tiles, s_preps, (lds_int *)&s_abort);
}

int jmhip_launch_deblock_sparse(jmhip_ctx *ctx, const RowArgs &A, const uint8_t *d_flags, int max_active_pct);      // deblock_sparse.hip

// prep + segment walks / rows on the context's stream; the caller has checked alignment (8-byte planes and pitches)
int jmhip_launch_deblock_rows(jmhip_ctx *ctx, uint8_t *d_Y, int pitchY, uint8_t *d_U, uint8_t *d_V, int pitchC,
                              const jmhip_db_mb *d_mbs, const jmhip_db_motion *d_motion, int direct8x8)
{
  const int mb_w = ctx->W / 16, mb_h = ctx->H / 16, nmb = mb_w * mb_h, fmt = ctx->cfg.yuv_format;
  const int nkinds = fmt ? 2 : 1, nsync = DB_SYNC_ROWINFO + 2 * mb_h;                                         // ticket, error | mode, tasks, task ticket
  const int no_prefill = ctx->db_no_prefill;                                         // A/B switches for profiling and tests (environment, read in jmhip_create)
  // segment walks (deblock_sparse.hip) for pictures with few active macroblocks; the decision is taken on the device, per frame
  const int max_active_pct = ctx->db_sparse_pct;
  const bool sparse_on = !no_prefill && max_active_pct > 0 && ctx->d_db_tasks && mb_w <= 256 && mb_h <= 256;
  const int lr = no_prefill ? 0 : (sparse_on ? 1 : LR), cr = no_prefill ? 0 : (sparse_on ? 1 : (fmt == 2 ? 2 : 4));
  hipLaunchKernelGGL(k_deblock_prep, dim3((nmb + 7) / 8), dim3(256), 0, ctx->stream, d_mbs, d_motion, mb_w, mb_h, fmt, direct8x8,
                     (DbPrep *)ctx->d_db_prep, ctx->d_db_sync, nsync, (unsigned long long *)ctx->d_db_hand,
                     (const uint8_t *)d_Y, pitchY, (const uint8_t *)d_U, (const uint8_t *)d_V, pitchC, lr, cr, sparse_on ? ctx->d_db_flags : nullptr);
  RowArgs A;
  A.Y = d_Y; A.U = d_U; A.V = d_V; A.pitchY = pitchY; A.pitchC = pitchC; A.prep = (const DbPrep *)ctx->d_db_prep; A.sync = ctx->d_db_sync; A.hand = (unsigned long long *)ctx->d_db_hand;
  A.mb_w = mb_w; A.mb_h = mb_h; A.fmt = fmt; A.nkinds = nkinds; A.stride = mb_w;
  A.store_bottom = nullptr; A.ctl = sparse_on ? ctx->d_db_sync + 2 : nullptr; A.tasks = (const int2 *)ctx->d_db_tasks;
  if (sparse_on) {
    RowArgs S = A;
    S.store_bottom = ctx->d_db_flags + nmb;
    jmhip_launch_deblock_sparse(ctx, S, ctx->d_db_flags, max_active_pct);
  }
  hipLaunchKernelGGL(k_deblock_rows, dim3((mb_h + LR - 1) / LR + (nkinds > 1 ? (mb_h + (fmt == 2 ? 2 : 4) - 1) / (fmt == 2 ? 2 : 4) : 0)), dim3(192), 0, ctx->stream, A);
  return JMHIP_OK;
}
#endif  // !DB_SPARSE
