// me_fast.hip -- K1+K2+K3, tuned path for search ranges <= 32 (gfx950).
//
// Same result, bit for bit, as k_me_fullsearch (me_fullsearch.hip) -- i.e. as JM's
// full_search_motion_estimation (lencod/src/me_fullsearch.c:39-103) /
// fast_full_search_motion_estimation (lencod/src/me_fullfast.c:618-689) over the window, all 41 partitions --
// but organised around what the MI355X VALU can issue (profiles/r01_valu_rates.txt):
//   v_sad_u8        4 abs-diff+acc per lane-op at the full VOP3 rate (the SAD roofline: ~145 T abs-diff/s)
//   v_add3_u32      1 op builds a packed (cost, tie-break) key
//   v_min3_u32      1 op folds TWO candidate keys into the running minimum
//   64-bit integer compare+select is 4.5x slower, v_alignbyte costs as much as the SAD it feeds.
//
// Mapping.  lane <-> window column (dx); a wave walks window rows two at a time; eight waves share one job (two jobs fit a CU's
// LDS, so 4 waves per SIMD at <= 128 VGPRs):
//   * the search window is staged in LDS as EIGHT byte-shifted copies, so the lane at column wx reads
//     its 16 reference bytes of a row as two ALIGNED ds_read_b64 from copy (wx & 7): no v_alignbyte, and
//     the copy stride (== 8 dwords mod 64) spreads the 32 lanes of a read over all 64 banks;
//   * the current macroblock is fetched with scalar loads and stays in SGPRs: it is the scalar operand
//     of v_sad_u8;
//   * the 16 4x4 SADs are pre-shifted (<<12) once, the 25 aggregation adds then produce the 41
//     partition SADs already in key position;
//   * key = (SAD << 12) + (lambda*mvbits_y << 7) + rank, built by ONE v_add3_u32: the y rate is uniform
//     per row (a per-job LDS table, read as a broadcast b128), the x rate is constant per lane and is
//     added once at the end; rank (7 bits) orders the positions of one column exactly as JM's spiral
//     does (mv_search.c:405-442), so the unsigned minimum reproduces "first spiral index wins";
//   * two rows per step share a column, so one v_min3_u32 retires two candidates.
// ~1.5 VALU ops per (position, partition) instead of ~12 in the generic kernel.
//
// The last window row (2R+1 is odd) is a single-row step; for R = 32 the 65th column is evaluated by
// one wave with the generic per-position code.  Per-lane minima are merged across the waves
// (same lane == same column, keys comparable), decoded to (cost, spiral index) and min-reduced.
// All 41 partitions are always evaluated (no branches in the hot loop); part_mask selects what is stored.
//
// Jobs this kernel cannot take (search_range > 32, a max_mvd guard that could trigger, lambda too large
// for the 32-bit key) are left to k_me_fullsearch: both kernels evaluate job_is_fast() and exactly one
// of them processes a job.
#include "jmhip_internal.h"
#include "me_common.h"

#define WROWS_MAX 80                        // 2*32 + 16
#define CPITCH 20                           // dwords per row of one shifted copy (columns 0..64 + 15 -> 80 bytes)
#define CSTRIDE (WROWS_MAX * CPITCH + 8)    // dwords between copies: == 8 (mod 64) banks
#define NCOPY 8
#define RYP 44                              // row pitch (dwords) of the y-rate table: 41 partitions padded to 11 x uint4
#define COPY_DWORDS (NCOPY * CSTRIDE)
#define NW 8                                 // waves per job: 2 jobs per CU (LDS) x 8 waves = 4 waves per SIMD (<= 128 VGPRs)
#define NT (NW * 64)
#define NWH (NW / 2)                         // the waves' minima are merged in two rounds so the merge area fits into the copy area
#define MERGE_DWORDS (NWH * NP * 64 + NP * 66)

__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c)
{
  unsigned r;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// 16 4x4 SADs for two vertically adjacent positions (rows wy, wy+1) of this lane's column
template <bool TWO>
__device__ __forceinline__ void sad_pair(const uint2 *cp /* lane's copy + column offset, 8-byte aligned */, int wy, const uint32_t (&cw)[64],
                                         unsigned (&sa)[16], unsigned (&sb)[16])
{
#pragma unroll
  for (int k = 0; k < 16; k++) { sa[k] = 0; sb[k] = 0; }
  const uint2 *row = cp + wy * (CPITCH / 2);
  // the two 8-byte halves are read by separate ds_read_b64: with the copy stride of 8 banks the 32 lanes of a half-wave cover all
  // 64 banks exactly once per read, whereas one 16-byte read at 8-byte alignment makes neighbouring lanes' windows overlap by two
  // banks (31 % of the LDS cycles were bank conflicts); hiding the adjacency keeps the compiler from merging them
  const uint2 *rowh = row + 1;
  asm volatile("" : "+v"(rowh));
#pragma unroll
  for (int r = 0; r < (TWO ? 17 : 16); r++) {
    const uint2 lo = row[0], hi = rowh[0];
    if (r < 16) {
      const int q = (r >> 2) * 4;
      sa[q + 0] = __builtin_amdgcn_sad_u8(lo.x, cw[r * 4 + 0], sa[q + 0]);
      sa[q + 1] = __builtin_amdgcn_sad_u8(lo.y, cw[r * 4 + 1], sa[q + 1]);
      sa[q + 2] = __builtin_amdgcn_sad_u8(hi.x, cw[r * 4 + 2], sa[q + 2]);
      sa[q + 3] = __builtin_amdgcn_sad_u8(hi.y, cw[r * 4 + 3], sa[q + 3]);
    }
    if (TWO && r >= 1) {
      const int q = ((r - 1) >> 2) * 4;
      sb[q + 0] = __builtin_amdgcn_sad_u8(lo.x, cw[(r - 1) * 4 + 0], sb[q + 0]);
      sb[q + 1] = __builtin_amdgcn_sad_u8(lo.y, cw[(r - 1) * 4 + 1], sb[q + 1]);
      sb[q + 2] = __builtin_amdgcn_sad_u8(hi.x, cw[(r - 1) * 4 + 2], sb[q + 2]);
      sb[q + 3] = __builtin_amdgcn_sad_u8(hi.y, cw[(r - 1) * 4 + 3], sb[q + 3]);
    }
    row += CPITCH / 2; rowh += CPITCH / 2;
  }
}

// 16 4x4 SADs -> 41 partition SADs, everything shifted left by 12 (ABI partition order, jmhip.h)
__device__ __forceinline__ void aggregate41_shifted(const unsigned (&s7)[16], unsigned (&sp)[NP])
{
#pragma unroll
  for (int k = 0; k < 16; k++) sp[25 + k] = s7[k] << 12;
#pragma unroll
  for (int by = 0; by < 4; by++) { sp[9 + by * 2] = sp[25 + by * 4] + sp[26 + by * 4]; sp[10 + by * 2] = sp[27 + by * 4] + sp[28 + by * 4]; }
#pragma unroll
  for (int bx = 0; bx < 4; bx++) { sp[17 + bx] = sp[25 + bx] + sp[29 + bx]; sp[21 + bx] = sp[33 + bx] + sp[37 + bx]; }
  sp[5] = sp[9] + sp[11];  sp[6] = sp[10] + sp[12];  sp[7] = sp[13] + sp[15];  sp[8] = sp[14] + sp[16];
  sp[1] = sp[5] + sp[6];   sp[2] = sp[7] + sp[8];
  sp[3] = sp[5] + sp[7];   sp[4] = sp[6] + sp[8];
  sp[0] = sp[1] + sp[2];
}

// rank of row dy inside column |dx| = a, in JM's spiral order: 0 .. 2R.
// |dy| <= a: the column entries of ring a come in ascending dy; |dy| > a: ring |dy|, (dx,-|dy|) before (dx,+|dy|).
__device__ __forceinline__ unsigned col_rank(int dy, int a)
{
  const int u = dy < 0 ? -dy : dy;
  return (unsigned)(u <= a ? dy + a : 2 * u - 1 + (dy > 0));
}
__device__ __forceinline__ int rank_to_dy(unsigned rank, int a)
{
  if ((int)rank <= 2 * a) return (int)rank - a;
  const int t = (int)rank + 1, u = t >> 1;
  return (t & 1) ? u : -u;
}

struct __attribute__((packed)) unaligned_u32 { uint32_t v; };

// (cost, spiral index) as a positive normal double whose ordering equals the ordering of the pair:
// v_min_f64 is a full-rate 64-bit minimum on gfx950, 64-bit integer compare+select is 4.5x slower
__device__ __forceinline__ double key_as_double(unsigned cost, unsigned idx)
{
  return __hiloint2double((int)(cost | 0x40000000u), (int)idx);
}

__global__ __launch_bounds__(NT, 4) void k_me_fs_fast(const jmhip_me_job *__restrict__ jobs, jmhip_me_result *__restrict__ results,
                                                       const uint8_t *__restrict__ cur, int cur_pitch,
                                                       const uint8_t *__restrict__ ref00, int pitch, long plane_stride, int W, int H,
                                                       const int16_t *__restrict__ spiral, unsigned *__restrict__ declined, int njobs)
{
  __shared__ __attribute__((aligned(16))) uint32_t s_mem[COPY_DWORDS > MERGE_DWORDS ? COPY_DWORDS : MERGE_DWORDS];   // window copies, later the wave-merge area
  __shared__ __attribute__((aligned(16))) uint32_t s_ry[65 * RYP];            // (lambda * mvbits(cand_y - pred_y[p])) << 7, [row][partition]
  __shared__ uint32_t s_rx64[RYP];                                              // x rate of column 64, per partition
  const int jb = xcd_job_index(blockIdx.x, njobs);
  const jmhip_me_job *__restrict__ job = jobs + jb;
  if (!job_is_fast(job)) {                                  // left to k_me_fullsearch, which only does real work when this counter is non-zero
    if (threadIdx.x == 0) atomicAdd(declined, 1u);
    return;
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = job->search_range, n1 = 2 * R + 1, wrows = 2 * R + 16;
  const int cx = job->center_x, cy = job->center_y, lambda = job->lambda;

  // ---- stage 1: the raw window (= copy 0), rows/cols clamped into the padded plane of the centre's phase
  {
    const int x0 = job->mb_x + (cx >> 2) - R, y0 = job->mb_y + (cy >> 2) - R;
    const uint8_t *plane = ref00 + ((cy & 3) * 4 + (cx & 3)) * plane_stride;
    const bool inside = x0 >= -JMHIP_PAD_X && x0 + CPITCH * 4 + 8 <= W + JMHIP_PAD_X;      // no horizontal clamp needed
    for (int k = tid; k < wrows * (CPITCH + 2); k += NT) {
      const int r = k / (CPITCH + 2), c = k - r * (CPITCH + 2);                            // 22 dwords per row: bytes 0..87
      const int yy = min(max(y0 + r, -JMHIP_PAD_Y), H + JMHIP_PAD_Y - 1) + JMHIP_PAD_Y;
      const uint8_t *prow = plane + (long)yy * pitch + JMHIP_PAD_X;
      uint32_t v;
      if (inside) v = ((const unaligned_u32 *)(prow + x0 + 4 * c))->v;
      else {
        v = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) v |= (uint32_t)prow[min(max(x0 + 4 * c + t, -JMHIP_PAD_X), W + JMHIP_PAD_X - 1)] << (8 * t);
      }
      // raw rows are kept 22 dwords wide in the upper part of the copy area (copies 6,7 are built last from registers)
      s_mem[(NCOPY - 2) * CSTRIDE + r * (CPITCH + 2) + c] = v;
    }
    if (tid < NP) s_rx64[tid] = (uint32_t)(lambda * mvbits(cx + 4 * (64 - R) - job->pred[tid][0])) << 7;
    // y-rate table
    for (int k = tid; k < n1 * NP; k += NT) {
      const int wy = k / NP, p = k - wy * NP;
      s_ry[wy * RYP + p] = (uint32_t)(lambda * mvbits(cy + 4 * (wy - R) - job->pred[p][1])) << 7;
    }
  }
  // ---- current macroblock: uniform addresses -> scalar loads, stays in SGPRs
  uint32_t cw[64];
  {
    const uint32_t *cp = (const uint32_t *)(cur + (long)job->mb_y * cur_pitch + job->mb_x);
    const int p4 = cur_pitch >> 2;
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int c = 0; c < 4; c++) cw[r * 4 + c] = cp[r * p4 + c];
    }
  }
  __syncthreads();
  // ---- stage 2: eight byte-shifted copies: copy s, row r, dword k = window bytes 4k+s .. 4k+s+3
  {
    constexpr int NIT = (WROWS_MAX * CPITCH + NT - 1) / NT;
    uint32_t d[NIT][3];                                     // this thread's items, read before anything is overwritten
    const int total = wrows * CPITCH;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int k = tid + NT * it;
      if (k < total) {
        const int r = k / CPITCH, c = k - r * CPITCH;
        const uint32_t *raw = s_mem + (NCOPY - 2) * CSTRIDE + r * (CPITCH + 2) + c;
        d[it][0] = raw[0]; d[it][1] = raw[1]; d[it][2] = raw[2];
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int k = tid + NT * it;
      if (k < total) {
        const uint32_t d0 = d[it][0], d1 = d[it][1], d2 = d[it][2];
        s_mem[0 * CSTRIDE + k] = d0;
        s_mem[1 * CSTRIDE + k] = __builtin_amdgcn_alignbyte(d1, d0, 1);
        s_mem[2 * CSTRIDE + k] = __builtin_amdgcn_alignbyte(d1, d0, 2);
        s_mem[3 * CSTRIDE + k] = __builtin_amdgcn_alignbyte(d1, d0, 3);
        s_mem[4 * CSTRIDE + k] = d1;
        s_mem[5 * CSTRIDE + k] = __builtin_amdgcn_alignbyte(d2, d1, 1);
        s_mem[6 * CSTRIDE + k] = __builtin_amdgcn_alignbyte(d2, d1, 2);
        s_mem[7 * CSTRIDE + k] = __builtin_amdgcn_alignbyte(d2, d1, 3);
      }
    }
  }
  __syncthreads();

  // ---- main pass: lane = column wx (0..63), wave w takes row pairs
  const int wx = lane, dx = wx - R, a = dx < 0 ? -dx : dx;
  const bool col_ok = wx < n1;
  const uint2 *cp = (const uint2 *)(s_mem + (wx & 7) * CSTRIDE + 2 * (wx >> 3));
  unsigned acc[NP];
#pragma unroll
  for (int p = 0; p < NP; p++) acc[p] = 0xffffffffu;

  const int npairs = R;                                  // rows 0 .. 2R-1
  // pairs go round the waves; the two odd jobs (the last window row, the 65th column) go to the waves with the fewest pairs
#pragma unroll 1
  for (int j = wave; j < npairs; j += NW) {
    const int wy = 2 * j, dyA = wy - R;
    unsigned sa[16], sb[16], pa[NP], pb[NP];
    sad_pair<true>(cp, wy, cw, sa, sb);
    aggregate41_shifted(sa, pa);
    aggregate41_shifted(sb, pb);
    const unsigned rkA = col_rank(dyA, a), rkB = col_rank(dyA + 1, a);
    const uint4 *ryA = (const uint4 *)(s_ry + wy * RYP), *ryB = (const uint4 *)(s_ry + (wy + 1) * RYP);
#pragma unroll
    for (int g = 0; g < 11; g++) {
      const uint4 ca = ryA[g], cb = ryB[g];                 // broadcast reads: 4 partitions' row constants
      const unsigned cav[4] = {ca.x, ca.y, ca.z, ca.w}, cbv[4] = {cb.x, cb.y, cb.z, cb.w};
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int p = 4 * g + t;
        if (p < NP) acc[p] = umin3(acc[p], pa[p] + cav[t] + rkA, pb[p] + cbv[t] + rkB);
      }
    }
  }
  if (wave == NW - 4) {                                  // last row (2R): single-row step
    const int wy = 2 * R;
    unsigned sa[16], sb[16], pa[NP];
    sad_pair<false>(cp, wy, cw, sa, sb);
    aggregate41_shifted(sa, pa);
    const unsigned rkA = col_rank(R, a);
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const unsigned k = pa[p] + s_ry[wy * RYP + p] + rkA;
      acc[p] = k < acc[p] ? k : acc[p];
    }
  }
  // The x rate is constant per (lane, partition): it commutes with the minimum over rows, so it is added once, after the
  // four waves' minima have been merged (below), by the wave that owns the partition -- not by every wave.
#pragma unroll
  for (int p = 0; p < NP; p++) acc[p] = col_ok ? acc[p] : 0xffffffffu;

  // ---- 65th column (R = 32): its 65 positions are spread over waves 1..3 (generic per-position code on
  //      copy 0, the plain window); keys use the same (cost << 7 | rank) form with a = 32.
  unsigned ex[NP];
  const int exq = (wave - (NW - 3)) + 3 * lane;          // position (row) index of this lane in column 64: the last three waves
  const bool ex_live = n1 > 64 && wave >= NW - 3 && exq < n1;
  if (n1 > 64 && wave >= NW - 3) {
#pragma unroll
    for (int p = 0; p < NP; p++) ex[p] = 0xffffffffu;
    if (ex_live) {                                       // wx = 64 is 8-byte aligned in copy 0: same aligned-read SAD code
      const uint2 *cp64 = (const uint2 *)(s_mem + 2 * (64 >> 3));
      unsigned sa[16], sb[16], pa[NP];
      sad_pair<false>(cp64, exq, cw, sa, sb);
      aggregate41_shifted(sa, pa);
      const unsigned rk = col_rank(exq - R, 64 - R);
#pragma unroll
      for (int p = 0; p < NP; p++) ex[p] = pa[p] + s_ry[exq * RYP + p] + s_rx64[p] + rk;
    }
  }
  __syncthreads();                                       // every wave is done reading the window copies
  // ---- publish per-lane minima in two rounds (upper half of the waves, then the lower half merged with them):
  //      s_mem[(w*NP + p)*64 + lane], w < NW/2; the 65th column's keys behind them
  uint32_t *s_ext = s_mem + NWH * NP * 64;               // [NP][66]
  if (wave >= NWH) {
#pragma unroll
    for (int p = 0; p < NP; p++) s_mem[((wave - NWH) * NP + p) * 64 + lane] = acc[p];
  }
  if (ex_live) {
#pragma unroll
    for (int p = 0; p < NP; p++) s_ext[p * 66 + exq] = ex[p];
  }
  __syncthreads();
  if (wave < NWH) {
#pragma unroll
    for (int p = 0; p < NP; p++) acc[p] = min(acc[p], s_mem[(wave * NP + p) * 64 + lane]);
  }
  __syncthreads();
  if (wave < NWH) {
#pragma unroll
    for (int p = 0; p < NP; p++) s_mem[(wave * NP + p) * 64 + lane] = acc[p];
  }
  __syncthreads();
  // ---- final: wave w reduces partitions p = w, w+NW, ...  Lanes merge the waves (same column: keys comparable) and add the
  //      column's x rate.  Across columns only the cost part of a key is comparable; ties between columns are decided by JM's
  //      spiral index.  Ties are rare, so: wave-min of the 32-bit cost; if exactly one candidate attains it, that lane decodes
  //      its position and stores -- otherwise (cost, spiral index) keys as order-preserving doubles are min-reduced.
  {
    const uint64_t mask = job->part_mask;
    const int dxe = 64 - R;
#pragma unroll 1
    for (int p = wave; p < NP; p += NW) {
      unsigned kk = s_mem[p * 64 + lane];
#pragma unroll
      for (int w = 1; w < NWH; w++) kk = min(kk, s_mem[(w * NP + p) * 64 + lane]);
      if (kk != 0xffffffffu) kk += (unsigned)(lambda * mvbits(cx + 4 * dx - job->pred[p][0])) << 7;
      unsigned e = 0xffffffffu, e2 = 0xffffffffu;
      if (n1 > 64) { e = s_ext[p * 66 + lane]; if (lane == 0) e2 = s_ext[p * 66 + 64]; }
      const unsigned c = kk >> 7, ce = e >> 7, ce2 = e2 >> 7;             // 0x1ffffff for "no candidate"
      unsigned m = umin3(c, ce, ce2);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, off, 64));
      const unsigned long long bc = __ballot(c == m), be = __ballot(ce == m), be2 = __ballot(ce2 == m);
      unsigned idx, cost = m;
      bool writer;
      if (__popcll(bc) + __popcll(be) + __popcll(be2) == 1) {              // wave-uniform: a unique minimum
        const bool mine = (c == m) | (ce == m) | (ce2 == m);
        const unsigned key = c == m ? kk : (ce == m ? e : e2);
        const int ddx = c == m ? dx : dxe, aa = c == m ? a : dxe;
        idx = (unsigned)spiral_index(ddx, rank_to_dy(key & 127u, aa));
        writer = mine;
      } else {
        double key = __longlong_as_double(0x7fe0000000000000LL);          // larger than any real key
        if (kk != 0xffffffffu) key = key_as_double(kk >> 7, (unsigned)spiral_index(dx, rank_to_dy(kk & 127u, a)));
        if (e != 0xffffffffu) key = fmin(key, key_as_double(e >> 7, (unsigned)spiral_index(dxe, rank_to_dy(e & 127u, dxe))));
        if (e2 != 0xffffffffu) key = fmin(key, key_as_double(e2 >> 7, (unsigned)spiral_index(dxe, rank_to_dy(e2 & 127u, dxe))));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) key = fmin(key, __shfl_xor(key, off, 64));
        const unsigned long long kb = (unsigned long long)__double_as_longlong(key);
        idx = (unsigned)(kb & 0xffffffffu);
        cost = (unsigned)((kb >> 32) & 0x3fffffffu);
        writer = lane == 0;
      }
      if (writer && ((mask >> p) & 1)) {
        jmhip_me_best b;
        b.mv_x = (int16_t)(cx + 4 * spiral[2 * idx]); b.mv_y = (int16_t)(cy + 4 * spiral[2 * idx + 1]);
        b.cost = (int32_t)cost;
        results[jb].best[p] = b;
      }
    }
  }
}

void jmhip_launch_me_fast(jmhip_ctx *ctx, int slot, const jmhip_me_job *d_jobs, int njobs, jmhip_me_result *d_results, unsigned *d_declined)
{
  hipLaunchKernelGGL(k_me_fs_fast, dim3(njobs), dim3(NT), 0, ctx->stream, d_jobs, d_results, ctx->d_cur, ctx->cur_pitch,
                     ctx->d_sub[slot], ctx->pitch, (long)ctx->plane_stride, ctx->W, ctx->H, ctx->d_spiral, d_declined, njobs);
}
