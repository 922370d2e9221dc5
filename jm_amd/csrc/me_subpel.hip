// me_subpel.hip -- K4: candidate-list SAD / Hadamard-SATD on the 16 quarter-pel planes and the
// 9+9 sub-pel refinement (gfx950).
//
// Device counterpart of (reference, lencod/src):
//   computeSAD              me_distortion.c:349-426   } MEBlock.computePred{F,H,Q}Pel, without the early exit
//   computeSATD             me_distortion.c:745-825   } (result-neutral, SURVEY.md 8a checklist item 2)
//   HadamardSAD4x4 / 8x8    me_distortion.c:175-258 / :266-341   (sum |H D H^T|, (s+1)>>1 / (s+2)>>2)
//   sub_pel_motion_estimation me_fullsearch.c:186-289 (RDOptimization != 0)
//   UMVLine4X               lencod/inc/refbuf.h:22-26: plane = p_curr_img_sub[y&3][x&3], block ORIGIN
//                           clamped to [-20, H+3] x [-32, W+15] (size_*_pad, mbuffer.c:564-565)
//
// Mapping: 16 lanes per block job, lane l < 9 evaluates candidate l of JM's 3x3 spiral; the 16-lane
// group min-reduces (cost, position) keys, which reproduces the sequential strict-'<' scan.
// Reads go straight to the planes (L2 resident: 16 planes x 2.2 MB at 1080p); algorithmic bytes per
// block job: 18 candidates x bsx*bsy reference samples + bsx*bsy current samples.
#include <stdlib.h>
#include "jmhip_internal.h"

struct PlaneSet { const uint8_t *base; int pitch; long plane_stride; int W, H; const unsigned *jerr; };   // jerr: the context's job error word (me_fullsearch.hip)

__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int mvbits(int d) { int a = iabs_(d); return a == 0 ? 1 : 2 * (31 - __clz(a)) + 3; }

// pointer to the sample at quarter-pel position (qx,qy), origin clamped as UMVLine4X does
__device__ __forceinline__ const uint8_t *umv_line(const PlaneSet &ps, int qy, int qx)
{
  int yy = min(max(qy >> 2, -JMHIP_PAD_Y), ps.H + 3), xx = min(max(qx >> 2, -JMHIP_PAD_X), ps.W + 15);
  return ps.base + ((qy & 3) * 4 + (qx & 3)) * ps.plane_stride + (long)(yy + JMHIP_PAD_Y) * ps.pitch + xx + JMHIP_PAD_X;
}

__device__ __forceinline__ int hadamard4(const int d[16])
{
  int m[16], s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int a = d[4 * i], b = d[4 * i + 1], c = d[4 * i + 2], e = d[4 * i + 3];
    int s0 = a + e, s1 = b + c, s2 = b - c, s3 = a - e;
    m[4 * i] = s0 + s1; m[4 * i + 1] = s0 - s1; m[4 * i + 2] = s2 + s3; m[4 * i + 3] = s3 - s2;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int a = m[i], b = m[4 + i], c = m[8 + i], e = m[12 + i];
    int s0 = a + e, s1 = b + c, s2 = b - c, s3 = a - e;
    s += iabs_(s0 + s1) + iabs_(s0 - s1) + iabs_(s2 + s3) + iabs_(s3 - s2);
  }
  return (s + 1) >> 1;
}

__device__ __forceinline__ void had8_1d(int v[8])
{
  int a[8];
#pragma unroll
  for (int i = 0; i < 4; i++) { a[i] = v[i] + v[i + 4]; a[i + 4] = v[i] - v[i + 4]; }
  int b[8] = {a[0] + a[2], a[1] + a[3], a[0] - a[2], a[1] - a[3], a[4] + a[6], a[5] + a[7], a[4] - a[6], a[5] - a[7]};
#pragma unroll
  for (int i = 0; i < 4; i++) { v[2 * i] = b[2 * i] + b[2 * i + 1]; v[2 * i + 1] = b[2 * i] - b[2 * i + 1]; }
}

__device__ int hadamard8(const uint8_t *cur, int cur_pitch, const uint8_t *ref, int pitch)
{
  int m[8][8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (int)cur[j * cur_pitch + i] - (int)ref[(long)j * pitch + i];
    had8_1d(v);
#pragma unroll
    for (int i = 0; i < 8; i++) m[j][i] = v[i];
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = m[j][i];
    had8_1d(v);
#pragma unroll
    for (int j = 0; j < 8; j++) s += iabs_(v[j]);
  }
  return (s + 2) >> 2;
}

// full distortion (not scaled) of one block at absolute quarter-pel position (qx,qy)
__device__ int block_dist(const PlaneSet &ps, const uint8_t *cur, int cur_pitch, int bsx, int bsy, int qx, int qy, int metric, int test8x8)
{
  int acc = 0;
  if (metric == JMHIP_METRIC_SAD) {
    const uint8_t *r = umv_line(ps, qy, qx);
    for (int y = 0; y < bsy; y++)
      for (int x = 0; x < bsx; x++) acc += iabs_((int)cur[y * cur_pitch + x] - (int)r[(long)y * ps.pitch + x]);
  } else if (!test8x8) {
    for (int y = 0; y < bsy; y += 4)
      for (int x = 0; x < bsx; x += 4) {
        const uint8_t *r = umv_line(ps, qy + 4 * y, qx + 4 * x);
        const uint8_t *c = cur + y * cur_pitch + x;
        int d[16];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int i = 0; i < 4; i++) d[4 * j + i] = (int)c[j * cur_pitch + i] - (int)r[(long)j * ps.pitch + i];
        acc += hadamard4(d);
      }
  } else {
    for (int y = 0; y < bsy; y += 8)
      for (int x = 0; x < bsx; x += 8)
        acc += hadamard8(cur + y * cur_pitch + x, cur_pitch, umv_line(ps, qy + 4 * y, qx + 4 * x), ps.pitch);
  }
  return acc;
}

__global__ __launch_bounds__(256) void k_me_eval(const jmhip_cand *__restrict__ cands, int n, int32_t *__restrict__ dist,
                                                 PlaneSet ps, const uint8_t *__restrict__ cur, int cur_pitch)
{
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  jmhip_cand c = cands[i];
  int d = block_dist(ps, cur + (long)c.pos_y * cur_pitch + c.pos_x, cur_pitch, c.bsx, c.bsy,
                     (c.pos_x << 2) + c.cand_x, (c.pos_y << 2) + c.cand_y, c.metric, c.test8x8);
  dist[i] = d << 5;
}

// ---- weighted / bi-predictive candidates: compute{SAD,SSE,SATD}WP, computeBiPred{SAD,SSE,SATD}1 / 2 (me_distortion.c:434-1530)
// One 16-lane group per candidate.  SAD / SSE: lane l sums sample row l (one UMVLine4X origin per reference for the whole block);
// SATD: lane l takes sub-block l (its own origin per reference per sub-block, as the reference fetches them); the group adds up.
struct SlotBases { const uint8_t *p[32]; };
static_assert(sizeof(jmhip_pred_cand) == 32, "jmhip_pred_cand is 32 bytes in include/jmhip.h");

struct PredSrc { const uint8_t *r1, *r2; int pitch, kind, w1, w2, off, rnd, sh; };
__device__ __forceinline__ int pred_at(const PredSrc &p, int j, int i)
{
  const int a = p.r1[(long)j * p.pitch + i];
  if (p.kind == JMHIP_PRED_UNI) return a;
  if (p.kind == JMHIP_PRED_UNI_WP) return min(max(((p.w1 * a + p.rnd) >> p.sh) + p.off, 0), 255);
  const int b = p.r2[(long)j * p.pitch + i];
  if (p.kind == JMHIP_PRED_AVG) return (a + b + 1) >> 1;
  return min(max(((p.w1 * a + p.w2 * b + p.rnd) >> p.sh) + p.off, 0), 255);
}

__global__ __launch_bounds__(64) void k_me_eval_pred(const jmhip_pred_cand *__restrict__ cands, int n, int32_t *__restrict__ dist, SlotBases slots, int nslots,
                                                     PlaneSet ps, const uint8_t *__restrict__ cur, int cur_pitch)
{
  if (*ps.jerr) return;                                   // a job record failed k_check_subpel_jobs
  const int g = blockIdx.x * 4 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const bool live = g < n;
  const jmhip_pred_cand c = cands[live ? g : 0];
  const bool two = c.pred == JMHIP_PRED_AVG || c.pred == JMHIP_PRED_BI_WP;
  const int s1 = c.slot[0], s2 = two ? c.slot[1] : c.slot[0];
  const bool ok = live && s1 >= 0 && s1 < nslots && s2 >= 0 && s2 < nslots;
  PlaneSet pa = ps, pb = ps;
  pa.base = slots.p[ok ? s1 : 0]; pb.base = slots.p[ok ? s2 : 0];
  PredSrc p; p.pitch = ps.pitch; p.kind = c.pred; p.w1 = c.weight[0]; p.w2 = c.weight[1]; p.off = c.offset; p.rnd = c.round; p.sh = c.shift;
  const uint8_t *cu = cur + (long)c.pos_y * cur_pitch + c.pos_x;
  const int q1x = (c.pos_x << 2) + c.cand_x[0], q1y = (c.pos_y << 2) + c.cand_y[0];
  const int q2x = two ? (c.pos_x << 2) + c.cand_x[1] : q1x, q2y = two ? (c.pos_y << 2) + c.cand_y[1] : q1y;
  int acc = 0;
  if (c.metric != JMHIP_METRIC_SATD) {
    if (l < c.bsy) {
      p.r1 = umv_line(pa, q1y, q1x); p.r2 = umv_line(pb, q2y, q2x);
      for (int x = 0; x < c.bsx; x++) {
        const int d = (int)cu[l * cur_pitch + x] - pred_at(p, l, x);
        acc += c.metric == JMHIP_METRIC_SAD ? iabs_(d) : d * d;
      }
    }
  } else if (!c.test8x8) {
    const int nbx = c.bsx >> 2;
    if (l < nbx * (c.bsy >> 2)) {
      const int y = (l / nbx) * 4, x = (l % nbx) * 4;
      p.r1 = umv_line(pa, q1y + 4 * y, q1x + 4 * x); p.r2 = umv_line(pb, q2y + 4 * y, q2x + 4 * x);
      int d[16];
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) d[4 * j + i] = (int)cu[(y + j) * cur_pitch + x + i] - pred_at(p, j, i);
      acc = hadamard4(d);
    }
  } else {
    const int nbx = c.bsx >> 3;
    if (l < nbx * (c.bsy >> 3)) {
      const int y = (l / nbx) * 8, x = (l % nbx) * 8;
      p.r1 = umv_line(pa, q1y + 4 * y, q1x + 4 * x); p.r2 = umv_line(pb, q2y + 4 * y, q2x + 4 * x);
      // me_distortion.c:1167: computeBiPredSATD2 reads the eighth source sample of a row without advancing, so row j of the sub-block is
      // read j samples early in the block's row-major copy (bsx samples per row): sample k = (y + j) * bsx + x + i - j
      const int slip = c.pred == JMHIP_PRED_BI_WP, lg = c.bsx == 16 ? 4 : 3;
      int m[8][8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        int v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int k = ((y + j) << lg) + x + i - slip * j;
          v[i] = (int)cu[(k >> lg) * cur_pitch + (k & (c.bsx - 1))] - pred_at(p, j, i);
        }
        had8_1d(v);
#pragma unroll
        for (int i = 0; i < 8; i++) m[j][i] = v[i];
      }
      int s = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = m[j][i];
        had8_1d(v);
#pragma unroll
        for (int j = 0; j < 8; j++) s += iabs_(v[j]);
      }
      acc = (s + 2) >> 2;
    }
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 16);
  if (live && l == 0) dist[g] = ok ? acc << 5 : -1;
}

// JM's 3x3 spiral (mv_search.c:405-442 with search_range 1): position index -> (dx,dy)
__device__ __constant__ int8_t c_sp9[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {1, 1}};

// one 16-lane group refines one block; every lane of the group returns the result
__device__ jmhip_me_best subpel_group(const jmhip_subpel_job &j, int l, const PlaneSet &ps, const uint8_t *__restrict__ cur, int cur_pitch)
{
  const uint8_t *c = cur + (long)j.pos_y * cur_pitch + j.pos_x;
  const int pxp = j.pos_x << 2, pyp = j.pos_y << 2;
  int mvx = j.mv_x, mvy = j.mv_y;
  long long min_mcost = j.start_hp ? (long long)j.min_mcost : 0x7fffffffffffLL;
#pragma unroll 1
  for (int stage = 0; stage < 2; stage++) {
    const int step = stage == 0 ? 2 : 1, start = stage == 0 ? j.start_hp : j.start_qp;
    const int lambda = stage == 0 ? j.lambda_h : j.lambda_q, metric = stage == 0 ? j.metric_h : j.metric_q;
    if (stage == 1 && !j.start_qp) min_mcost = 0x7fffffffffffLL;                   // me_fullsearch.c:252-253
    unsigned long long key = ~0ull;
    if (l < 9) {
      long long cost;
      if (l < start) cost = (l == 0) ? min_mcost : 0x7fffffffffffLL;               // position 0 keeps the carried-in cost
      else {
        int cx = mvx + c_sp9[l][0] * step, cy = mvy + c_sp9[l][1] * step;
        cost = (long long)lambda * (mvbits(cx - j.pred_x) + mvbits(cy - j.pred_y));
        cost += (long long)block_dist(ps, c, cur_pitch, j.bsx, j.bsy, cx + pxp, cy + pyp, metric, j.test8x8) << 5;
      }
      // strict '<' against the running minimum, first position wins: position 0 (when skipped) wins ties
      key = ((unsigned long long)cost << 4) | (unsigned)l;
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      unsigned long long o = __shfl_xor(key, off, 16);
      key = o < key ? o : key;
    }
    const int bl = (int)(key & 15);
    const long long bc = (long long)(key >> 4);
    if (bc < min_mcost) { min_mcost = bc; mvx += c_sp9[bl][0] * step; mvy += c_sp9[bl][1] * step; }
  }
  jmhip_me_best b; b.mv_x = (int16_t)mvx; b.mv_y = (int16_t)mvy;
  b.cost = min_mcost > 0x7fffffffLL ? 0x7fffffff : (int32_t)min_mcost;
  return b;
}

__global__ __launch_bounds__(64) void k_me_subpel(const jmhip_subpel_job *__restrict__ jobs, int n, jmhip_me_best *__restrict__ out,
                                                  PlaneSet ps, const uint8_t *__restrict__ cur, int cur_pitch)
{
  if (*ps.jerr) return;                                   // a job record failed k_check_subpel_jobs
  const int g = blockIdx.x * 4 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const bool live = g < n;
  jmhip_subpel_job j = jobs[live ? g : 0];
  jmhip_me_best b = subpel_group(j, l, ps, cur, cur_pitch);
  if (live && l == 0) out[g] = b;
}

// (blocktype-independent) geometry of the 41 partitions in ABI order: x, y, w, h in luma samples
__device__ __constant__ uint8_t c_part_geom[JMHIP_NPART][4] = {
  {0,0,16,16}, {0,0,16,8},{0,8,16,8}, {0,0,8,16},{8,0,8,16}, {0,0,8,8},{8,0,8,8},{0,8,8,8},{8,8,8,8},
  {0,0,8,4},{8,0,8,4},{0,4,8,4},{8,4,8,4},{0,8,8,4},{8,8,8,4},{0,12,8,4},{8,12,8,4},
  {0,0,4,8},{4,0,4,8},{8,0,4,8},{12,0,4,8},{0,8,4,8},{4,8,4,8},{8,8,4,8},{12,8,4,8},
  {0,0,4,4},{4,0,4,4},{8,0,4,4},{12,0,4,4},{0,4,4,4},{4,4,4,4},{8,4,4,4},{12,4,4,4},
  {0,8,4,4},{4,8,4,4},{8,8,4,4},{12,8,4,4},{0,12,4,4},{4,12,4,4},{8,12,4,4},{12,12,4,4}
};

// BlockMotionSearch glue (mv_search.c:960-981): the integer-pel winner of every searched partition of
// every window job goes straight into the sub-pel refinement, all on the device.
__global__ __launch_bounds__(64) void k_me_refine(const jmhip_me_job *__restrict__ jobs, int njobs, const jmhip_me_result *__restrict__ ires,
                                                  jmhip_refine_params prm, jmhip_me_result *__restrict__ out,
                                                  PlaneSet ps, const uint8_t *__restrict__ cur, int cur_pitch)
{
  if (*ps.jerr) return;                                   // a job record failed k_check_me_jobs
  const int g = blockIdx.x * 4 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const int ji = g / JMHIP_NPART, p = g - ji * JMHIP_NPART;
  const bool live = ji < njobs;
  const jmhip_me_job *job = jobs + (live ? ji : 0);
  const bool active = live && ((job->part_mask >> p) & 1);
  jmhip_subpel_job j;
  j.pos_x = (int16_t)(job->mb_x + c_part_geom[p][0]); j.pos_y = (int16_t)(job->mb_y + c_part_geom[p][1]);
  j.bsx = c_part_geom[p][2]; j.bsy = c_part_geom[p][3];
  j.pred_x = job->pred[p][0]; j.pred_y = job->pred[p][1];
  const jmhip_me_best ib = ires[live ? ji : 0].best[p];
  j.mv_x = ib.mv_x; j.mv_y = ib.mv_y;
  j.lambda_h = prm.lambda_h; j.lambda_q = prm.lambda_q; j.metric_h = prm.metric_h; j.metric_q = prm.metric_q;
  j.start_hp = prm.start_hp; j.start_qp = prm.start_qp;
  j.test8x8 = (int8_t)(prm.transform8x8_mode && p <= 8);      // mv_search.c:1630 (types 1-3) / :1770 (type 4)
  j.min_mcost = ib.cost;
  jmhip_me_best b = subpel_group(j, l, ps, cur, cur_pitch);
  if (active && l == 0) out[ji].best[p] = b;
}

static PlaneSet planes_of(jmhip_ctx *ctx, int slot)
{
  PlaneSet ps; ps.base = ctx->d_sub[slot]; ps.pitch = ctx->pitch; ps.plane_stride = (long)ctx->plane_stride; ps.W = ctx->W; ps.H = ctx->H; ps.jerr = ctx->d_me_declined + 4;
  return ps;
}

extern "C" int jmhip_me_subpel_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_subpel_job *d_jobs, int32_t n, jmhip_me_best *d_results)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!d_jobs || !d_results || n < 0 || slot < 0 || slot >= ctx->cfg.num_ref_slots) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_subpel_dev: bad argument");
  if (n == 0) return JMHIP_OK;
  jmhip_launch_check_subpel_jobs(ctx, d_jobs, n);
  jmhip_time_begin(ctx, 2);
  hipLaunchKernelGGL(k_me_subpel, dim3((n + 3) / 4), dim3(64), 0, ctx->stream, d_jobs, n, d_results, planes_of(ctx, slot), ctx->d_cur, ctx->cur_pitch);
  jmhip_time_end(ctx, 2);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

static int check_block(jmhip_ctx *ctx, int i, int px, int py, int bsx, int bsy)
{
  if (px < 0 || py < 0 || bsx < 4 || bsy < 4 || bsx > 16 || bsy > 16 || (bsx & 3) || (bsy & 3) || px + bsx > ctx->W || py + bsy > ctx->H)
    return jmhip_fail(ctx, JMHIP_EINVAL, "entry %d: block %dx%d at (%d,%d) invalid for a %dx%d picture", i, bsx, bsy, px, py, ctx->W, ctx->H);
  return JMHIP_OK;
}

extern "C" int jmhip_me_subpel(jmhip_ctx *ctx, int32_t slot, const jmhip_subpel_job *jobs, int32_t n, jmhip_me_best *results)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!jobs || !results || n < 0) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_subpel: bad argument");
  if (n == 0) return JMHIP_OK;
  int r;
  for (int i = 0; i < n; i++) if ((r = check_block(ctx, i, jobs[i].pos_x, jobs[i].pos_y, jobs[i].bsx, jobs[i].bsy))) return r;
  void *dj, *dr;
  if ((r = jmhip_scratch(ctx, 0, sizeof(jmhip_subpel_job) * (size_t)n, &dj))) return r;
  if ((r = jmhip_scratch(ctx, 1, sizeof(jmhip_me_best) * (size_t)n, &dr))) return r;
  HIPCHK(ctx, hipMemcpyAsync(dj, jobs, sizeof(jmhip_subpel_job) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  if ((r = jmhip_me_subpel_dev(ctx, slot, (const jmhip_subpel_job *)dj, n, (jmhip_me_best *)dr))) return r;
  HIPCHK(ctx, hipMemcpyAsync(results, dr, sizeof(jmhip_me_best) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_me_refine_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *d_jobs, int32_t njobs, const jmhip_me_result *d_int,
                                   const jmhip_refine_params *prm, jmhip_me_result *d_out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!d_jobs || !d_int || !prm || !d_out || njobs < 0 || slot < 0 || slot >= ctx->cfg.num_ref_slots) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_refine_dev: bad argument");
  if (njobs == 0) return JMHIP_OK;
  jmhip_launch_check_me_jobs(ctx, d_jobs, njobs);
  jmhip_time_begin(ctx, 2);
  if (ctx->refine_per_block) {                 // the per-block kernel (one 16-lane group per partition): A/B testing
    const long groups = (long)njobs * JMHIP_NPART;
    hipLaunchKernelGGL(k_me_refine, dim3((unsigned)((groups + 3) / 4)), dim3(64), 0, ctx->stream, d_jobs, njobs, d_int, *prm, d_out,
                       planes_of(ctx, slot), ctx->d_cur, ctx->cur_pitch);
  } else jmhip_launch_refine_mb(ctx, slot, d_jobs, njobs, d_int, prm, d_out);
  jmhip_time_end(ctx, 2);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_me_eval(jmhip_ctx *ctx, int32_t slot, const jmhip_cand *cands, int32_t n, int32_t *dist)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!cands || !dist || n < 0 || slot < 0 || slot >= ctx->cfg.num_ref_slots) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_eval: bad argument");
  if (n == 0) return JMHIP_OK;
  int r;
  for (int i = 0; i < n; i++) if ((r = check_block(ctx, i, cands[i].pos_x, cands[i].pos_y, cands[i].bsx, cands[i].bsy))) return r;
  void *dc, *dd;
  if ((r = jmhip_scratch(ctx, 0, sizeof(jmhip_cand) * (size_t)n, &dc))) return r;
  if ((r = jmhip_scratch(ctx, 1, sizeof(int32_t) * (size_t)n, &dd))) return r;
  HIPCHK(ctx, hipMemcpyAsync(dc, cands, sizeof(jmhip_cand) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_me_eval, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const jmhip_cand *)dc, n, (int32_t *)dd, planes_of(ctx, slot), ctx->d_cur, ctx->cur_pitch);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(dist, dd, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

static int launch_eval_pred(jmhip_ctx *ctx, const jmhip_pred_cand *d_cands, int n, int32_t *d_dist)
{
  SlotBases sb;
  for (int k = 0; k < 32; k++) sb.p[k] = k < ctx->cfg.num_ref_slots ? ctx->d_sub[k] : nullptr;
  hipLaunchKernelGGL(k_me_eval_pred, dim3((n + 3) / 4), dim3(64), 0, ctx->stream, d_cands, n, d_dist, sb, (int)ctx->cfg.num_ref_slots,
                     planes_of(ctx, 0), ctx->d_cur, ctx->cur_pitch);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

// d_cands are not validated (device memory): a slot outside the context yields dist = -1 for that candidate
extern "C" int jmhip_me_eval_pred_dev(jmhip_ctx *ctx, const jmhip_pred_cand *d_cands, int32_t n, int32_t *d_dist)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!d_cands || !d_dist || n < 0) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_eval_pred_dev: bad argument");
  if (n == 0) return JMHIP_OK;
  return launch_eval_pred(ctx, d_cands, n, d_dist);
}

extern "C" int jmhip_me_eval_pred(jmhip_ctx *ctx, const jmhip_pred_cand *cands, int32_t n, int32_t *dist)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!cands || !dist || n < 0) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_eval_pred: bad argument");
  if (n == 0) return JMHIP_OK;
  int r;
  for (int i = 0; i < n; i++) {
    const jmhip_pred_cand &c = cands[i];
    if ((r = check_block(ctx, i, c.pos_x, c.pos_y, c.bsx, c.bsy))) return r;
    const bool two = c.pred == JMHIP_PRED_AVG || c.pred == JMHIP_PRED_BI_WP;
    if (c.pred < 0 || c.pred > JMHIP_PRED_UNI || (c.metric != JMHIP_METRIC_SAD && c.metric != JMHIP_METRIC_SSE && c.metric != JMHIP_METRIC_SATD) ||
        c.shift < 0 || c.shift > 8 || c.slot[0] < 0 || c.slot[0] >= ctx->cfg.num_ref_slots || (two && (c.slot[1] < 0 || c.slot[1] >= ctx->cfg.num_ref_slots)) ||
        (c.metric == JMHIP_METRIC_SATD && c.test8x8 && ((c.bsx & 7) || (c.bsy & 7))))
      return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_eval_pred: candidate %d: bad prediction kind, metric, shift, slot or 8x8 sub-blocks on a block that is not a multiple of 8", i);
  }
  void *dc, *dd;
  if ((r = jmhip_scratch(ctx, 0, sizeof(jmhip_pred_cand) * (size_t)n, &dc))) return r;
  if ((r = jmhip_scratch(ctx, 1, sizeof(int32_t) * (size_t)n, &dd))) return r;
  HIPCHK(ctx, hipMemcpyAsync(dc, cands, sizeof(jmhip_pred_cand) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  if ((r = launch_eval_pred(ctx, (const jmhip_pred_cand *)dc, n, (int32_t *)dd))) return r;
  HIPCHK(ctx, hipMemcpyAsync(dist, dd, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}
