/*
 * oracle/jmo_me.c -- TEST INFRASTRUCTURE (parity oracle, see jmo.h).
 * CPU restatement of JM 19.0 integer/sub-pel motion estimation:
 *   lencod/src/mv_search.c, me_fullsearch.c, me_fullfast.c, me_distortion.c,
 *   lencod/inc/refbuf.h, lencod/inc/mv_search.h.
 * Written from the algorithm, not copied: plain arrays instead of JM's
 * Macroblock/MEBlock/VideoParameters object graph.
 */
#include <stdlib.h>
#include <string.h>
#include "jmo.h"

static inline int iabs_(int x) { return x < 0 ? -x : x; }
static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }
static inline int iclip3(int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); }

/* mvbits LUT, mv_search.c:366-374: mvbits[0]=1; for bits=3,5,7..: |d| in [2^((bits>>1)-1), 2^(bits>>1)) -> bits.
 * Closed form: 2*floor(log2|d|)+3. */
int jmo_mvbits(int d)
{
  int a = iabs_(d), l = 0;
  if (a == 0) return 1;
  while (a > 1) { a >>= 1; l++; }
  return 2 * l + 3;
}

/* spiral search order, mv_search.c:405-442 */
void jmo_spiral(int search_range, jmo_mv *sp)
{
  int k = 1, l, i;
  sp[0].x = sp[0].y = 0;
  for (l = 1; l <= imax_(1, search_range); l++) {
    for (i = -l + 1; i < l; i++) {
      sp[k].x = (int16_t)i;  sp[k++].y = (int16_t)-l;
      sp[k].x = (int16_t)i;  sp[k++].y = (int16_t)l;
    }
    for (i = -l; i <= l; i++) {
      sp[k].x = (int16_t)-l; sp[k++].y = (int16_t)i;
      sp[k].x = (int16_t)l;  sp[k++].y = (int16_t)i;
    }
  }
}

int jmo_spiral_index(int dx, int dy)
{
  int ax = iabs_(dx), ay = iabs_(dy), l = imax_(ax, ay), base;
  if (l == 0) return 0;
  base = (2 * l - 1) * (2 * l - 1);
  if (ay == l && ax < l) return base + 2 * (dx + l - 1) + (dy > 0);
  return base + 2 * (2 * l - 1) + 2 * (dy + l) + (dx > 0);
}

/* UMVLine4X, lencod/inc/refbuf.h:22-26: sub-plane chosen by the two low bits,
 * block ORIGIN clamped to [-PAD, size_pad] (size_*_pad: lencod/src/mbuffer.c:564-565). */
static inline const jmo_pel *umv_line4x(const jmo_refpic *r, int y, int x)
{
  int size_x_pad = r->width  + 2 * JMO_PAD_X - 1 - 16 - JMO_PAD_X;
  int size_y_pad = r->height + 2 * JMO_PAD_Y - 1 - 16 - JMO_PAD_Y;
  int yy = iclip3(-JMO_PAD_Y, size_y_pad, y >> 2);
  int xx = iclip3(-JMO_PAD_X, size_x_pad, x >> 2);
  return r->sub[y & 3][x & 3] + (long)yy * r->pitch + xx;
}

/* me_distortion.c:175-258.  Any butterfly order gives the same sum of absolute
 * transformed values; restated as H*D*H^T with the 4-point Hadamard. */
int jmo_hadamard_sad4x4(const int16_t d[16])
{
  int m[16], t[16], i, satd = 0;
  for (i = 0; i < 4; i++) {           /* rows */
    int a = d[4*i], b = d[4*i+1], c = d[4*i+2], e = d[4*i+3];
    int s0 = a + e, s1 = b + c, s2 = b - c, s3 = a - e;
    m[4*i] = s0 + s1; m[4*i+1] = s0 - s1; m[4*i+2] = s2 + s3; m[4*i+3] = s3 - s2;
  }
  for (i = 0; i < 4; i++) {           /* columns */
    int a = m[i], b = m[4+i], c = m[8+i], e = m[12+i];
    int s0 = a + e, s1 = b + c, s2 = b - c, s3 = a - e;
    t[i] = s0 + s1; t[4+i] = s0 - s1; t[8+i] = s2 + s3; t[12+i] = s3 - s2;
  }
  for (i = 0; i < 16; i++) satd += iabs_(t[i]);
  return (satd + 1) >> 1;
}

/* me_distortion.c:266-341 */
int jmo_hadamard_sad8x8(const int16_t d[64])
{
  int m[64], i, j, sad = 0;
  for (j = 0; j < 8; j++) {
    int v[8], w[8];
    for (i = 0; i < 4; i++) { v[i] = d[8*j+i] + d[8*j+i+4]; v[i+4] = d[8*j+i] - d[8*j+i+4]; }
    w[0] = v[0] + v[2]; w[1] = v[1] + v[3]; w[2] = v[0] - v[2]; w[3] = v[1] - v[3];
    w[4] = v[4] + v[6]; w[5] = v[5] + v[7]; w[6] = v[4] - v[6]; w[7] = v[5] - v[7];
    for (i = 0; i < 4; i++) { m[8*j+2*i] = w[2*i] + w[2*i+1]; m[8*j+2*i+1] = w[2*i] - w[2*i+1]; }
  }
  for (i = 0; i < 8; i++) {
    int v[8], w[8], k;
    for (k = 0; k < 4; k++) { v[k] = m[8*k+i] + m[8*(k+4)+i]; v[k+4] = m[8*k+i] - m[8*(k+4)+i]; }
    w[0] = v[0] + v[2]; w[1] = v[1] + v[3]; w[2] = v[0] - v[2]; w[3] = v[1] - v[3];
    w[4] = v[4] + v[6]; w[5] = v[5] + v[7]; w[6] = v[4] - v[6]; w[7] = v[5] - v[7];
    for (k = 0; k < 4; k++) { sad += iabs_(w[2*k] + w[2*k+1]) + iabs_(w[2*k] - w[2*k+1]); }
  }
  return (sad + 2) >> 2;
}

/* computeSAD, me_distortion.c:349-426 (ChromaMEEnable = 0).  The row-wise early
 * exit returns the threshold itself (dist_scale_f == min_mcost, mv_search.h:19-20). */
jmo_dist jmo_compute_sad(const jmo_refpic *ref, const jmo_pel *orig, int bsx, int bsy,
                         jmo_dist min_mcost, int cand_x, int cand_y)
{
  int mcost = 0, x, y;
  int imin_cost = (int)(min_mcost >> JMO_LAMBDA_BITS);      /* dist_down, ifunctions.h:321 */
  const jmo_pel *src = orig;
  const jmo_pel *rl = umv_line4x(ref, cand_y, cand_x);
  for (y = 0; y < bsy; y++) {
    for (x = 0; x < bsx; x++) mcost += iabs_((int)src[x] - (int)rl[x]);
    if (mcost > imin_cost) return min_mcost;
    src += bsx;
    rl += ref->pitch;
  }
  return ((jmo_dist)mcost) << JMO_LAMBDA_BITS;
}

/* computeSATD, me_distortion.c:745-825: per 4x4 (or 8x8 when test8x8) sub-block,
 * each sub-block fetched through its own UMVLine4X origin clamp. */
jmo_dist jmo_compute_satd(const jmo_refpic *ref, const jmo_pel *orig, int bsx, int bsy,
                          int test8x8, jmo_dist min_mcost, int cand_x, int cand_y)
{
  int imin_cost = (int)(min_mcost >> JMO_LAMBDA_BITS);
  int mcost = 0, bs = test8x8 ? 8 : 4, x, y, i, j;
  int16_t diff[64];
  for (y = 0; y < bsy; y += bs) {
    for (x = 0; x < bsx; x += bs) {
      const jmo_pel *rl = umv_line4x(ref, cand_y + (y << 2), cand_x + (x << 2));
      const jmo_pel *sl = orig + y * bsx + x;
      for (j = 0; j < bs; j++)
        for (i = 0; i < bs; i++)
          diff[j * bs + i] = (int16_t)((int)sl[j * bsx + i] - (int)rl[(long)j * ref->pitch + i]);
      mcost += test8x8 ? jmo_hadamard_sad8x8(diff) : jmo_hadamard_sad4x4(diff);
      if (mcost > imin_cost) return min_mcost;
    }
  }
  return ((jmo_dist)mcost) << JMO_LAMBDA_BITS;
}

/* The weighted / bi-predictive candidate distortions of lencod/src/me_distortion.c (luma only: ChromaMEEnable = 0):
 *   pred 0  computeBiPred{SAD,SATD,SSE}1  :525 / :943 / :1353   p = (r1 + r2 + 1) >> 1
 *   pred 1  computeBiPred{SAD,SATD,SSE}2  :624 / :1038 / :1438  p = clip1(((w1*r1 + w2*r2 + round) >> shift) + offset),
 *                                                              round = 2*wp_luma_round, shift = luma_log_weight_denom + 1
 *   pred 2  compute{SAD,SATD,SSE}WP       :434 / :833 / :1261   p = clip1(((w1*r1 + round) >> shift) + offset),
 *                                                              round = wp_luma_round, shift = luma_log_weight_denom
 *   pred 3  compute{SAD,SATD,SSE}         :349 / :745 / :1190   p = r1
 * metric 0 = SAD, 1 = SSE, 2 = SATD.  SAD / SSE: one UMVLine4X origin per reference and an early exit per sample row; SATD: one origin
 * per reference per 4x4 (8x8 when test8x8) sub-block and an early exit per sub-block.  The early exits return min_mcost itself. */
static inline int pred_sample(int pred, int r1, int r2, const jmo_wp *wp, int max_pel)
{
  int v;
  switch (pred) {
  case 0: return (r1 + r2 + 1) >> 1;
  case 1: v = ((wp->weight[0] * r1 + wp->weight[1] * r2 + wp->round) >> wp->shift) + wp->offset; break;
  case 2: v = ((wp->weight[0] * r1 + wp->round) >> wp->shift) + wp->offset; break;
  default: return r1;
  }
  return v < 0 ? 0 : (v > max_pel ? max_pel : v);
}

jmo_dist jmo_compute_pred_dist(const jmo_refpic *ref1, const jmo_refpic *ref2, const jmo_pel *orig, int bsx, int bsy, int test8x8,
                               int metric, int pred, const jmo_wp *wp, int max_pel, jmo_dist min_mcost,
                               int cand1_x, int cand1_y, int cand2_x, int cand2_y)
{
  const int imin_cost = (int)(min_mcost >> JMO_LAMBDA_BITS);
  const int two = pred < 2;
  int mcost = 0, x, y, i, j;
  if (metric != 2) {
    const jmo_pel *r1 = umv_line4x(ref1, cand1_y, cand1_x);
    const jmo_pel *r2 = two ? umv_line4x(ref2, cand2_y, cand2_x) : r1;
    const long p2 = two ? ref2->pitch : ref1->pitch;
    for (y = 0; y < bsy; y++) {
      for (x = 0; x < bsx; x++) {
        int d = (int)orig[y * bsx + x] - pred_sample(pred, r1[(long)y * ref1->pitch + x], r2[y * p2 + x], wp, max_pel);
        mcost += metric == 0 ? iabs_(d) : d * d;
      }
      if (mcost > imin_cost) return min_mcost;
    }
  } else {
    const int bs = test8x8 ? 8 : 4;
    int16_t diff[64];
    for (y = 0; y < bsy; y += bs)
      for (x = 0; x < bsx; x += bs) {
        const jmo_pel *r1 = umv_line4x(ref1, cand1_y + (y << 2), cand1_x + (x << 2));
        const jmo_pel *r2 = two ? umv_line4x(ref2, cand2_y + (y << 2), cand2_x + (x << 2)) : r1;
        const long p2 = two ? ref2->pitch : ref1->pitch;
        const jmo_pel *sl = orig + y * bsx + x;
        /* computeBiPredSATD2's 8x8 path reads the eighth source sample of a row without advancing the pointer (me_distortion.c:1167,
         * `*src_line` where the other seven have `*src_line++`), so every further row of the sub-block starts one source sample earlier:
         * row j is read from sl + j*(bsx - 1).  The reference's results -- and so the bitstream -- carry that; it is restated as is. */
        const int src_pitch = (pred == 1 && test8x8) ? bsx - 1 : bsx;
        for (j = 0; j < bs; j++)
          for (i = 0; i < bs; i++)
            diff[j * bs + i] = (int16_t)((int)sl[j * src_pitch + i] - pred_sample(pred, r1[(long)j * ref1->pitch + i], r2[j * p2 + i], wp, max_pel));
        mcost += test8x8 ? jmo_hadamard_sad8x8(diff) : jmo_hadamard_sad4x4(diff);
        if (mcost > imin_cost) return min_mcost;
      }
  }
  return ((jmo_dist)mcost) << JMO_LAMBDA_BITS;
}

/* mv_cost, lencod/inc/mv_search.h:100-112 (JCOST_CALC_SCALEUP) */
static inline jmo_dist mv_cost(int lambda, int cx, int cy, int px, int py)
{
  return (jmo_dist)lambda * (jmo_dist)(jmo_mvbits(cx - px) + jmo_mvbits(cy - py));
}

/* full_search_motion_estimation, me_fullsearch.c:39-103, rdopt on (check_for_00 = 0) */
jmo_dist jmo_full_search(const jmo_refpic *ref, const jmo_pel *orig, jmo_fs_job *job, long *sad_evals)
{
  int R = job->search_range;
  int max_pos = (2 * R + 1) * (2 * R + 1), pos, best_pos = 0;
  jmo_mv *sp = (jmo_mv *)malloc(sizeof(jmo_mv) * (size_t)imax_(9, max_pos));
  int pxp = job->pos_x << 2, pyp = job->pos_y << 2;        /* pos_x_padded, mv_search.c:685 */
  int cx = pxp + job->center.x, cy = pyp + job->center.y;   /* me_fullsearch.c:62-63 */
  int px = pxp + job->pred.x,   py = pyp + job->pred.y;     /* :64-65 */
  jmo_dist min_mcost = job->min_mcost, mcost;
  long evals = 0;
  jmo_spiral(R, sp);
  for (pos = 0; pos < max_pos; pos++) {
    int candx = cx + (sp[pos].x << 2), candy = cy + (sp[pos].y << 2);
    mcost = mv_cost(job->lambda_factor, candx, candy, px, py);
    if (mcost >= min_mcost) continue;                        /* :83 */
    mcost += jmo_compute_sad(ref, orig, job->bsx, job->bsy, min_mcost - mcost, candx, candy);
    evals++;
    if (mcost < min_mcost) { best_pos = pos; min_mcost = mcost; }
  }
  if (best_pos) {
    job->center.x = (int16_t)(job->center.x + (sp[best_pos].x << 2));
    job->center.y = (int16_t)(job->center.y + (sp[best_pos].y << 2));
  }
  free(sp);
  if (sad_evals) *sad_evals = evals;
  return min_mcost;
}

/* setup_fast_full_search me_fullfast.c:269-608 (plain SAD branch :492-556) and
 * update_full_search_large_blocks :195-260 */
void jmo_ffs_setup(const jmo_refpic *ref, const jmo_pel cur[256], int mb_x, int mb_y,
                   jmo_mv center, int R, uint32_t *bs)
{
  int max_pos = (2 * R + 1) * (2 * R + 1), pos, b, x, y;
  jmo_mv *sp = (jmo_mv *)malloc(sizeof(jmo_mv) * (size_t)imax_(9, max_pos));
  int offx = (mb_x << 2) + center.x, offy = (mb_y << 2) + center.y;   /* search_center_padded :329 */
#define BS(t, k) (bs + ((size_t)(t) * 16 + (k)) * (size_t)max_pos)
  jmo_spiral(R, sp);
  for (pos = 0; pos < max_pos; pos++) {
    const jmo_pel *rp = umv_line4x(ref, offy + (sp[pos].y << 2), offx + (sp[pos].x << 2));  /* MB-origin clamp :498 */
    uint32_t s[16];
    memset(s, 0, sizeof s);
    for (y = 0; y < 16; y++)
      for (x = 0; x < 16; x++)
        s[(y >> 2) * 4 + (x >> 2)] += (uint32_t)iabs_((int)rp[(long)y * ref->pitch + x] - (int)cur[y * 16 + x]);
    for (b = 0; b < 16; b++) BS(7, b)[pos] = s[b];
  }
  /* aggregation, index = 4x4 raster index of the partition's top-left block */
  for (pos = 0; pos < max_pos; pos++) {
    static const int k6[8] = {0, 1, 2, 3, 8, 9, 10, 11};
    static const int k5[8] = {0, 2, 4, 6, 8, 10, 12, 14};
    static const int k4[4] = {0, 2, 8, 10};
    int i;
    for (i = 0; i < 8; i++) BS(6, k6[i])[pos] = BS(7, k6[i])[pos] + BS(7, k6[i] + 4)[pos];  /* 4x8 */
    for (i = 0; i < 8; i++) BS(5, k5[i])[pos] = BS(7, k5[i])[pos] + BS(7, k5[i] + 1)[pos];  /* 8x4 */
    for (i = 0; i < 4; i++) BS(4, k4[i])[pos] = BS(6, k4[i])[pos] + BS(6, k4[i] + 1)[pos];  /* 8x8 */
    BS(3, 0)[pos] = BS(4, 0)[pos] + BS(4, 8)[pos];  BS(3, 2)[pos] = BS(4, 2)[pos] + BS(4, 10)[pos];  /* 8x16 */
    BS(2, 0)[pos] = BS(4, 0)[pos] + BS(4, 2)[pos];  BS(2, 8)[pos] = BS(4, 8)[pos] + BS(4, 10)[pos];  /* 16x8 */
    BS(1, 0)[pos] = BS(3, 0)[pos] + BS(3, 2)[pos];                                                    /* 16x16 */
  }
#undef BS
  free(sp);
}

/* fast_full_search_motion_estimation me_fullfast.c:618-689, rdopt on */
jmo_dist jmo_ffs_search(const uint32_t *bs, int max_pos_table, int blocktype, int block_index,
                        jmo_mv center, jmo_mv pred, int R, int lambda, int max_mvd_,
                        jmo_dist min_mcost, jmo_mv *best_mv)
{
  int max_pos = (2 * R + 1) * (2 * R + 1), pos, best_pos = 0;
  int max_mvd = max_mvd_ - 1;                                 /* :638 */
  const uint32_t *row = bs + ((size_t)blocktype * 16 + block_index) * (size_t)max_pos_table;
  jmo_mv *sp = (jmo_mv *)malloc(sizeof(jmo_mv) * (size_t)imax_(9, max_pos));
  jmo_spiral(R, sp);
  for (pos = 0; pos < max_pos; pos++) {
    jmo_dist mcost = ((jmo_dist)row[pos]) << JMO_LAMBDA_BITS;
    int cx = center.x + (sp[pos].x << 2), cy = center.y + (sp[pos].y << 2);
    int mvd = imax_(iabs_(cx - pred.x), iabs_(cy - pred.y));  /* GetMaxMVD mv_search.h:133 */
    if (mcost < min_mcost && mvd < max_mvd) {
      mcost += mv_cost(lambda, cx, cy, pred.x, pred.y);
      if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
    }
  }
  best_mv->x = (int16_t)(center.x + (sp[best_pos].x << 2));
  best_mv->y = (int16_t)(center.y + (sp[best_pos].y << 2));
  free(sp);
  return min_mcost;
}

/* sub_pel_motion_estimation me_fullsearch.c:186-289, rdopt on (check_position0 = 0).
 * spiral_hpel_search = 2*spiral, spiral_search = 1*spiral in quarter-pel units
 * (mv_search.c:414-440); search_pos2 = search_pos4 = 9 (mv_search.c:720-721). */
static jmo_dist subpel_dist(const jmo_refpic *ref, const jmo_pel *orig, const jmo_subpel_job *j,
                            int metric, jmo_dist thr, int cx, int cy)
{
  if (metric == 0) return jmo_compute_sad(ref, orig, j->bsx, j->bsy, thr, cx, cy);
  return jmo_compute_satd(ref, orig, j->bsx, j->bsy, j->test8x8, thr, cx, cy);
}

jmo_dist jmo_sub_pel_search(const jmo_refpic *ref, const jmo_pel *orig, jmo_subpel_job *job)
{
  jmo_mv sp[9];
  jmo_dist min_mcost = job->min_mcost, mcost;
  int pos, best_pos, pxp = job->pos_x << 2, pyp = job->pos_y << 2;
  int max_pos2 = 9;                                           /* imax(1, search_pos2) */
  jmo_spiral(1, sp);
  for (best_pos = 0, pos = job->start_hp; pos < max_pos2; pos++) {
    int cx = job->mv.x + (sp[pos].x << 1), cy = job->mv.y + (sp[pos].y << 1);
    mcost = mv_cost(job->lambda_h, cx, cy, job->pred.x, job->pred.y);
    if (mcost >= min_mcost) continue;
    mcost += subpel_dist(ref, orig, job, job->metric_h, min_mcost - mcost, cx + pxp, cy + pyp);
    if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
  }
  if (best_pos) { job->mv.x = (int16_t)(job->mv.x + (sp[best_pos].x << 1)); job->mv.y = (int16_t)(job->mv.y + (sp[best_pos].y << 1)); }
  if (!job->start_qp) min_mcost = JMO_DIST_MAX;               /* :252-253 */
  for (best_pos = 0, pos = job->start_qp; pos < 9; pos++) {
    int cx = job->mv.x + sp[pos].x, cy = job->mv.y + sp[pos].y;
    mcost = mv_cost(job->lambda_q, cx, cy, job->pred.x, job->pred.y);
    if (mcost >= min_mcost) continue;
    mcost += subpel_dist(ref, orig, job, job->metric_q, min_mcost - mcost, cx + pxp, cy + pyp);
    if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
  }
  if (best_pos) { job->mv.x = (int16_t)(job->mv.x + sp[best_pos].x); job->mv.y = (int16_t)(job->mv.y + sp[best_pos].y); }
  return min_mcost;
}
