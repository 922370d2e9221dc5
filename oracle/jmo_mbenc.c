/*
 * oracle/jmo_mbenc.c -- TEST INFRASTRUCTURE (parity oracle, see jmo.h).
 *
 * CPU restatement of the reference's RDO-off macroblock pipeline for P and I slices of frame pictures:
 *
 *   encode_one_macroblock_low            lencod/src/md_low.c:104-687
 *   PartitionMotionSearch / SubPartitionMotionSearch / BlockMotionSearch   lencod/src/mv_search.c:1564 / :1686 / :857
 *   get_neighbors :268, GetMotionVectorPredictorNormal lcommon/src/mv_prediction.c:194, CheckSearchRange mv_search.c:822
 *   full_search_motion_estimation / sub_pel_motion_estimation               lencod/src/me_fullsearch.c:39 / :186 (rdopt == 0 paths)
 *   FindSkipModeMotionVector mv_search.c:1333, GetSkipCostMB :1257
 *   list_prediction_cost lencod/src/mode_decision.c:275, submacroblock_mode_decision_low mode_decision_P8x8.c:681
 *   mode_decision_for_I4x4_MB rd_intra_jm.c:386 (mode_decision_for_I4x4_blocks_JM_Low rd_intra_jm_low.c:39), find_sad_16x16_JM
 *   luma_residual_coding macroblock.c:1182 (_16x16 :841, _8x8 :919, reset_block :806), set_coeff_and_recon_8x8_p_slice rdopt.c:1326
 *   intra_chroma_prediction + rdo_low_intra_chroma_decision intra_chroma.c:530 / :460, chroma_residual_coding macroblock.c:1439
 *   set_modes_and_refs_for_blocks_p_slice rdopt.c:958, SetMotionVectorsMBPSlice md_common.c:172, the skip test md_low.c:658
 *
 * Scope (everything else is rejected by the caller): frame macroblocks, 4:2:0, 8 bit, 4x4 transform only, CAVLC level clamp, no adaptive rounding,
 * no weighted prediction, no rate control, SearchMode = -1 (full search), unconstrained intra prediction, default quantiser offsets.
 * The intermediate contents of enc_picture->mv_info inside the current macroblock are reproduced write by write, because the
 * predictors of later partitions read them (set_me_parameters mv_search.c:100, assign_enc_picture_params rdopt.c:3580,
 * set_ref_and_motion_vectors_P_slice rdopt.c:2885).
 *
 * Pinned by tests/golden/mb_low_*.npz: per-macroblock dumps of the real encoder (oracle/ref_tap_mb.c).
 */
#include <stdlib.h>
#include <string.h>
#include "jmo.h"

static inline int iabs_(int x) { return x < 0 ? -x : x; }
static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }
static inline int iclip3(int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); }
static inline int imedian(int a, int b, int c) { return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b)); }

typedef struct { jmo_mv mv; int8_t ref; } mvinfo;

typedef struct {
  const jmo_mbenc_cfg *c;
  int wmb, hmb, w4;
  const jmo_pel *cur[3];          /* source planes at the coded size (pitch = width / width/2) */
  const jmo_refpic *refl[2];      /* per list: [num_ref] luma quarter-pel planes (list 1: B slices) */
  const jmo_pel *const *refcl[2]; /* per list: [num_ref][2] integer chroma planes (pitch width/2) */
  int cl;                         /* the list the current search works on (mv_block->list): 0 outside B slices */
  jmo_pel *rec[3];                /* reconstruction (before the loop filter), written macroblock by macroblock */
  mvinfo *mil[2];                 /* per list: [h4][w4] enc_picture->mv_info (.mv[list], .ref_idx[list]) */
  int8_t *ipm;                    /* [h4][w4] p_Vid->ipredmode */
  jmo_mv *spiral; int spiral_R;
  /* per macroblock */
  int mbx, mby, addr;
  int availA, availB, availC, availD;    /* left, up, up-right, up-left macroblock inside the slice */
  jmo_mv all_mvl[2][JMO_MAX_REF][8][4][4];   /* currSlice->all_mv[list][ref][mode][by][bx] */
  jmo_dist motion_costl[8][2][JMO_MAX_REF][4];   /* p_Vid->motion_cost[mode][list][ref][block] */
  const struct jmo_b_cfg_s *bc;          /* B slices: list 1, direct mode, bi-predictive search (jmo_mbenc_b.inc) */
  jmo_mv bipred_mv[2][2][8][4][4];       /* currSlice->bipred_mv[set][list][0][mode][by][bx] (reference 0 only) */
  int8_t direct_ref[16][2], direct_pdir[16];   /* currSlice->direct_ref_idx / direct_pdir of the macroblock, 4x4 raster */
  jmo_pel orig[256];
  struct epzs_state *ez;                 /* SearchMode = 3: the slice's EPZS state (jmo_mbenc_epzs.inc) */
} enc;

static inline jmo_dist mv_cost(int lambda, int cx, int cy, int px, int py) { return (jmo_dist)lambda * (jmo_dist)(jmo_mvbits(cx - px) + jmo_mvbits(cy - py)); }

/* ---- neighbours: get4x4Neighbour for luma positions relative to the current macroblock (non-MBAFF, lencod/src/mb_access.c) ---- */
typedef struct { int avail, x4, y4; } nb;
static nb neighbour4(const enc *e, int x, int y)      /* sample position relative to the macroblock */
{
  nb n = {0, 0, 0};
  int ok;
  if (x < 0) ok = y < 0 ? e->availD : (y < 16 ? e->availA : 0);
  else if (x < 16) ok = y < 0 ? e->availB : (y < 16 ? 1 : 0);
  else ok = y < 0 ? e->availC : 0;
  if (ok) { n.avail = 1; n.x4 = (e->mbx * 16 + x) >> 2; n.y4 = (e->mby * 16 + y) >> 2; }
  return n;
}

/* get_neighbors mv_search.c:268-307 */
static void get_neighbors(const enc *e, nb b[4], int mb_x, int mb_y, int bsx)
{
  b[0] = neighbour4(e, mb_x - 1, mb_y);
  b[1] = neighbour4(e, mb_x, mb_y - 1);
  b[2] = neighbour4(e, mb_x + bsx, mb_y - 1);
  b[3] = neighbour4(e, mb_x - 1, mb_y - 1);
  if (mb_y > 0) {
    if (mb_x < 8) {
      if (mb_y == 8) { if (bsx == 16) b[2].avail = 0; }
      else if (mb_x + bsx == 8) b[2].avail = 0;
    } else if (mb_x + bsx == 16) b[2].avail = 0;
  }
  if (!b[2].avail) b[2] = b[3];
}

/* GetMotionVectorPredictorNormal lcommon/src/mv_prediction.c:194-325 */
static jmo_mv mv_predictor(const enc *e, const nb b[4], int ref, int mb_x, int mb_y, int bsx, int bsy)
{
  const mvinfo *m[3];
  int r[3], k, type = 0;        /* 0 median, 1 L, 2 U, 3 UR */
  jmo_mv z = {0, 0}, p;
  for (k = 0; k < 3; k++) { m[k] = b[k].avail ? &e->mil[e->cl][b[k].y4 * e->w4 + b[k].x4] : NULL; r[k] = m[k] ? m[k]->ref : -1; }
  if (r[0] == ref && r[1] != ref && r[2] != ref) type = 1;
  else if (r[0] != ref && r[1] == ref && r[2] != ref) type = 2;
  else if (r[0] != ref && r[1] != ref && r[2] == ref) type = 3;
  if (bsx == 8 && bsy == 16) {
    if (mb_x == 0) { if (r[0] == ref) type = 1; }
    else { if (r[2] == ref) type = 3; }
  } else if (bsx == 16 && bsy == 8) {
    if (mb_y == 0) { if (r[1] == ref) type = 2; }
    else { if (r[0] == ref) type = 1; }
  }
  if (type == 0) {
    if (!(b[1].avail || b[2].avail)) return m[0] ? m[0]->mv : z;
    {
      jmo_mv a = m[0] ? m[0]->mv : z, bb = m[1] ? m[1]->mv : z, cc = m[2] ? m[2]->mv : z;
      p.x = (int16_t)imedian(a.x, bb.x, cc.x); p.y = (int16_t)imedian(a.y, bb.y, cc.y);
      return p;
    }
  }
  return m[type - 1] ? m[type - 1]->mv : z;
}

/* FindSkipModeMotionVector mv_search.c:1333-1405 */
static jmo_mv skip_mv(const enc *e)
{
  nb b[4];
  jmo_mv z = {0, 0};
  int zl, za;
  get_neighbors(e, b, 0, 0, 16);
  zl = !b[0].avail ? 1 : (e->mil[e->cl][b[0].y4 * e->w4 + b[0].x4].ref == 0 && e->mil[e->cl][b[0].y4 * e->w4 + b[0].x4].mv.x == 0 && e->mil[e->cl][b[0].y4 * e->w4 + b[0].x4].mv.y == 0);
  za = !b[1].avail ? 1 : (e->mil[e->cl][b[1].y4 * e->w4 + b[1].x4].ref == 0 && e->mil[e->cl][b[1].y4 * e->w4 + b[1].x4].mv.x == 0 && e->mil[e->cl][b[1].y4 * e->w4 + b[1].x4].mv.y == 0);
  if (za || zl) return z;
  return mv_predictor(e, b, 0, 0, 0, 16, 16);
}

static void get_orig_block(const enc *e, int bx, int by, int bsx, int bsy, jmo_pel *o)
{
  int j, i;
  for (j = 0; j < bsy; j++) for (i = 0; i < bsx; i++) o[j * bsx + i] = e->orig[(by + j) * 16 + bx + i];
}

/* the same with 8x8 Hadamard blocks (p_Vid->distortion8x8 = distortion8x8SATD) */
static jmo_dist satd_blocks8(const jmo_pel *orig, int opitch, const jmo_pel *pred, int ppitch, int w, int h)
{
  jmo_dist c = 0;
  int by, bx, j, i;
  int16_t d[64];
  for (by = 0; by < h; by += 8)
    for (bx = 0; bx < w; bx += 8) {
      for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) d[j * 8 + i] = (int16_t)((int)orig[(by + j) * opitch + bx + i] - (int)pred[(by + j) * ppitch + bx + i]);
      c += ((jmo_dist)jmo_hadamard_sad8x8(d)) << JMO_LAMBDA_BITS;
    }
  return c;
}

/* SATD of the macroblock against a 16x16 prediction, sixteen 4x4 blocks (p_Vid->distortion4x4 = distortion4x4SATD, MDDistortion = 2) */
static jmo_dist satd_blocks(const jmo_pel *orig, int opitch, const jmo_pel *pred, int ppitch, int w, int h)
{
  jmo_dist c = 0;
  int by, bx, j, i;
  int16_t d[16];
  for (by = 0; by < h; by += 4)
    for (bx = 0; bx < w; bx += 4) {
      for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) d[j * 4 + i] = (int16_t)((int)orig[(by + j) * opitch + bx + i] - (int)pred[(by + j) * ppitch + bx + i]);
      c += ((jmo_dist)jmo_hadamard_sad4x4(d)) << JMO_LAMBDA_BITS;
    }
  return c;
}

#include "jmo_mbenc_epzs.inc"

/* BlockMotionSearch mv_search.c:857-1024 with SearchMode = 3 (EPZS, EPZSSubPelGrid = 1): the centre is the predictor itself (:924-927), the
 * integer search EPZS_integer_motion_estimation or its sub-macroblock variant (mv_search.c:1634-1640, :1772-1788), the sub-pel search
 * EPZS_sub_pel_motion_estimation unless a later reference already looks hopeless (:965) */
static jmo_dist block_motion_search_epzs(enc *e, int ref, int blocktype, int mb_x, int mb_y, int bsx, int bsy, jmo_mv *out)
{
  const jmo_mbenc_cfg *c = e->c;
  epzs_blk B;
  nb b[4];
  jmo_pel orig[256];
  jmo_mv pred, mv, center;
  int min_x = -(c->search_range << 2), max_x = c->search_range << 2, min_y = min_x, max_y = max_x;
  jmo_dist min_mcost;
  get_neighbors(e, b, mb_x, mb_y, bsx);
  get_orig_block(e, mb_x, mb_y, bsx, bsy, orig);
  pred = mv_predictor(e, b, ref, mb_x, mb_y, bsx, bsy);
  mv = pred;
  center = mv;
  mv.x = (int16_t)iclip3(min_x, max_x, mv.x);
  mv.y = (int16_t)iclip3(min_y, max_y, mv.y);
  if (mv.x != center.x || mv.y != center.y) {            /* CheckSearchRange :822-849 */
    const int lim = c->max_mvd - 2;
    int left = mv.x + min_x, right = mv.x + max_x, top = mv.y + min_y, down = mv.y + max_y;
    left = iclip3(center.x - lim, center.x + lim, left);   right = iclip3(center.x - lim, center.x + lim, right);
    top = iclip3(center.y - lim, center.y + lim, top);     down = iclip3(center.y - lim, center.y + lim, down);
    if (left < right && top < down) {
      mv.x = (int16_t)((left + right) >> 1); mv.y = (int16_t)((top + down) >> 1);
      max_x = imin_(mv.x - left, right - mv.x); max_y = imin_(mv.y - top, down - mv.y);
    } else mv = center;
  }
  mv.x = (int16_t)iclip3(c->mv_limit[0], c->mv_limit[1], mv.x);
  mv.y = (int16_t)iclip3(c->mv_limit[2], c->mv_limit[3], mv.y);
  memset(&B, 0, sizeof B);
  B.e = e; B.s = (epzs *)e->ez; B.rp = &e->refl[e->cl][ref]; B.orig = orig; B.bt = blocktype; B.bsx = bsx; B.bsy = bsy; B.ref = ref; B.mb_x = mb_x; B.mb_y = mb_y;
  B.pxp = (e->mbx * 16 + mb_x) << 2; B.pyp = (e->mby * 16 + mb_y) << 2; B.x4 = (e->mbx * 16 + mb_x) >> 2; B.y4 = (e->mby * 16 + mb_y) >> 2;
  B.lambda = c->lambda_mf[0]; B.pred = pred; B.mv = mv; B.max_x = max_x; B.max_y = max_y; B.b = b;
  min_mcost = epzs_integer(&B, blocktype > 4);
  if (c->subpel) {
    const jmo_dist prev = *epzs_prevsad(&B);
    if (ref == 0 || 2 * min_mcost < 7 * prev) {             /* min_mcost < 3.5 * prevSad[pic_pix_x >> 2] */
      min_mcost = epzs_subpel(&B, c->lambda_mf, c->start_qp, JMO_DIST_MAX);     /* start_me_refinement_hp = 0: the stage starts from DISTBLK_MAX (:969-972) */
    }
  }
  mv = B.mv;
  mv.x = (int16_t)iclip3(c->mv_limit[0], c->mv_limit[1], mv.x);
  mv.y = (int16_t)iclip3(c->mv_limit[2], c->mv_limit[3], mv.y);
  if (blocktype == 1 && c->slice_type == 0) {
    jmo_mv s = skip_mv(e), z = {0, 0};
    jmo_pel pr[256];
    jmo_dist cost;
    int j, i;
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) e->all_mvl[e->cl][0][0][j][i] = s;
    jmo_luma_pred(&e->refl[e->cl][0], NULL, 0, e->mbx * 16, e->mby * 16, 16, 16, s, z, pr);
    cost = (c->transform8x8 ? satd_blocks8(e->orig, 16, pr, 16, 16, 16) : satd_blocks(e->orig, 16, pr, 16, 16, 16)) - (jmo_dist)c->lambda_mf[2] * 8;
    if (cost < min_mcost) { min_mcost = cost; mv = s; }
  }
  *out = mv;
  return min_mcost;
}

/* BlockMotionSearch mv_search.c:857-1024 for one (block, reference), rdopt == 0, SearchMode = -1.  Returns min_mcost; *out = the vector. */
static jmo_dist block_motion_search(enc *e, int ref, int blocktype, int mb_x, int mb_y, int bsx, int bsy, jmo_mv *out)
{
  const jmo_mbenc_cfg *c = e->c;
  const jmo_refpic *rp = &e->refl[e->cl][ref];
  const int pos_x = e->mbx * 16 + mb_x, pos_y = e->mby * 16 + mb_y, pxp = pos_x << 2, pyp = pos_y << 2;
  nb b[4];
  jmo_mv pred, mv, center;
  jmo_pel orig[256];
  int min_x = -(c->search_range << 2), max_x = c->search_range << 2, min_y = min_x, max_y = max_x;    /* p_Vid->searchRange lencod.c:649-652; full_search == 2 */
  jmo_dist min_mcost = JMO_DIST_MAX, mcost;
  int R, max_pos, pos, best_pos = 0, cx, cy, px, py, check00, check_position0;

  if (c->search_mode == 3) return block_motion_search_epzs(e, ref, blocktype, mb_x, mb_y, bsx, bsy, out);
  get_neighbors(e, b, mb_x, mb_y, bsx);
  get_orig_block(e, mb_x, mb_y, bsx, bsy, orig);
  pred = mv_predictor(e, b, ref, mb_x, mb_y, bsx, bsy);
  mv.x = (int16_t)(((pred.x + 2) >> 2) * 4);             /* JM_INT_DIVIDE, mv_search.c:931-932 */
  mv.y = (int16_t)(((pred.y + 2) >> 2) * 4);
  center = mv;
  mv.x = (int16_t)iclip3(min_x, max_x, mv.x);            /* :949-950: the (0,0) vector stays inside the window */
  mv.y = (int16_t)iclip3(min_y, max_y, mv.y);
  if (mv.x != center.x || mv.y != center.y) {            /* CheckSearchRange :822-849 */
    const int lim = c->max_mvd - 2;
    int left = mv.x + min_x, right = mv.x + max_x, top = mv.y + min_y, down = mv.y + max_y;
    left = iclip3(center.x - lim, center.x + lim, left);   right = iclip3(center.x - lim, center.x + lim, right);
    top = iclip3(center.y - lim, center.y + lim, top);     down = iclip3(center.y - lim, center.y + lim, down);
    if (left < right && top < down) {
      mv.x = (int16_t)((left + right) >> 1); mv.y = (int16_t)((top + down) >> 1);
      min_x = left - mv.x; max_x = imin_(mv.x - left, right - mv.x);
      min_y = top - mv.y;  max_y = imin_(mv.y - top, down - mv.y);
    } else mv = center;
  }
  mv.x = (int16_t)iclip3(c->mv_limit[0], c->mv_limit[1], mv.x);     /* clip_mv_range(.., 0, mv, Q_PEL) :957 */
  mv.y = (int16_t)iclip3(c->mv_limit[2], c->mv_limit[3], mv.y);
  (void)min_x; (void)min_y;

  if (c->search_mode == 1) {
    /* ---- fast_full_search_motion_estimation me_fullfast.c:618-689 (rdopt == 0): every block of the macroblock is searched around ONE centre per reference,
     * the rounded 16x16 predictor (setup_fast_full_search :310-328), the (0,0) vector first (:650-657), the max_mvd guard (:638, :671).
     * SCOPE: the centre stays on the sample grid with (0,0) inside the window, i.e. mv_limit[2] + 4 R <= -4 R and mv_limit[3] - 4 R >= 4 R (the same horizontally).  With the limits
     * of levels 1 / 1b and R = 32 the second clip (:326-327) can leave the centre at 127; JM then finds no (0,0) position (:354-365) and prices the (0,0) vector with the pos_00
     * an earlier macroblock left behind -- raster-order state this restatement (and the device) does not carry: jmo_encode_slice refuses such a configuration. ---- */
    nb b16[4];
    jmo_mv p16, ctr;
    const int rq = c->search_range << 2, guard = c->max_mvd - 1;
    int pos00 = -1;
    get_neighbors(e, b16, 0, 0, 16);
    p16 = mv_predictor(e, b16, ref, 0, 0, 16, 16);
    ctr.x = (int16_t)iclip3(-rq, rq, ((p16.x + 2) >> 2) * 4);
    ctr.y = (int16_t)iclip3(-rq, rq, ((p16.y + 2) >> 2) * 4);
    ctr.x = (int16_t)iclip3(c->mv_limit[0] + rq, c->mv_limit[1] - rq, ctr.x);
    ctr.y = (int16_t)iclip3(c->mv_limit[2] + rq, c->mv_limit[3] - rq, ctr.y);
    R = imax_(max_x, max_y) >> 2;                              /* :633 (imax) */
    max_pos = (2 * R + 1) * (2 * R + 1);
    if (e->spiral_R < c->search_range) { free(e->spiral); e->spiral = (jmo_mv *)malloc(sizeof(jmo_mv) * (size_t)imax_(9, (2 * c->search_range + 1) * (2 * c->search_range + 1))); jmo_spiral(c->search_range, e->spiral); e->spiral_R = c->search_range; }
    cx = pxp + ctr.x; cy = pyp + ctr.y;
    if (imax_(iabs_(0 - pred.x), iabs_(0 - pred.y)) < guard) {
      min_mcost = jmo_compute_sad(rp, orig, bsx, bsy, JMO_DIST_MAX, pxp, pyp) + mv_cost(c->lambda_mf[0], 0, 0, pred.x, pred.y);
      pos00 = 0;                                               /* marks "the (0,0) vector holds the minimum" */
    }
    best_pos = -1;
    for (pos = 0; pos < max_pos; pos++) {
      const int vx = ctr.x + (e->spiral[pos].x << 2), vy = ctr.y + (e->spiral[pos].y << 2);
      if (imax_(iabs_(vx - pred.x), iabs_(vy - pred.y)) >= guard) continue;
      mcost = jmo_compute_sad(rp, orig, bsx, bsy, JMO_DIST_MAX, pxp + vx, pyp + vy);
      if (mcost >= min_mcost) continue;
      mcost += mv_cost(c->lambda_mf[0], vx, vy, pred.x, pred.y);
      if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
    }
    if (best_pos >= 0) { mv.x = (int16_t)(ctr.x + (e->spiral[best_pos].x << 2)); mv.y = (int16_t)(ctr.y + (e->spiral[best_pos].y << 2)); }
    else if (pos00 == 0) { mv.x = 0; mv.y = 0; }
    else { mv = ctr; }                                         /* best_pos = 0 without any candidate: the centre */
    (void)cx; (void)cy;
    best_pos = 0;
  } else {
  /* ---- full_search_motion_estimation me_fullsearch.c:39-103 ---- */
  R = imin_(max_x, max_y) >> 2;
  max_pos = (2 * R + 1) * (2 * R + 1);
  if (e->spiral_R < R) { free(e->spiral); e->spiral = (jmo_mv *)malloc(sizeof(jmo_mv) * (size_t)imax_(9, max_pos)); jmo_spiral(R, e->spiral); e->spiral_R = R; }
  check00 = (blocktype == 1 && c->slice_type != 1 && ref == 0);
  cx = pxp + mv.x; cy = pyp + mv.y; px = pxp + pred.x; py = pyp + pred.y;
  for (pos = 0; pos < max_pos; pos++) {
    const int candx = cx + (e->spiral[pos].x << 2), candy = cy + (e->spiral[pos].y << 2);
    mcost = mv_cost(c->lambda_mf[0], candx, candy, px, py);
    if (check00 && candx == pxp && candy == pyp) {
      const jmo_dist t = (jmo_dist)c->lambda_mf[0] * 16;
      mcost = mcost > t ? mcost - t : 0;
    }
    if (mcost >= min_mcost) continue;
    mcost += jmo_compute_sad(rp, orig, bsx, bsy, min_mcost - mcost, candx, candy);
    if (mcost < min_mcost) { best_pos = pos; min_mcost = mcost; }
  }
  if (best_pos) { mv.x = (int16_t)(mv.x + (e->spiral[best_pos].x << 2)); mv.y = (int16_t)(mv.y + (e->spiral[best_pos].y << 2)); }
  }

  /* ---- sub_pel_motion_estimation me_fullsearch.c:186-289; start_me_refinement_hp = 0 (SAD then SATD), start_me_refinement_qp = cfg (1 when the
   * half- and quarter-pel metrics are the same, mv_search.c:445-446: the quarter-pel stage then keeps the half-pel minimum and skips position 0) ---- */
  if (c->subpel) {
    jmo_mv sp[9];
    const int test8x8 = c->transform8x8 && blocktype <= 4;      /* mv_block.test8x8: mv_search.c:1624 (Transform8x8Mode), :1768 (... && blocktype == 4) */
    jmo_spiral(1, sp);
    check_position0 = (c->slice_type != 1 && ref == 0 && blocktype == 1 && mv.x == 0 && mv.y == 0);
    min_mcost = JMO_DIST_MAX;                                 /* mv_search.c:971-974 */
    for (best_pos = 0, pos = 0; pos < 9; pos++) {
      const int qx = mv.x + (sp[pos].x << 1), qy = mv.y + (sp[pos].y << 1);
      mcost = mv_cost(c->lambda_mf[1], qx, qy, pred.x, pred.y);
      if (mcost >= min_mcost) continue;
      mcost += jmo_compute_satd(rp, orig, bsx, bsy, test8x8, min_mcost - mcost, qx + pxp, qy + pyp);
      if (pos == 0 && check_position0) mcost -= (jmo_dist)c->lambda_mf[1] * 16;
      if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
    }
    if (best_pos) { mv.x = (int16_t)(mv.x + (sp[best_pos].x << 1)); mv.y = (int16_t)(mv.y + (sp[best_pos].y << 1)); }
    if (!c->start_qp) min_mcost = JMO_DIST_MAX;                 /* me_fullsearch.c:252-253 */
    for (best_pos = 0, pos = c->start_qp; pos < 9; pos++) {
      const int qx = mv.x + sp[pos].x, qy = mv.y + sp[pos].y;
      mcost = mv_cost(c->lambda_mf[2], qx, qy, pred.x, pred.y);
      if (mcost >= min_mcost) continue;
      mcost += jmo_compute_satd(rp, orig, bsx, bsy, test8x8, min_mcost - mcost, qx + pxp, qy + pyp);
      if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
    }
    if (best_pos) { mv.x = (int16_t)(mv.x + sp[best_pos].x); mv.y = (int16_t)(mv.y + sp[best_pos].y); }
  }
  mv.x = (int16_t)iclip3(c->mv_limit[0], c->mv_limit[1], mv.x);     /* :981 */
  mv.y = (int16_t)iclip3(c->mv_limit[2], c->mv_limit[3], mv.y);

  /* ---- the skip vector's cost against the 16x16 result (rdopt == 0) mv_search.c:983-998, GetSkipCostMB :1257-1325 ---- */
  if (blocktype == 1 && c->slice_type == 0) {
    jmo_mv s = skip_mv(e), z = {0, 0};
    jmo_pel pr[256];
    jmo_dist cost;
    int j, i;
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) e->all_mvl[e->cl][0][0][j][i] = s;
    jmo_luma_pred(&e->refl[e->cl][0], NULL, 0, e->mbx * 16, e->mby * 16, 16, 16, s, z, pr);
    cost = (c->transform8x8 ? satd_blocks8(e->orig, 16, pr, 16, 16, 16) : satd_blocks(e->orig, 16, pr, 16, 16, 16)) - (jmo_dist)c->lambda_mf[2] * 8;
    if (cost < min_mcost) { min_mcost = cost; mv = s; }
  }
  *out = mv;
  return min_mcost;
}

static void set_me_parameters(enc *e, jmo_mv mv, int ref, int x4, int y4, int w4, int h4)   /* mv_search.c:100-113 */
{
  int j, i;
  for (j = y4; j < y4 + h4; j++) for (i = x4; i < x4 + w4; i++) { mvinfo *m = &e->mil[e->cl][(e->mby * 4 + j) * e->w4 + e->mbx * 4 + i]; m->mv = mv; m->ref = (int8_t)ref; }
}

static const int PART_W[8] = {16, 16, 16, 8, 8, 8, 4, 4}, PART_H[8] = {16, 16, 8, 16, 8, 4, 8, 4};

/* list_prediction_cost mode_decision.c:275 (LIST_0, no SP, checkref 0) with update_mcost :253 */
static jmo_dist list0_cost(const enc *e, int mode, int block, int *best_ref)
{
  const int ref_lambda = e->c->lambda_mf[2] >> 2;            /* rdopt == 0 */
  jmo_dist bm = JMO_DIST_MAX;
  int ref;
  for (ref = 0; ref < e->c->num_ref; ref++) {
    jmo_dist mc = e->motion_costl[mode][e->cl][ref][block];
    if (mc < bm) {
      mc += e->c->num_ref <= 1 ? 0 : (jmo_dist)ref_lambda * e->c->refbits[ref];
      if (mc < bm) { bm = mc; *best_ref = ref; }
    }
  }
  return bm;
}

/* one 4x4 luma block through residual_transform_quant_luma_4x4; levels into the dense scan-order array */
static int tq4x4(const enc *e, const jmo_pel *orig, int opitch, const jmo_pel *pred, int ppitch, int intra, int16_t dense[16], int *coeff_cost,
                 jmo_pel *rec, int rpitch)
{
  jmo_pel o[16], p[16], r[16];
  int level[17], run[17], fadj[16], j, i, nz, pos, k;
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) { o[j * 4 + i] = orig[j * opitch + i]; p[j * 4 + i] = pred[j * ppitch + i]; }
  level[0] = 0;
  {                                                            /* residual_transform_quant_luma_4x4 block.c:661-725 with the slice's quantiser offsets */
    int res[16], tb[16], rr[16], any = 0;
    jmo_qparam q[16];
    (void)fadj;
    for (k = 0; k < 16; k++) { res[k] = (int)o[k] - (int)p[k]; any |= res[k]; }
    nz = 0;
    if (any) {
      jmo_qparams_4x4_m(e->c->qp, e->c->off4[0][intra ? 1 : 0], q);
      jmo_forward4x4(res, tb);
      nz = jmo_quant_4x4_normal(tb, q, e->c->qp / 6, !e->c->cabac, &JMO_SNGL_SCAN[0][0], JMO_COEFF_COST4x4[0], level, run, coeff_cost);
    }
    if (nz) {
      jmo_inverse4x4(tb, rr);
      for (k = 0; k < 16; k++) { int v = ((rr[k] + 32) >> 6) + (int)p[k]; r[k] = (jmo_pel)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    } else
      for (k = 0; k < 16; k++) r[k] = p[k];
  }
  memset(dense, 0, 16 * sizeof(int16_t));
  for (pos = 0, k = 0; k < 16 && level[k] != 0; k++) { pos += run[k]; dense[pos++] = (int16_t)level[k]; }
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) rec[j * rpitch + i] = r[j * 4 + i];
  return nz;
}

/* one 8x8 luma block through residual_transform_quant_luma_8x8 / _cavlc (transform8x8.c:522 / :604); the 64 levels into lev[4][16] in the frame zig-zag
 * order of the 8x8 scan: lev[s >> 4][s & 15] */
static int tq8x8(const enc *e, const jmo_pel *orig, int opitch, const jmo_pel *pred, int ppitch, int intra, int16_t lev[4][16], int *coeff_cost,
                 jmo_pel *rec, int rpitch)
{
  jmo_pel o[64], p[64], r[64];
  jmo_qparam q[64];
  int level[68], run[68], fadj[64], j, i, nz, k, l;
  const int cavlc = !e->c->cabac;
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) { o[j * 8 + i] = orig[j * opitch + i]; p[j * 8 + i] = pred[j * ppitch + i]; }
  jmo_qparams_8x8_m(e->c->qp, e->c->off8[intra ? 1 : 0], q);
  level[0] = level[17] = level[34] = level[51] = 0;
  nz = jmo_rtq_luma_8x8(o, p, q, e->c->qp / 6, cavlc, 0, 0, 255, level, run, coeff_cost, r, fadj, NULL);
  memset(lev, 0, 64 * sizeof(int16_t));
  if (cavlc) {
    for (l = 0; l < 4; l++) {
      int pos = 0;
      for (k = 0; k < 16 && level[17 * l + k] != 0; k++) { const int sidx = 4 * (pos + run[17 * l + k]) + l; pos += run[17 * l + k] + 1; lev[sidx >> 4][sidx & 15] = (int16_t)level[17 * l + k]; }
    }
  } else {
    int pos = 0;
    for (k = 0; k < 64 && level[k] != 0; k++) { pos += run[k]; lev[pos >> 4][pos & 15] = (int16_t)level[k]; pos++; }
  }
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) rec[j * rpitch + i] = r[j * 8 + i];
  return nz;
}

typedef struct {
  jmo_pel pred[256], rec[256];
  int16_t lev[16][16];            /* block index 4*b8 + b4 */
  int cbp; int64_t cbp_blk; int cnt_nonz;
} luma_result;

/* luma_residual_coding_8x8 macroblock.c:919-1018 for one 8x8 block, 4x4 transform, P slice */
static int luma_rc_8x8(enc *e, luma_result *L, int b8, int mode, int ref, int t8)
{
  const int mb_y = (b8 >> 1) << 3, mb_x = (b8 & 1) << 3;
  jmo_mv z = {0, 0};
  int coeff_cost = 0, by, bx, j, i;
  if (t8) {                                                    /* luma_transform_size_8x8_flag: one 8x8 prediction, one 8x8 transform (:991-1011) */
    jmo_pel p[64];
    jmo_luma_pred(&e->refl[e->cl][ref], NULL, 0, e->mbx * 16 + mb_x, e->mby * 16 + mb_y, 8, 8, e->all_mvl[e->cl][ref][mode][mb_y >> 2][mb_x >> 2], z, p);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) L->pred[(mb_y + j) * 16 + mb_x + i] = p[j * 8 + i];
    if (tq8x8(e, e->orig + mb_y * 16 + mb_x, 16, L->pred + mb_y * 16 + mb_x, 16, 0, &L->lev[b8 * 4], &coeff_cost, L->rec + mb_y * 16 + mb_x, 16)) {
      L->cbp_blk |= (int64_t)51 << (4 * b8 - 2 * (b8 & 1));
      L->cbp |= 1 << b8;
    }
  } else {
  if (mode < 5) {
    jmo_pel p[64];
    jmo_luma_pred(&e->refl[e->cl][ref], NULL, 0, e->mbx * 16 + mb_x, e->mby * 16 + mb_y, 8, 8, e->all_mvl[e->cl][ref][mode][mb_y >> 2][mb_x >> 2], z, p);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) L->pred[(mb_y + j) * 16 + mb_x + i] = p[j * 8 + i];
  }
  for (by = mb_y; by < mb_y + 8; by += 4)
    for (bx = mb_x; bx < mb_x + 8; bx += 4) {
      const int b4 = ((by >> 2) & 1) * 2 + ((bx >> 2) & 1);
      if (mode >= 5) {
        jmo_pel p[16];
        jmo_luma_pred(&e->refl[e->cl][ref], NULL, 0, e->mbx * 16 + bx, e->mby * 16 + by, 4, 4, e->all_mvl[e->cl][ref][mode][by >> 2][bx >> 2], z, p);
        for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) L->pred[(by + j) * 16 + bx + i] = p[j * 4 + i];
      }
      if (tq4x4(e, e->orig + by * 16 + bx, 16, L->pred + by * 16 + bx, 16, 0, L->lev[b8 * 4 + b4], &coeff_cost, L->rec + by * 16 + bx, 16)) {
        L->cbp_blk |= (int64_t)1 << ((bx >> 2) + by);
        L->cbp |= 1 << b8;
      }
    }
  }
  if (coeff_cost <= 4) {                                       /* _LUMA_COEFF_COST_, reset_block macroblock.c:806-829 */
    L->cbp &= 63 - (1 << b8);
    L->cbp_blk &= ~((int64_t)51 << (4 * b8 - 2 * (b8 & 1)));
    for (j = 0; j < 4; j++) memset(L->lev[b8 * 4 + j], 0, 16 * sizeof(int16_t));
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) L->rec[(mb_y + j) * 16 + mb_x + i] = L->pred[(mb_y + j) * 16 + mb_x + i];
    coeff_cost = 0;
  }
  return coeff_cost;
}

/* luma_residual_coding macroblock.c:1182-1257 for mb_type 1, 2, 3 */
static void luma_rc_mb(enc *e, luma_result *L, int mode, const int ref8[4], int t8)
{
  int sum = 0, b8, j, i;
  jmo_mv z = {0, 0};
  memset(L, 0, sizeof *L);
  if (mode == 1) {
    jmo_luma_pred(&e->refl[e->cl][ref8[0]], NULL, 0, e->mbx * 16, e->mby * 16, 16, 16, e->all_mvl[e->cl][ref8[0]][1][0][0], z, L->pred);
    for (b8 = 0; b8 < 4; b8++) {                               /* luma_residual_coding_16x16 :841-908 */
      const int mb_y = (b8 >> 1) << 3, mb_x = (b8 & 1) << 3;
      int coeff_cost = 0, by, bx;
      if (t8) {
        if (tq8x8(e, e->orig + mb_y * 16 + mb_x, 16, L->pred + mb_y * 16 + mb_x, 16, 0, &L->lev[b8 * 4], &coeff_cost, L->rec + mb_y * 16 + mb_x, 16)) {
          L->cbp_blk |= (int64_t)51 << (4 * b8 - 2 * (b8 & 1));
          L->cbp |= 1 << b8;
        }
      } else
      for (by = mb_y; by < mb_y + 8; by += 4)
        for (bx = mb_x; bx < mb_x + 8; bx += 4) {
          const int b4 = ((by >> 2) & 1) * 2 + ((bx >> 2) & 1);
          if (tq4x4(e, e->orig + by * 16 + bx, 16, L->pred + by * 16 + bx, 16, 0, L->lev[b8 * 4 + b4], &coeff_cost, L->rec + by * 16 + bx, 16)) {
            L->cbp_blk |= (int64_t)1 << ((bx >> 2) + by);
            L->cbp |= 1 << b8;
          }
        }
      if (coeff_cost <= 4) {
        L->cbp &= 63 - (1 << b8);
        L->cbp_blk &= ~((int64_t)51 << (4 * b8 - 2 * (b8 & 1)));
        for (j = 0; j < 4; j++) memset(L->lev[b8 * 4 + j], 0, 16 * sizeof(int16_t));
        for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) L->rec[(mb_y + j) * 16 + mb_x + i] = L->pred[(mb_y + j) * 16 + mb_x + i];
        coeff_cost = 0;
      }
      sum += coeff_cost;
    }
  } else
    for (b8 = 0; b8 < 4; b8++) sum += luma_rc_8x8(e, L, b8, mode, ref8[b8], t8);
  if (sum <= 5) {                                              /* _LUMA_MB_COEFF_COST_ :1248-1255 (the coefficient lists stay; cbp hides them) */
    L->cbp &= 0xfffff0; L->cbp_blk &= 0xff0000;
    memcpy(L->rec, L->pred, sizeof L->rec);
  }
}

/* predictor samples of set_intrapred_4x4 intra4x4.c:421-519 from the reconstruction */
static void intra4_neighbours(const enc *e, int bx, int by, jmo_pel ee[13], int *left, int *up, int *all)
{
  const int W = e->c->width, X = e->mbx * 16 + bx, Y = e->mby * 16 + by;
  nb a = neighbour4(e, bx - 1, by), b = neighbour4(e, bx, by - 1), c = neighbour4(e, bx + 4, by - 1), d = neighbour4(e, bx - 1, by - 1);
  const jmo_pel *r = e->rec[0];
  int k;
  if (c.avail && bx == 4 && (by == 4 || by == 12)) c.avail = 0;
  *left = a.avail; *up = b.avail; *all = b.avail && a.avail && d.avail;
  for (k = 0; k < 4; k++) ee[1 + k] = b.avail ? r[(Y - 1) * W + X + k] : 128;
  for (k = 0; k < 4; k++) ee[5 + k] = c.avail ? r[(Y - 1) * W + X + 4 + k] : ee[4];
  for (k = 0; k < 4; k++) ee[9 + k] = a.avail ? r[(Y + k) * W + X - 1] : 128;
  ee[0] = d.avail ? r[(Y - 1) * W + X - 1] : 128;
}

/* mode_decision_for_I4x4_MB rd_intra_jm.c:386 -> Mode_Decision_for_IntraSubMBlocks :357 -> mode_decision_for_I4x4_blocks_JM_Low.
 * Writes the reconstruction and ipredmode of the macroblock as it goes (later blocks predict from them). */
static jmo_dist intra4x4_mb(enc *e, jmo_mb_record *o, int *cbp_out)
{
  const int lambda = e->c->lambda_mdfp, W = e->c->width;
  jmo_dist cost = 0;
  int cbp = 0, b8, b4;
  for (b8 = 0; b8 < 4; b8++) {
    jmo_dist cost8 = (jmo_dist)lambda * 6;
    for (b4 = 0; b4 < 4; b4++) {
      const int bx = ((b8 & 1) << 3) + ((b4 & 1) << 2), by = ((b8 >> 1) << 3) + ((b4 >> 1) << 2);
      const int X = e->mbx * 16 + bx, Y = e->mby * 16 + by;
      nb lb = neighbour4(e, bx - 1, by), tb = neighbour4(e, bx, by - 1);
      const int upm = tb.avail ? e->ipm[tb.y4 * e->w4 + tb.x4] : -1, lm = lb.avail ? e->ipm[lb.y4 * e->w4 + lb.x4] : -1;
      const int mpm = (upm < 0 || lm < 0) ? 2 : (upm < lm ? upm : lm);
      jmo_pel ee[13], pr[16], best_pr[16], ob[16];
      jmo_dist min_cost = JMO_DIST_MAX, c;
      int left, up, all, m, best = 0, j, i, dummy = 0;
      intra4_neighbours(e, bx, by, ee, &left, &up, &all);
      for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) ob[j * 4 + i] = e->orig[(by + j) * 16 + bx + i];
      memset(best_pr, 0, sizeof best_pr);
      for (m = 0; m < 9; m++) {
        const int ok = all || m == 2 || (up && (m == 0 || m == 7 || m == 3)) || (left && (m == 1 || m == 8));   /* VERT 0, HOR 1, DC 2, DDL 3, .., VL 7, HU 8 */
        if (!ok) continue;
        jmo_intrapred_4x4(ee, m, left, up, pr);
        c = (jmo_dist)lambda * (m == mpm ? 1 : 4);
        if (c < min_cost) {
          int16_t d[16];
          for (j = 0; j < 16; j++) d[j] = (int16_t)((int)ob[j] - (int)pr[j]);
          c += ((jmo_dist)jmo_hadamard_sad4x4(d)) << JMO_LAMBDA_BITS;              /* compute_satd4x4_cost rdopt.c:3819 */
          if (c < min_cost) { best = m; min_cost = c; memcpy(best_pr, pr, sizeof pr); }
        }
      }
      e->ipm[(Y >> 2) * e->w4 + (X >> 2)] = (int8_t)best;
      o->ipredmode[(by >> 2) * 4 + (bx >> 2)] = (int8_t)best;
      o->ipred_syntax[4 * b8 + b4] = (int8_t)(mpm == best ? -1 : (best < mpm ? best : best - 1));
      if (tq4x4(e, ob, 4, best_pr, 4, 1, o->luma[4 * b8 + b4], &dummy, e->rec[0] + Y * W + X, W)) cbp |= 1 << b8;
      cost8 += min_cost;
    }
    cost += cost8;
  }
  *cbp_out = cbp;
  return cost;
}

static void intra16_neighbours(const enc *e, jmo_pel ee[33], int *left, int *up, int *upleft)
{
  const int W = e->c->width, X = e->mbx * 16, Y = e->mby * 16;
  const jmo_pel *r = e->rec[0];
  int k;
  *left = e->availA; *up = e->availB; *upleft = e->availD;
  for (k = 0; k < 16; k++) ee[1 + k] = e->availB ? r[(Y - 1) * W + X + k] : 128;
  for (k = 0; k < 16; k++) ee[17 + k] = e->availA ? r[(Y + k) * W + X - 1] : 128;
  ee[0] = e->availD ? r[(Y - 1) * W + X - 1] : 128;
}

/* chroma: intra_chroma_prediction intra_chroma.c:530 + rdo_low_intra_chroma_decision :460, then chroma_residual_coding macroblock.c:1439.
 * 4:2:0: 8 x 8 samples per plane and macroblock; 4:2:2 (yuv_format 2): 8 x 16 -- eight 4x4 blocks, the vector of the luma block at the same row,
 * the 2x4 DC transform with the quantiser of qpc + 3 (block.c:1056-1093). */
static void chroma_mb(enc *e, jmo_mb_record *o, int intra, int mode, const int8_t b8mode[4], const int ref8[4])
{
  const jmo_mbenc_cfg *c = e->c;
  const int y422 = c->yuv_format == 2, RH = y422 ? 16 : 8, nblk = y422 ? 8 : 4;
  const int CW = c->width >> 1, CH = y422 ? c->height : c->height >> 1, cx = e->mbx * 8, cy = e->mby * RH;
  jmo_pel ipred[2][4][128];
  int uv, m, j, i, cr_cbp = 0, mask = 0;
  int64_t cbp_blk = 0;
  jmo_qparam q_ac[16], q_dcs[16], q_dc;
  {                                                              /* the decision runs for every macroblock; only intra ones keep it */
    jmo_dist min_cost = JMO_DIST_MAX;
    int best = 0;
    for (uv = 0; uv < 2; uv++) {
      jmo_pel up[8], left[16];
      const jmo_pel *r = e->rec[1 + uv];
      for (i = 0; i < 8; i++) up[i] = e->availB ? r[(cy - 1) * CW + cx + i] : 0;
      for (i = 0; i < RH; i++) left[i] = e->availA ? r[(cy + i) * CW + cx - 1] : 0;
      mask = jmo_intra_chroma_pred(up, left, e->availD ? r[(cy - 1) * CW + cx - 1] : 0, e->availB, e->availA, e->availD, RH, 255, ipred[uv]);
    }
    for (m = 0; m < 4; m++) {
      jmo_dist cost = 0;
      if ((m == 2 && !e->availB) || (m == 1 && !e->availA) || (m == 3 && (!e->availA || !e->availB || !e->availD))) continue;
      (void)mask;
      for (uv = 0; uv < 2; uv++) cost += satd_blocks(e->cur[1 + uv] + cy * CW + cx, CW, ipred[uv][m], 8, 8, RH);
      if (cost < min_cost) { best = m; min_cost = cost; }
    }
    o->c_ipred_mode = (int8_t)best;
  }
  for (uv = 0; uv < 2; uv++) {
    jmo_pel pred[128], orig[128], rec[128];
    int dc_level[9], dc_run[9], ac_level[8][16], ac_run[8][16], fadj[128], b4, pos, k;
    const int qpc = c->qpc + uv * c->qpc_cr_delta;
    jmo_qparams_4x4_m(qpc, c->off4[1 + uv][intra ? 1 : 0], q_ac);
    q_dc = q_ac[0];
    if (y422) { jmo_qparams_4x4_m(qpc + 3, c->off4[1 + uv][intra ? 1 : 0], q_dcs); q_dc = q_dcs[0]; }
    if (intra) memcpy(pred, ipred[uv][o->c_ipred_mode], sizeof pred);
    else
      for (j = 0; j < RH; j += 4)
        for (i = 0; i < 8; i += 4) {
          const int b8 = (y422 ? j >> 3 : j >> 2) * 2 + (i >> 2), md = mode == 8 ? b8mode[b8] : mode, ref = ref8[b8];
          jmo_mv mv0[4][2], mv1[4][2];
          jmo_pel out[16];
          int jj, h;
          memset(mv1, 0, sizeof mv1);
          for (jj = 0; jj < 4; jj++) for (h = 0; h < 2; h++) mv0[jj][h] = e->all_mvl[e->cl][ref][md][y422 ? (j + jj) >> 2 : (j + jj) >> 1][(i + 2 * h) >> 1];
          jmo_chroma_pred4x4(e->refcl[e->cl][ref * 2 + uv], NULL, CW, CW, CH, y422 ? 2 : 1, 0, cx + i, cy + j, mv0, mv1, out);
          for (jj = 0; jj < 4; jj++) for (h = 0; h < 4; h++) pred[(j + jj) * 8 + i + h] = out[jj * 4 + h];
        }
    for (j = 0; j < RH; j++) for (i = 0; i < 8; i++) orig[j * 8 + i] = e->cur[1 + uv][(cy + j) * CW + cx + i];
    for (k = 0; k < 8; k++) ac_level[k][0] = 0;
    dc_level[0] = 0;
    cr_cbp = jmo_rtq_chroma(y422 ? 2 : 1, uv, cr_cbp, &cbp_blk, q_ac, &q_dc, qpc / 6, y422 ? (qpc + 3) / 6 : qpc / 6, !c->cabac, 0, 0, 255, orig, pred, rec,
                            dc_level, dc_run, ac_level, ac_run, fadj);
    for (pos = 0, k = 0; k < nblk && dc_level[k] != 0; k++) { pos += dc_run[k]; o->chroma_dc[uv][pos++] = (int16_t)dc_level[k]; }
    for (b4 = 0; b4 < nblk; b4++)
      for (pos = 1, k = 0; k < 15 && ac_level[b4][k] != 0; k++) { pos += ac_run[b4][k]; o->chroma_ac[uv][b4][pos++] = (int16_t)ac_level[b4][k]; }
    for (j = 0; j < RH; j++) for (i = 0; i < 8; i++) e->rec[1 + uv][(cy + j) * CW + cx + i] = rec[j * 8 + i];
  }
  o->cbp = (int16_t)(o->cbp + (cr_cbp << 4));
  o->cbp_blk |= (uint64_t)cbp_blk;
}

/* the 4x4-Hadamard and the 8x8-Hadamard SATD of a 16x16 prediction, 8x8 block by 8x8 block (transform_decision macroblock.c:1347, get_best_transform_8x8 md_low.c:43) */
static void satd_4_and_8(const enc *e, const jmo_pel *pred4, const jmo_pel *pred8, jmo_dist *c4, jmo_dist *c8)
{
  *c4 = satd_blocks(e->orig, 16, pred4, 16, 16, 16);
  *c8 = satd_blocks8(e->orig, 16, pred8, 16, 16, 16);
}

/* transform_decision (macroblock.c:1347-1425, block_check = -1) for mode 1..3: the prediction is made 8x8 block by 8x8 block */
static int transform_decision(enc *e, int mode, const int ref8[4], jmo_dist *cost)
{
  jmo_pel pr[256], p[64];
  jmo_mv z = {0, 0};
  jmo_dist c4, c8;
  int b8, j, i;
  for (b8 = 0; b8 < 4; b8++) {
    const int mb_y = (b8 >> 1) << 3, mb_x = (b8 & 1) << 3;
    jmo_luma_pred(&e->refl[e->cl][ref8[b8]], NULL, 0, e->mbx * 16 + mb_x, e->mby * 16 + mb_y, 8, 8, e->all_mvl[e->cl][ref8[b8]][mode][mb_y >> 2][mb_x >> 2], z, p);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) pr[(mb_y + j) * 16 + mb_x + i] = p[j * 8 + i];
  }
  satd_4_and_8(e, pr, pr, &c4, &c8);
  if (c8 < c4) return 1;
  *cost += c4 - c8;
  return 0;
}

/* set_intrapred_8x8 intra8x8.c:497-601 with LowPassForIntra8x8Pred :85-140: the 25 predictor samples Z, A..P, Q..X of 8x8 block (bx, by) */
static void intra8_neighbours(const enc *e, int bx, int by, jmo_pel pp[25], int *left, int *up, int *all)
{
  const int W = e->c->width, X = e->mbx * 16 + bx, Y = e->mby * 16 + by;
  nb a = neighbour4(e, bx - 1, by), b = neighbour4(e, bx, by - 1), c = neighbour4(e, bx + 8, by - 1), d = neighbour4(e, bx - 1, by - 1);
  const jmo_pel *r = e->rec[0];
  jmo_pel P[25], L[25];
  int k;
  if (bx == 8 && by == 8) c.avail = 0;
  *left = a.avail; *up = b.avail; *all = b.avail && a.avail && d.avail;
  for (k = 0; k < 8; k++) P[1 + k] = b.avail ? r[(Y - 1) * W + X + k] : 128;
  for (k = 0; k < 8; k++) P[9 + k] = c.avail ? r[(Y - 1) * W + X + 8 + k] : P[8];
  for (k = 0; k < 8; k++) P[17 + k] = a.avail ? r[(Y + k) * W + X - 1] : 128;
  P[0] = d.avail ? r[(Y - 1) * W + X - 1] : 128;
  memcpy(L, P, sizeof P);
  if (b.avail) {
    L[1] = (jmo_pel)(((d.avail ? P[0] : P[1]) + 2 * P[1] + P[2] + 2) >> 2);
    for (k = 2; k < 16; k++) L[k] = (jmo_pel)((P[k - 1] + 2 * P[k] + P[k + 1] + 2) >> 2);
    L[16] = (jmo_pel)((P[15] + 3 * P[16] + 2) >> 2);
  }
  if (d.avail) {
    if (b.avail && a.avail) L[0] = (jmo_pel)((2 * P[0] + P[1] + P[17] + 2) >> 2);
    else if (b.avail) L[0] = (jmo_pel)((3 * P[0] + P[1] + 2) >> 2);
    else if (a.avail) L[0] = (jmo_pel)((3 * P[0] + P[17] + 2) >> 2);
  }
  if (a.avail) {
    L[17] = (jmo_pel)(((d.avail ? P[0] : P[17]) + 2 * P[17] + P[18] + 2) >> 2);
    for (k = 18; k < 24; k++) L[k] = (jmo_pel)((P[k - 1] + 2 * P[k] + P[k + 1] + 2) >> 2);
    L[24] = (jmo_pel)((P[23] + 3 * P[24] + 2) >> 2);
  }
  memcpy(pp, L, sizeof L);
}

/* mode_decision_for_I8x8_MB transform8x8.c:241 -> mode_decision_for_I8x8_blocks_JM_Low rd_intra_jm_low.c:162.  Writes the macroblock's reconstruction as it goes
 * (later blocks predict from it); ipm8 = p_Vid->ipredmode8x8 of the macroblock (4x4 raster). */
static jmo_dist intra8x8_mb(enc *e, jmo_mb_record *o, int8_t ipm8[16], int *cbp_out)
{
  const int lambda = e->c->lambda_mdfp, W = e->c->width;
  jmo_dist cost = (jmo_dist)lambda * 6;
  int cbp = 0, b8;
  for (b8 = 0; b8 < 4; b8++) {
    const int bx = (b8 & 1) << 3, by = (b8 >> 1) << 3, X = e->mbx * 16 + bx, Y = e->mby * 16 + by;
    nb lb = neighbour4(e, bx - 1, by), tb = neighbour4(e, bx, by - 1);
    int upm, lm, mpm, left, up, all, m, best = 0, j, i, k, dummy = 0;
    jmo_pel pp[25], pr[64], best_pr[64], ob[64];
    jmo_dist min_cost = JMO_DIST_MAX, c;
    int16_t d[64];
    if (b8 >> 1) upm = tb.avail ? ipm8[((by >> 2) - 1) * 4 + (bx >> 2)] : -1;
    else upm = tb.avail ? e->ipm[tb.y4 * e->w4 + tb.x4] : -1;
    if (b8 & 1) lm = lb.avail ? ipm8[(by >> 2) * 4 + (bx >> 2) - 1] : -1;
    else lm = lb.avail ? e->ipm[lb.y4 * e->w4 + lb.x4] : -1;
    mpm = (upm < 0 || lm < 0) ? 2 : (upm < lm ? upm : lm);
    intra8_neighbours(e, bx, by, pp, &left, &up, &all);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) ob[j * 8 + i] = e->orig[(by + j) * 16 + bx + i];
    memset(best_pr, 0, sizeof best_pr);
    for (k = -1; k < 9; k++) {                                 /* the most probable mode first, then the others in ascending order */
      m = k < 0 ? mpm : k;
      if (k >= 0 && m == mpm) continue;
      if (!(m == 2 || ((m == 0 || m == 7 || m == 3) && up) || ((m == 1 || m == 8) && left) || all)) continue;
      jmo_intrapred_8x8(pp, m, left, up, pr);
      c = (jmo_dist)lambda * (k < 0 ? 1 : 4);
      if (k >= 0 && !(c < min_cost)) continue;
      for (j = 0; j < 64; j++) d[j] = (int16_t)((int)ob[j] - (int)pr[j]);
      c += ((jmo_dist)jmo_hadamard_sad8x8(d)) << JMO_LAMBDA_BITS;                  /* compute_satd8x8_cost transform8x8.c:880 */
      if (k < 0 || c < min_cost) { best = m; min_cost = c; memcpy(best_pr, pr, sizeof pr); }
    }
    for (j = 0; j < 2; j++) for (i = 0; i < 2; i++) { ipm8[((by >> 2) + j) * 4 + (bx >> 2) + i] = (int8_t)best; o->ipredmode[((by >> 2) + j) * 4 + (bx >> 2) + i] = (int8_t)best; }
    o->ipred_syntax[4 * b8] = (int8_t)(mpm == best ? -1 : (best < mpm ? best : best - 1));
    if (tq8x8(e, ob, 8, best_pr, 8, 1, &o->luma[4 * b8], &dummy, e->rec[0] + Y * W + X, W)) cbp |= 1 << b8;
    cost += min_cost;
  }
  *cbp_out = cbp;
  return cost;
}

static void store_luma(enc *e, const jmo_pel *src)
{
  const int W = e->c->width;
  int j, i;
  for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) e->rec[0][(e->mby * 16 + j) * W + e->mbx * 16 + i] = src[j * 16 + i];
}

#include "jmo_mbenc_b.inc"

static void encode_mb(enc *e, jmo_mb_record *o, jmo_mb_debug *dbg)
{
  const jmo_mbenc_cfg *c = e->c;
  const int W = c->width, first = c->first_mb;
  const int pslice = c->slice_type == 0;
  jmo_dist min_cost = JMO_DIST_MAX, min_rdcost, rd_cost;
  int best_mode = 10, mode, block, ref, j, i, k;
  int best_ref[8][4];                      /* b8x8info->best[mode][b8].ref[LIST_0] */
  int8_t p8mode[4] = {0, 0, 0, 0}, p8tmode[4] = {0, 0, 0, 0};
  int p8ref[4] = {0, 0, 0, 0}, p8tref[4] = {0, 0, 0, 0};
  jmo_mv p8tmv[16];
  jmo_dist p8cost = 0, p8tcost = 0;
  luma_result P8, P8T, L;
  int p8_valid = 0, p8_t8 = 0;
  int cur_t8 = 0, best_transform_flag = 0;      /* currMB->luma_transform_size_8x8_flag as md_low.c carries it, and the flag of the best of modes 1..3 */
  int8_t ipm8[16];
  jmo_pel i8rec[256];
  int tmp8 = 0;

  memset(o, 0, sizeof *o);
  e->availA = e->mbx > 0 && e->addr - 1 >= first;
  e->availB = e->mby > 0 && e->addr - e->wmb >= first;
  e->availC = e->mby > 0 && e->mbx < e->wmb - 1 && e->addr - e->wmb + 1 >= first;
  e->availD = e->mby > 0 && e->mbx > 0 && e->addr - e->wmb - 1 >= first;
  for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) e->orig[j * 16 + i] = e->cur[0][(e->mby * 16 + j) * W + e->mbx * 16 + i];
  { jmo_mv z = {0, 0}; set_me_parameters(e, z, -1, 0, 0, 4, 4); }            /* reset_macroblock macroblock.c:259-281 */
  memset(best_ref, 0, sizeof best_ref);
  memset(e->motion_costl, 0, sizeof e->motion_costl);
  memset(e->all_mvl[0], 0, sizeof e->all_mvl[0]);

  if (pslice) {
    /* ---- 16x16, 16x8, 8x16: md_low.c:185-263 ---- */
    for (mode = 1; mode < 4; mode++) {
      jmo_dist cost = 0;
      if (!c->inter_valid[mode]) continue;
      for (block = 0; block < (mode == 1 ? 1 : 2); block++) {
        const int bx = mode == 3 ? 8 * block : 0, by = mode == 2 ? 8 * block : 0, bw = PART_W[mode], bh = PART_H[mode];
        int bref = 0;
        for (ref = 0; ref < c->num_ref; ref++) {                                /* PartitionMotionSearch mv_search.c:1625-1660 */
          jmo_mv mv;
          e->motion_costl[mode][e->cl][ref][block] = block_motion_search(e, ref, mode, bx, by, bw, bh, &mv);
          for (j = 0; j < bh / 4; j++) for (i = 0; i < bw / 4; i++) e->all_mvl[e->cl][ref][mode][by / 4 + j][bx / 4 + i] = mv;
          set_me_parameters(e, mv, ref, bx / 4, by / 4, bw / 4, bh / 4);
        }
        cost += list0_cost(e, mode, block, &bref);
        for (j = 0; j < bh / 4; j++)                                             /* assign_enc_picture_params rdopt.c:3580: vectors, not ref_idx */
          for (i = 0; i < bw / 4; i++) e->mil[e->cl][(e->mby * 4 + by / 4 + j) * e->w4 + e->mbx * 4 + bx / 4 + i].mv = e->all_mvl[e->cl][bref][mode][by / 4 + j][bx / 4 + i];
        if (mode == 1) for (k = 0; k < 4; k++) best_ref[1][k] = bref;          /* set_block8x8_info rdopt.c:3680 */
        else if (mode == 2) { best_ref[2][2 * block] = bref; best_ref[2][2 * block + 1] = bref; }
        else { best_ref[3][block] = bref; best_ref[3][block + 2] = bref; }
        if (mode > 1 && block == 0) set_me_parameters(e, e->all_mvl[e->cl][bref][mode][0][0], bref, 0, 0, bw / 4, bh / 4);   /* set_ref_and_motion_vectors_P_slice */
      }
      cur_t8 = 0;
      if (c->transform8x8) cur_t8 = transform_decision(e, mode, best_ref[mode], &cost);      /* md_low.c:244-249 */
      if (cost < min_cost) { best_mode = mode; min_cost = cost; best_transform_flag = cur_t8; }
    }
    /* ---- P8x8: md_low.c:265-354, submacroblock_mode_decision_low mode_decision_P8x8.c:681.  With Transform8x8Mode = 1 the four blocks are first decided with
     * 8x8 partitions only and the 8x8 transform (tr8x8), then with all sub-modes and the 4x4 transform (tr4x4); both passes search again ---- */
    if (c->inter_valid[4] || c->inter_valid[5] || c->inter_valid[6] || c->inter_valid[7]) {
      int pass;
      p8_valid = 1;
      for (pass = c->transform8x8 ? 1 : 0; pass >= 0; pass--) {                   /* pass 1: tr8x8, pass 0: tr4x4 */
        luma_result *PR = pass ? &P8T : &P8;
        jmo_dist pcost = 0;
        memset(PR, 0, sizeof *PR);
        for (block = 0; block < 4; block++) {
          const int x0 = (block & 1) * 8, y0 = (block >> 1) * 8;
          jmo_dist min8 = JMO_DIST_MAX;
          int any = 0, bm = 0, br = 0;
          for (mode = 4; mode < (pass ? 5 : 8); mode++) {
            jmo_dist cost;
            int bref = 0;
            const int bw = PART_W[mode], bh = PART_H[mode];
            if (!c->inter_valid[mode]) continue;
            any = 1;
            for (ref = 0; ref < c->num_ref; ref++) {                              /* SubPartitionMotionSearch mv_search.c:1796-1830 */
              int v, h;
              e->motion_costl[mode][e->cl][ref][block] = 0;
              for (v = y0; v < y0 + 8; v += bh)
                for (h = x0; h < x0 + 8; h += bw) {
                  jmo_mv mv;
                  e->motion_costl[mode][e->cl][ref][block] += block_motion_search(e, ref, mode, h, v, bw, bh, &mv);
                  for (j = 0; j < bh / 4; j++) for (i = 0; i < bw / 4; i++) e->all_mvl[e->cl][ref][mode][v / 4 + j][h / 4 + i] = mv;
                  set_me_parameters(e, mv, ref, h / 4, v / 4, bw / 4, bh / 4);
                }
            }
            cost = list0_cost(e, mode, block, &bref);
            for (j = 0; j < 2; j++) for (i = 0; i < 2; i++) e->mil[e->cl][(e->mby * 4 + y0 / 4 + j) * e->w4 + e->mbx * 4 + x0 / 4 + i].ref = (int8_t)bref;   /* :847-854 */
            if (cost != JMO_DIST_MAX) cost += (c->num_ref <= 1 ? 0 : (jmo_dist)c->lambda_mf[2] * c->refbits[mode - 4]) - 1;    /* :898-900: ref_cost(.., B8Mode2Value = mode - 4, ..) - 1 */
            if (cost < min8) { min8 = cost; bm = mode; br = bref; }
          }
          if (!any) continue;
          if (pass) { p8tmode[block] = (int8_t)bm; p8tref[block] = br; } else { p8mode[block] = (int8_t)bm; p8ref[block] = br; }
          if (min8 != JMO_DIST_MAX && pcost != JMO_DIST_MAX) pcost += min8; else pcost = JMO_DIST_MAX;
          {
            const int cnt = luma_rc_8x8(e, PR, block, bm, br, pass);
            if (cnt) { PR->cnt_nonz += cnt; }
            /* cbp8x8: the block's bit when the returned coefficient cost is non-zero (:1010-1014); luma_rc_8x8 set the cbp from the blocks' nonzero flags,
             * which is the same condition unless the cost is 0 with coefficients present -- impossible: every kept level costs >= 0 and a block
             * survives reset_block only with cost > 4.  Kept as JM writes it: */
            if (!cnt) PR->cbp &= ~(1 << block);
          }
          if (pass) for (j = 0; j < 2; j++) for (i = 0; i < 2; i++) p8tmv[(y0 / 4 + j) * 4 + x0 / 4 + i] = e->all_mvl[e->cl][br][bm][y0 / 4 + j][x0 / 4 + i];    /* store_8x8_motion_vectors */
          for (j = 0; j < 2; j++)                                                  /* set_ref_and_motion_vectors_P_slice rdopt.c:2885, 8x8 region */
            for (i = 0; i < 2; i++) {
              mvinfo *m = &e->mil[e->cl][(e->mby * 4 + y0 / 4 + j) * e->w4 + e->mbx * 4 + x0 / 4 + i];
              m->mv = e->all_mvl[e->cl][br][bm][y0 / 4 + j][x0 / 4 + i]; m->ref = (int8_t)br;
            }
        }
        if (pass) p8tcost = pcost; else p8cost = pcost;
      }
      if (c->transform8x8) cur_t8 = 0;                                            /* md_low.c:289 */
      if (p8cost < min_cost || (c->transform8x8 && p8tcost < min_cost)) {         /* md_low.c:312-348 */
        best_mode = 8;
        if (c->transform8x8) {
          if (p8tcost < p8cost) { min_cost = p8tcost; p8_t8 = 1; }
          else if (p8cost < p8tcost) { min_cost = p8cost; p8_t8 = 0; }
          else {                                                                  /* get_best_transform_8x8 md_low.c:43 */
            jmo_dist c4, c8;
            satd_4_and_8(e, P8.pred, P8T.pred, &c4, &c8);
            p8_t8 = c8 < c4;
            min_cost = p8_t8 ? p8tcost : p8cost;
          }
        } else { min_cost = p8cost; p8_t8 = 0; }
        cur_t8 = p8_t8;
      }
    }
    { jmo_mv s = skip_mv(e); for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) e->all_mvl[e->cl][0][0][j][i] = s; }        /* md_low.c:357-358 */
  }
  min_rdcost = min_cost;

  /* ---- Intra8x8: md_low.c:432-486 (enc_mb.valid[I8MB] = Transform8x8Mode wherever Intra4x4 is allowed) ---- */
  tmp8 = cur_t8;                                                  /* tmp_8x8_flag md_low.c:368 */
  if (c->transform8x8 && c->intra8_valid) {
    jmo_mb_record t;
    int cbp8 = 0;
    memset(&t, 0, sizeof t);
    memset(ipm8, 2, sizeof ipm8);
    rd_cost = intra8x8_mb(e, &t, ipm8, &cbp8);
    if (rd_cost <= min_rdcost) {
      min_rdcost = rd_cost; best_mode = 13; cur_t8 = 1; tmp8 = 1;
      memcpy(o->luma, t.luma, sizeof o->luma); memcpy(o->ipredmode, t.ipredmode, 16); memcpy(o->ipred_syntax, t.ipred_syntax, 16);
      o->cbp = (int16_t)cbp8;
      for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) i8rec[j * 16 + i] = e->rec[0][(e->mby * 16 + j) * W + e->mbx * 16 + i];      /* temp_img */
    } else cur_t8 = tmp8;
  }
  /* ---- Intra4x4: md_low.c:489-521 ---- */
  {
    jmo_mb_record t;
    int cbp4 = 0;
    if (c->intra4_valid) {
      memset(&t, 0, sizeof t);
      rd_cost = intra4x4_mb(e, &t, &cbp4);
      if (rd_cost <= min_rdcost) {
        cur_t8 = 0; tmp8 = 0;
        min_rdcost = rd_cost; best_mode = 9;
        memcpy(o->luma, t.luma, sizeof o->luma); memcpy(o->ipredmode, t.ipredmode, 16); memcpy(o->ipred_syntax, t.ipred_syntax, 16);
        o->cbp = (int16_t)cbp4;
      } else cur_t8 = tmp8;
    }
  }
  /* ---- Intra16x16: md_low.c:522-556 ---- */
  if (c->intra16_valid) {
    jmo_pel ee[33], pred4[4][256];
    int left, up, upleft, mask, i16 = 2;
    intra16_neighbours(e, ee, &left, &up, &upleft);
    mask = (up ? 1 : 0) | (left ? 2 : 0) | 4 | ((left && up && upleft) ? 8 : 0);
    cur_t8 = 0;                                                  /* find_best_mode_I16x16_MB rd_intra_jm.c:423 */
    rd_cost = (int)jmo_intra16_search(ee, left, up, mask, 2, 255, e->orig, pred4, &i16);
    o->i16mode = (int8_t)i16;
    if (rd_cost < min_rdcost) {
      jmo_qparam q[16];
      int dc_level[17], dc_run[17], ac_level[16][16], ac_run[16][16], fadj[64], pos, b;
      jmo_pel rec[256];
      best_mode = 10; min_rdcost = rd_cost;
      jmo_qparams_4x4_m(c->qp, c->off4[0][1], q);
      dc_level[0] = 0;
      o->cbp = (int16_t)jmo_rtq_luma_16x16(e->orig, pred4[i16], q, c->qp / 6, !c->cabac, 0, 0, 255, dc_level, dc_run, ac_level, ac_run, rec, fadj);
      memset(o->luma, 0, sizeof o->luma);
      for (pos = 0, k = 0; k < 16 && dc_level[k] != 0; k++) { pos += dc_run[k]; o->luma_dc[pos++] = (int16_t)dc_level[k]; }
      for (b = 0; b < 16; b++)
        for (pos = 1, k = 0; k < 15 && ac_level[b][k] != 0; k++) { pos += ac_run[b][k]; o->luma[b][pos++] = (int16_t)ac_level[b][k]; }
      store_luma(e, rec);
    } else cur_t8 = tmp8;                                        /* md_low.c:553 */
  }

  /* ---- final parameters: md_low.c:560-669 ---- */
  o->mb_type = (int8_t)best_mode;
  o->min_rdcost = min_rdcost;
  {
    int ref8[4] = {0, 0, 0, 0};
    if (best_mode == 8) {                                        /* set_coeff_and_recon_8x8_p_slice rdopt.c:1326-1545 */
      if (cur_t8 && P8T.cbp == 0) cur_t8 = 0;                    /* md_low.c:568-569: then the tr4x4 data are used */
      if (cur_t8) {
        for (k = 0; k < 4; k++) { o->b8mode[k] = p8tmode[k]; ref8[k] = p8tref[k]; }
        for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) e->all_mvl[e->cl][ref8[(j >> 1) * 2 + (i >> 1)]][4][j][i] = p8tmv[j * 4 + i];      /* RestoreMV8x8 */
        memcpy(o->luma, P8T.lev, sizeof o->luma);
        if (P8T.cnt_nonz <= 5) { o->cbp = 0; o->cbp_blk = 0; store_luma(e, P8T.pred); }       /* _LUMA_8x8_COEFF_COST_ */
        else { o->cbp = (int16_t)P8T.cbp; o->cbp_blk = (uint64_t)P8T.cbp_blk; store_luma(e, P8T.rec); }
      } else {
      for (k = 0; k < 4; k++) { o->b8mode[k] = p8mode[k]; ref8[k] = p8ref[k]; }
      memcpy(o->luma, P8.lev, sizeof o->luma);
      if (P8.cnt_nonz <= 5) { o->cbp = 0; o->cbp_blk = 0; store_luma(e, P8.pred); }
      else { o->cbp = (int16_t)P8.cbp; o->cbp_blk = (uint64_t)P8.cbp_blk; store_luma(e, P8.rec); }
      }
    } else if (best_mode >= 1 && best_mode <= 3) {
      for (k = 0; k < 4; k++) { o->b8mode[k] = (int8_t)best_mode; ref8[k] = best_ref[best_mode][k]; }
      cur_t8 = best_transform_flag;                              /* md_low.c:607-608 */
      luma_rc_mb(e, &L, best_mode, ref8, cur_t8);
      memcpy(o->luma, L.lev, sizeof o->luma);
      o->cbp = (int16_t)L.cbp; o->cbp_blk = (uint64_t)L.cbp_blk;
      store_luma(e, L.rec);
    } else if (best_mode == 9) {
      for (k = 0; k < 4; k++) o->b8mode[k] = 11;               /* IBLOCK */
    } else if (best_mode == 13) {
      for (k = 0; k < 4; k++) o->b8mode[k] = 13;               /* I8MB (set_modes_and_refs_for_blocks) */
      store_luma(e, i8rec);                                     /* md_low.c:585 */
      for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) e->ipm[(e->mby * 4 + j) * e->w4 + e->mbx * 4 + i] = ipm8[j * 4 + i];
    }
    if (best_mode != 9 && best_mode != 13) { memset(o->ipredmode, 2, 16); memset(o->ipred_syntax, 2, 16); for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) e->ipm[(e->mby * 4 + j) * e->w4 + e->mbx * 4 + i] = 2; }
    if ((o->cbp & 15) == 0 && best_mode != 9 && best_mode != 13) cur_t8 = 0;      /* md_low.c:627-628 */
    for (k = 0; k < 4; k++) o->b8ref[k] = (int8_t)(best_mode >= 9 ? -1 : ref8[k]);
    chroma_mb(e, o, best_mode >= 9, best_mode, o->b8mode, ref8);
    /* SetMotionVectorsMBPSlice md_common.c:172 + the reference indices of set_modes_and_refs_for_blocks_p_slice rdopt.c:1041-1130 */
    for (j = 0; j < 4; j++)
      for (i = 0; i < 4; i++) {
        mvinfo *m = &e->mil[e->cl][(e->mby * 4 + j) * e->w4 + e->mbx * 4 + i];
        const int b8 = (j >> 1) * 2 + (i >> 1);
        if (best_mode >= 9) { m->mv.x = m->mv.y = 0; m->ref = -1; }
        else { m->ref = (int8_t)ref8[b8]; m->mv = e->all_mvl[e->cl][ref8[b8]][o->b8mode[b8]][j][i]; }
      }
    /* the skip test md_low.c:658-665 */
    if (pslice && best_mode == 1 && o->cbp == 0 && ref8[0] == 0) {
      const mvinfo *m = &e->mil[e->cl][(e->mby * 4) * e->w4 + e->mbx * 4];
      if (m->mv.x == e->all_mvl[e->cl][0][0][0][0].x && m->mv.y == e->all_mvl[e->cl][0][0][0][0].y) { o->mb_type = 0; memset(o->b8mode, 0, 4); cur_t8 = 0; }
    }
    o->transform8x8 = (int8_t)cur_t8;
    if (best_mode == 10) { /* i16offset is derived by the caller: I16Offset(cbp, i16mode) rdopt.c:868 */ }
    for (j = 0; j < 4; j++)
      for (i = 0; i < 4; i++) { const mvinfo *m = &e->mil[e->cl][(e->mby * 4 + j) * e->w4 + e->mbx * 4 + i]; o->mv[j * 4 + i][0] = m->mv.x; o->mv[j * 4 + i][1] = m->mv.y; }
  }
  if (dbg) {
    memset(dbg, 0, sizeof *dbg);
    for (mode = 1; mode < 8; mode++) {
      for (k = 0; k < 4; k++) dbg->motion_cost[mode][k] = e->motion_costl[mode][e->cl][0][k];
      for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) { dbg->all_mv[mode][j * 4 + i][0] = e->all_mvl[e->cl][0][mode][j][i].x; dbg->all_mv[mode][j * 4 + i][1] = e->all_mvl[e->cl][0][mode][j][i].y; }
    }
    dbg->best_mode = best_mode;
    (void)p8_valid;
  }
}

/* Encode the macroblocks [first_mb, first_mb + num_mb) of one slice in raster order.
 * cur_y/u/v: source planes at the coded size.  refs[num_ref]: luma quarter-pel planes; refc[2*num_ref]: integer chroma planes (U, V per reference).
 * rec_y/u/v: reconstruction planes of the picture (read for intra prediction, written).  mv / ref_idx / ipredmode: per-4x4 picture arrays
 * (mv[(y4 * w4 + x4) * 2 + {0,1}]), carried across the slices of a picture by the caller.  out[num_mb], dbg[num_mb] or NULL. */
int jmo_encode_slice(const jmo_mbenc_cfg *cfg, const jmo_pel *cur_y, const jmo_pel *cur_u, const jmo_pel *cur_v,
                     const jmo_refpic *refs, const jmo_pel *const *refc, jmo_pel *rec_y, jmo_pel *rec_u, jmo_pel *rec_v,
                     int16_t *mv, int8_t *ref_idx, int8_t *ipredmode, jmo_mb_record *out, jmo_mb_debug *dbg)
{
  return jmo_encode_slice_ex(cfg, NULL, cur_y, cur_u, cur_v, refs, refc, rec_y, rec_u, rec_v, mv, ref_idx, ipredmode, out, dbg);
}

int jmo_encode_slice_ex(const jmo_mbenc_cfg *cfg, jmo_epzs_cfg *ez, const jmo_pel *cur_y, const jmo_pel *cur_u, const jmo_pel *cur_v,
                        const jmo_refpic *refs, const jmo_pel *const *refc, jmo_pel *rec_y, jmo_pel *rec_u, jmo_pel *rec_v,
                        int16_t *mv, int8_t *ref_idx, int8_t *ipredmode, jmo_mb_record *out, jmo_mb_debug *dbg)
{
  enc *e = (enc *)calloc(1, sizeof(enc));
  int k, n4;
  if (!e) return -1;
  if (cfg->num_ref > JMO_MAX_REF || (cfg->slice_type != 0 && cfg->slice_type != 2)) { free(e); return -2; }
  if (cfg->search_mode == 1 && cfg->slice_type == 0) {          /* fast full search: see the scope note in block_motion_search */
    const int rq = cfg->search_range << 2;
    if (cfg->mv_limit[3] - rq < rq || cfg->mv_limit[2] + rq > -rq || cfg->mv_limit[1] - rq < rq || cfg->mv_limit[0] + rq > -rq) { free(e); return -4; }
  }
  if (cfg->search_mode == 3 && cfg->slice_type == 0) {
    if (!ez) { free(e); return -3; }
    e->ez = (struct epzs_state *)epzs_new(cfg, ez);
    if (!e->ez) { free(e); return -1; }
  }
  e->c = cfg; e->wmb = cfg->width / 16; e->hmb = cfg->height / 16; e->w4 = cfg->width / 4;
  n4 = e->w4 * (cfg->height / 4);
  e->cur[0] = cur_y; e->cur[1] = cur_u; e->cur[2] = cur_v;
  e->refl[0] = refs; e->refcl[0] = refc; e->cl = 0;
  e->rec[0] = rec_y; e->rec[1] = rec_u; e->rec[2] = rec_v;
  e->mil[0] = (mvinfo *)malloc(sizeof(mvinfo) * (size_t)n4);
  for (k = 0; k < n4; k++) { e->mil[e->cl][k].mv.x = mv[2 * k]; e->mil[e->cl][k].mv.y = mv[2 * k + 1]; e->mil[e->cl][k].ref = ref_idx[k]; }
  e->ipm = ipredmode;
  for (k = 0; k < cfg->num_mb; k++) {
    e->addr = cfg->first_mb + k; e->mbx = e->addr % e->wmb; e->mby = e->addr / e->wmb;
    encode_mb(e, &out[k], dbg ? &dbg[k] : NULL);
  }
  for (k = 0; k < n4; k++) { mv[2 * k] = e->mil[e->cl][k].mv.x; mv[2 * k + 1] = e->mil[e->cl][k].mv.y; ref_idx[k] = e->mil[e->cl][k].ref; }
  epzs_free((epzs *)e->ez);
  free(e->spiral); free(e->mil[0]); free(e);
  return 0;
}

int jmo_encode_slice_b(const jmo_mbenc_cfg *cfg, const jmo_b_cfg *bcfg, const jmo_pel *cur_y, const jmo_pel *cur_u, const jmo_pel *cur_v,
                       const jmo_refpic *refs, const jmo_pel *const *refc, const jmo_refpic *refs1, const jmo_pel *const *refc1,
                       jmo_pel *rec_y, jmo_pel *rec_u, jmo_pel *rec_v, int16_t *mv, int8_t *ref_idx, int16_t *mv1, int8_t *ref_idx1, int8_t *ipredmode,
                       jmo_mb_record *out, jmo_mb_debug *dbg)
{
  enc *e = (enc *)calloc(1, sizeof(enc));
  int k, n4, l;
  if (!e) return -1;
  if (cfg->num_ref > JMO_MAX_REF || bcfg->num_ref1 > JMO_MAX_REF || cfg->num_ref < 1 || bcfg->num_ref1 < 1 || cfg->slice_type != 1 || cfg->search_mode == 3) { free(e); return -2; }
  if (cfg->search_mode == 1) {                                    /* fast full search: see the scope note in block_motion_search */
    const int rq = cfg->search_range << 2;
    if (cfg->mv_limit[3] - rq < rq || cfg->mv_limit[2] + rq > -rq || cfg->mv_limit[1] - rq < rq || cfg->mv_limit[0] + rq > -rq) { free(e); return -4; }
  }
  e->c = cfg; e->bc = bcfg; e->wmb = cfg->width / 16; e->hmb = cfg->height / 16; e->w4 = cfg->width / 4;
  n4 = e->w4 * (cfg->height / 4);
  e->cur[0] = cur_y; e->cur[1] = cur_u; e->cur[2] = cur_v;
  e->refl[0] = refs; e->refcl[0] = refc; e->refl[1] = refs1; e->refcl[1] = refc1; e->cl = 0;
  e->rec[0] = rec_y; e->rec[1] = rec_u; e->rec[2] = rec_v;
  for (l = 0; l < 2; l++) {
    const int16_t *m = l ? mv1 : mv;
    const int8_t *r = l ? ref_idx1 : ref_idx;
    e->mil[l] = (mvinfo *)malloc(sizeof(mvinfo) * (size_t)n4);
    for (k = 0; k < n4; k++) { e->mil[l][k].mv.x = m[2 * k]; e->mil[l][k].mv.y = m[2 * k + 1]; e->mil[l][k].ref = r[k]; }
  }
  e->ipm = ipredmode;
  for (k = 0; k < cfg->num_mb; k++) {
    e->addr = cfg->first_mb + k; e->mbx = e->addr % e->wmb; e->mby = e->addr / e->wmb;
    encode_mb_b(e, &out[k], dbg ? &dbg[k] : NULL);
  }
  for (l = 0; l < 2; l++) {
    int16_t *m = l ? mv1 : mv;
    int8_t *r = l ? ref_idx1 : ref_idx;
    for (k = 0; k < n4; k++) { m[2 * k] = e->mil[l][k].mv.x; m[2 * k + 1] = e->mil[l][k].mv.y; r[k] = e->mil[l][k].ref; }
    free(e->mil[l]);
  }
  free(e->spiral); free(e);
  return 0;
}
