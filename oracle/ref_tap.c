/*
 * oracle/ref_tap.c -- TEST INFRASTRUCTURE: taps on the REAL reference encoder.
 *
 * Linked (by oracle/Makefile.ref, target `tap`) together with the unmodified JM 19.0
 * lencod objects using GNU ld's --wrap, this file intercepts the reference's own
 * hot-path entry points while the real encoder runs a real configuration, and dumps
 * their inputs and outputs.  tests/golden/make_golden.py turns the dumps into the
 * small committed fixtures that pin the oracle (and, through it, the HIP kernels).
 *
 * It is our own code written against JM's public headers; it contains no reference
 * source.  It only builds where /root/reference is present and only into oracle/_ref/.
 *
 * Tapped (reference file:line):
 *   full_search_motion_estimation      lencod/src/me_fullsearch.c:39
 *   sub_pel_motion_estimation          lencod/src/me_fullsearch.c:186
 *   setup_fast_full_search             lencod/src/me_fullfast.c:269
 *   fast_full_search_motion_estimation lencod/src/me_fullfast.c:618
 *   getSubImagesLuma                   lencod/src/img_luma.c:611
 *   DeblockFrame                       lencod/src/loopFilter.c:63
 *   forward4x4 / inverse4x4            lcommon/src/transform.c:20 / :70
 *   quant_4x4_normal / quant_4x4_around  lencod/src/quant4x4_normal.c:39 / quant4x4_around.c:40
 *   sample_reconstruct                 lcommon/src/blk_prediction.c:48
 * Output directory: $JM_TAP_DIR (default "."); record budget: $JM_TAP_MAX (default 6000 per stream).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "global.h"
#include "image.h"
#include "mbuffer.h"
#include "mv_search.h"
#include "me_fullsearch.h"
#include "me_fullfast.h"
#include "transform.h"
#include "quant4x4.h"
#include "blk_prediction.h"
#include "loop_filter.h"
#include "img_luma.h"
#include "intra8x8.h"

static FILE *tap_open(const char *name)
{
  char path[1024];
  const char *d = getenv("JM_TAP_DIR");
  snprintf(path, sizeof path, "%s/%s", d ? d : ".", name);
  return fopen(path, "ab");
}
static int tap_max(void) { const char *m = getenv("JM_TAP_MAX"); return m ? atoi(m) : 6000; }
static void put_i32(FILE *f, int v) { int32_t x = v; fwrite(&x, 4, 1, f); }
static void put_i64(FILE *f, int64 v) { int64_t x = v; fwrite(&x, 8, 1, f); }
static void put_plane(FILE *f, imgpel **rows, int y0, int x0, int h, int w)
{
  int y;
  put_i32(f, h); put_i32(f, w);
  for (y = 0; y < h; y++) fwrite(&rows[y0 + y][x0], sizeof(imgpel), (size_t)w, f);
}

static int g_refs_made = 0;        /* number of getSubImagesLuma calls so far = reference pictures produced */
static StorablePicture *g_ref_ptr[64];   /* reference picture made by call number k (mod 64): identity for the prediction taps */
static int ref_index_of(StorablePicture *s)
{
  int k;
  for (k = g_refs_made - 1; k >= 0 && k >= g_refs_made - 64; k--) if (g_ref_ptr[k & 63] == s) return k;
  return -1;
}
static int g_cur_dumped = -1;

static void dump_cur_frame(VideoParameters *p_Vid)
{
  if (g_cur_dumped != g_refs_made) {
    FILE *f = tap_open("cur_frames.bin");
    put_i32(f, g_refs_made);
    put_plane(f, p_Vid->pCurImg, 0, 0, p_Vid->height, p_Vid->width);
    fclose(f);
    /* the whole source picture as read_one_frame + pad_borders left it (lcommon/src/input.c:792, :880): Y, U, V at the coded size */
    f = tap_open("cur_yuv.bin");
    put_i32(f, g_refs_made); put_i32(f, p_Vid->yuv_format);
    put_plane(f, p_Vid->pImgOrg[0], 0, 0, p_Vid->height, p_Vid->width);
    if (p_Vid->yuv_format != YUV400) {
      put_plane(f, p_Vid->pImgOrg[1], 0, 0, p_Vid->height_cr, p_Vid->width_cr);
      put_plane(f, p_Vid->pImgOrg[2], 0, 0, p_Vid->height_cr, p_Vid->width_cr);
    }
    fclose(f);
    g_cur_dumped = g_refs_made;
  }
}

/* ------------------------------------------------------------------ ME: full search */
extern distblk __real_full_search_motion_estimation(Macroblock *, MotionVector *, MEBlock *, distblk, int);
distblk __wrap_full_search_motion_estimation(Macroblock *currMB, MotionVector *pred_mv, MEBlock *mv_block,
                                             distblk min_mcost, int lambda_factor)
{
  static int n = 0;
  MotionVector c = mv_block->mv[(int)mv_block->list];
  distblk r = __real_full_search_motion_estimation(currMB, pred_mv, mv_block, min_mcost, lambda_factor);
  if (n < tap_max()) {
    FILE *f = tap_open("me_fs.bin");
    dump_cur_frame(currMB->p_Vid);
    put_i32(f, g_refs_made); put_i32(f, mv_block->ref_idx); put_i32(f, mv_block->blocktype);
    put_i32(f, mv_block->pos_x); put_i32(f, mv_block->pos_y);
    put_i32(f, mv_block->blocksize_x); put_i32(f, mv_block->blocksize_y);
    put_i32(f, pred_mv->mv_x); put_i32(f, pred_mv->mv_y);
    put_i32(f, c.mv_x); put_i32(f, c.mv_y);
    put_i32(f, imin(mv_block->searchRange.max_x, mv_block->searchRange.max_y) >> 2);
    put_i32(f, lambda_factor);
    put_i64(f, min_mcost);
    put_i32(f, mv_block->mv[(int)mv_block->list].mv_x); put_i32(f, mv_block->mv[(int)mv_block->list].mv_y);
    put_i64(f, r);
    fclose(f); n++;
  }
  return r;
}

/* ------------------------------------------------------------------ ME: sub-pel */
extern distblk __real_sub_pel_motion_estimation(Macroblock *, MotionVector *, MEBlock *, distblk, int *);
distblk __wrap_sub_pel_motion_estimation(Macroblock *currMB, MotionVector *pred, MEBlock *mv_block,
                                         distblk min_mcost, int *lambda)
{
  static int n = 0;
  VideoParameters *p_Vid = currMB->p_Vid;
  MotionVector in = mv_block->mv[(int)mv_block->list];
  distblk r = __real_sub_pel_motion_estimation(currMB, pred, mv_block, min_mcost, lambda);
  if (n < tap_max()) {
    FILE *f = tap_open("me_subpel.bin");
    dump_cur_frame(p_Vid);
    put_i32(f, g_refs_made); put_i32(f, mv_block->ref_idx); put_i32(f, mv_block->blocktype);
    put_i32(f, mv_block->pos_x); put_i32(f, mv_block->pos_y);
    put_i32(f, mv_block->blocksize_x); put_i32(f, mv_block->blocksize_y);
    put_i32(f, pred->mv_x); put_i32(f, pred->mv_y);
    put_i32(f, in.mv_x); put_i32(f, in.mv_y);
    put_i32(f, lambda[H_PEL]); put_i32(f, lambda[Q_PEL]);
    put_i32(f, p_Vid->p_Inp->MEErrorMetric[H_PEL]); put_i32(f, p_Vid->p_Inp->MEErrorMetric[Q_PEL]);
    put_i32(f, p_Vid->start_me_refinement_hp); put_i32(f, p_Vid->start_me_refinement_qp);
    put_i32(f, mv_block->test8x8);
    put_i64(f, min_mcost);
    put_i32(f, mv_block->mv[(int)mv_block->list].mv_x); put_i32(f, mv_block->mv[(int)mv_block->list].mv_y);
    put_i64(f, r);
    fclose(f); n++;
  }
  return r;
}

/* ------------------------------------------------------------------ ME: fast full search */
extern void __real_setup_fast_full_search(Macroblock *, MEBlock *, int);
void __wrap_setup_fast_full_search(Macroblock *currMB, MEBlock *mv_block, int list)
{
  static int n = 0;
  VideoParameters *p_Vid = currMB->p_Vid;
  __real_setup_fast_full_search(currMB, mv_block, list);
  if (n < 40) {                                  /* tables are big: 7*16*max_pos u32 each */
    MEFullFast *ff = p_Vid->p_ffast_me;
    int ref = mv_block->ref_idx, R = ff->max_search_range[list][ref];
    int max_pos = (2 * R + 1) * (2 * R + 1), t, k;
    FILE *f = tap_open("me_ffs_setup.bin");
    dump_cur_frame(p_Vid);
    put_i32(f, g_refs_made); put_i32(f, ref); put_i32(f, currMB->pix_x); put_i32(f, currMB->opix_y);
    put_i32(f, ff->search_center[list][ref].mv_x); put_i32(f, ff->search_center[list][ref].mv_y);
    put_i32(f, R); put_i32(f, max_pos);
    for (t = 1; t < 8; t++)
      for (k = 0; k < 16; k++)
        fwrite(ff->BlockSAD[list][ref][t][k], sizeof(distpel), (size_t)max_pos, f);
    fclose(f); n++;
  }
}
extern distblk __real_fast_full_search_motion_estimation(Macroblock *, MotionVector *, MEBlock *, distblk, int);
distblk __wrap_fast_full_search_motion_estimation(Macroblock *currMB, MotionVector *pred_mv, MEBlock *mv_block,
                                                  distblk min_mcost, int lambda_factor)
{
  static int n = 0;
  VideoParameters *p_Vid = currMB->p_Vid;
  distblk r = __real_fast_full_search_motion_estimation(currMB, pred_mv, mv_block, min_mcost, lambda_factor);
  if (n < tap_max()) {
    int list = mv_block->list, ref = mv_block->ref_idx;
    FILE *f = tap_open("me_ffs.bin");
    dump_cur_frame(p_Vid);
    put_i32(f, g_refs_made); put_i32(f, ref); put_i32(f, mv_block->blocktype);
    put_i32(f, currMB->pix_x); put_i32(f, currMB->opix_y);
    put_i32(f, mv_block->block_x); put_i32(f, mv_block->block_y);
    put_i32(f, pred_mv->mv_x); put_i32(f, pred_mv->mv_y);
    put_i32(f, p_Vid->p_ffast_me->search_center[list][ref].mv_x); put_i32(f, p_Vid->p_ffast_me->search_center[list][ref].mv_y);
    put_i32(f, imax(mv_block->searchRange.max_x, mv_block->searchRange.max_y) >> 2);
    put_i32(f, p_Vid->p_ffast_me->max_search_range[list][ref]);
    put_i32(f, lambda_factor); put_i32(f, p_Vid->max_mvd);
    put_i64(f, min_mcost);
    put_i32(f, mv_block->mv[list].mv_x); put_i32(f, mv_block->mv[list].mv_y);
    put_i64(f, r);
    fclose(f); n++;
  }
  return r;
}

/* ------------------------------------------------------------------ sub-pel planes */
extern void __real_getSubImagesLuma(VideoParameters *, StorablePicture *);
void __wrap_getSubImagesLuma(VideoParameters *p_Vid, StorablePicture *s)
{
  __real_getSubImagesLuma(p_Vid, s);
  if (g_refs_made < 4) {
    FILE *f = tap_open("subimages.bin");
    int j, i;
    put_i32(f, g_refs_made); put_i32(f, s->size_x); put_i32(f, s->size_y); put_i32(f, p_Vid->max_imgpel_value);
    put_plane(f, s->imgY, 0, 0, s->size_y, s->size_x);
    for (j = 0; j < 4; j++)
      for (i = 0; i < 4; i++)
        put_plane(f, s->p_curr_img_sub[j][i], -IMG_PAD_SIZE_Y, -IMG_PAD_SIZE_X, s->size_y_padded, s->size_x_padded);
    fclose(f);
    if (p_Vid->yuv_format != YUV400) {           /* the integer chroma planes of the same picture (chroma prediction) */
      f = tap_open("refchroma.bin");
      put_i32(f, g_refs_made); put_i32(f, p_Vid->yuv_format);
      put_plane(f, s->imgUV[0], 0, 0, s->size_y_cr, s->size_x_cr);
      put_plane(f, s->imgUV[1], 0, 0, s->size_y_cr, s->size_x_cr);
      fclose(f);
    }
  }
  g_ref_ptr[g_refs_made & 63] = s;
  g_refs_made++;
}

/* ------------------------------------------------------------------ chroma sub-images (K6)
 * getSubImagesChroma (lencod/src/img_chroma.c:338; called by UnifiedOneForthPix, image.c:2205, when ChromaMCBuffer = 1): for the first
 * reference pictures, every sub-image of both planes with its padding.
 * record: picture index, yuv_format, subimages_y, subimages_x, pad_y, pad_x | per plane, per (suby, subx): the padded plane */
extern void __real_getSubImagesChroma(VideoParameters *, StorablePicture *);
void __wrap_getSubImagesChroma(VideoParameters *p_Vid, StorablePicture *s)
{
  static int n = 0;
  __real_getSubImagesChroma(p_Vid, s);
  if (n < 2 && (p_Vid->yuv_format == YUV420 || p_Vid->yuv_format == YUV422) && !p_Vid->p_Inp->OnTheFlyFractMCP) {
    const int ny = p_Vid->yuv_format == YUV420 ? 8 : 4, nx = 8, py = p_Vid->pad_size_uv_y, px = p_Vid->pad_size_uv_x;
    int uv, j, i;
    FILE *f = tap_open("chromasub.bin");
    put_i32(f, g_refs_made - 1); put_i32(f, p_Vid->yuv_format); put_i32(f, ny); put_i32(f, nx); put_i32(f, py); put_i32(f, px);
    for (uv = 0; uv < 2; uv++)
      for (j = 0; j < ny; j++)
        for (i = 0; i < nx; i++) put_plane(f, s->p_img_sub[uv + 1][j][i], -py, -px, s->size_y_cr + 2 * py, s->size_x_cr + 2 * px);
    fclose(f);
  }
  n++;
}

/* ------------------------------------------------------------------ motion-compensated prediction
 *   luma_prediction          lencod/src/mc_prediction.c:144   (bound to p_Dpb->pf_luma_prediction in lencod.c:367)
 *   chroma_prediction_4x4    lencod/src/mc_prediction.c:568
 * Records: the block, the direction, per list {reference picture made by getSubImagesLuma call k, motion vector(s)} exactly as the
 * function resolves them, and the samples it left in currSlice->mb_pred. */
/* the weighted-prediction parameters exactly as luma_prediction (:203-228) / chroma_prediction_4x4 (:615-640) hand them to
 * weighted_mc_prediction / weighted_bi_prediction: weight[list], offset, round, shift (zeros when the call is un-weighted) */
static void put_mc_weights(FILE *f, Slice *sl, int wp, int p_dir, int r0, int r1, int comp)
{
  int w[2] = {0, 0}, off = 0, rnd = 0, sh = 0;
  const int round_ = comp ? sl->wp_chroma_round : sl->wp_luma_round, denom = comp ? sl->chroma_log_weight_denom : sl->luma_log_weight_denom;
  if (wp && p_dir == 2) {
    w[0] = sl->wbp_weight[0][r0][r1][comp]; w[1] = sl->wbp_weight[1][r0][r1][comp];
    off = (sl->wp_offset[0][r0][comp] + sl->wp_offset[1][r1][comp] + 1) >> 1; rnd = round_ << 1; sh = denom + 1;
  } else if (wp) {
    const int r = p_dir ? r1 : r0;
    w[p_dir] = sl->wp_weight[p_dir][r][comp]; off = sl->wp_offset[p_dir][r][comp]; rnd = round_; sh = denom;
  }
  put_i32(f, w[0]); put_i32(f, w[1]); put_i32(f, off); put_i32(f, rnd); put_i32(f, sh);
}
static MotionVector *****mc_mv_array(Macroblock *currMB, int p_dir, int m0, int m1, int r0, int r1, short bipred_me)
{
  Slice *sl = currMB->p_Slice;
  if (bipred_me && r0 == 0 && r1 == 0 && p_dir == 2 && is_bipred_enabled(currMB->p_Vid, m0) && is_bipred_enabled(currMB->p_Vid, m1))
    return sl->bipred_mv[bipred_me - 1];
  return sl->all_mv;
}
extern void __real_luma_prediction(Macroblock *, int, int, int, int, int, int *, char *, short);
void __wrap_luma_prediction(Macroblock *currMB, int block_x, int block_y, int bsx, int bsy, int p_dir, int list_mode[2], char *ref_idx, short bipred_me)
{
  static int n = 0, calls = 0;
  __real_luma_prediction(currMB, block_x, block_y, bsx, bsy, p_dir, list_mode, ref_idx, bipred_me);
  if (calls++ % 16 == 0 && n < tap_max() / 4 && g_refs_made <= 4) {      /* every 16th call: the sample reaches the later pictures */
    Slice *sl = currMB->p_Slice;
    MotionVector *****mva = mc_mv_array(currMB, p_dir, list_mode[0], list_mode[1], ref_idx[0], ref_idx[1], bipred_me);
    int wp = (sl->weighted_prediction == 1) || (sl->weighted_prediction == 2 && p_dir == 2);
    int l, j;
    FILE *f = tap_open("mc_luma.bin");
    put_i32(f, g_refs_made); put_i32(f, currMB->pix_x + block_x); put_i32(f, currMB->opix_y + block_y);
    put_i32(f, bsx); put_i32(f, bsy); put_i32(f, p_dir); put_i32(f, wp);
    for (l = 0; l < 2; l++) {
      if (p_dir == l || p_dir == 2) {
        MotionVector *mv = &mva[l][(short)ref_idx[l]][list_mode[l]][block_y >> 2][block_x >> 2];
        put_i32(f, ref_index_of(sl->listX[l + currMB->list_offset][(short)ref_idx[l]])); put_i32(f, mv->mv_x); put_i32(f, mv->mv_y);
      } else { put_i32(f, -1); put_i32(f, 0); put_i32(f, 0); }
    }
    put_mc_weights(f, sl, wp, p_dir, (short)ref_idx[0], (short)ref_idx[1], 0);
    for (j = 0; j < bsy; j++) fwrite(&sl->mb_pred[0][block_y + j][block_x], sizeof(imgpel), (size_t)bsx, f);
    fclose(f); n++;
  }
}
extern void __real_chroma_prediction_4x4(Macroblock *, int, int, int, int, int, int, short, short, short);
void __wrap_chroma_prediction_4x4(Macroblock *currMB, int uv, int block_x, int block_y, int p_dir, int l0_mode, int l1_mode,
                                  short l0_ref_idx, short l1_ref_idx, short bipred_me)
{
  static int n = 0, calls = 0;
  __real_chroma_prediction_4x4(currMB, uv, block_x, block_y, p_dir, l0_mode, l1_mode, l0_ref_idx, l1_ref_idx, bipred_me);
  if (calls++ % 32 == 0 && n < tap_max() / 2 && g_refs_made <= 4) {
    VideoParameters *p_Vid = currMB->p_Vid;
    Slice *sl = currMB->p_Slice;
    MotionVector *****mva = mc_mv_array(currMB, p_dir, l0_mode, l1_mode, l0_ref_idx, l1_ref_idx, bipred_me);
    int wp = (sl->weighted_prediction == 1) || (sl->weighted_prediction == 2 && p_dir == 2);
    int rsx = 4 - p_Vid->chroma_shift_x, rsy = 4 - p_Vid->chroma_shift_y;      /* chroma sample -> luma 4x4 block index, as the function does */
    int mode[2] = {l0_mode, l1_mode}, ref[2] = {l0_ref_idx, l1_ref_idx};
    int l, j;
    FILE *f = tap_open("mc_chroma.bin");
    put_i32(f, g_refs_made); put_i32(f, p_Vid->yuv_format); put_i32(f, uv);
    put_i32(f, currMB->pix_c_x + block_x); put_i32(f, currMB->opix_c_y + block_y); put_i32(f, p_dir); put_i32(f, wp);
    put_i32(f, p_Vid->p_Inp->ChromaMCBuffer);
    for (l = 0; l < 2; l++) {
      if (p_dir == l || p_dir == 2) {
        StorablePicture *pic = sl->listX[l + currMB->list_offset][ref[l]];
        MotionVector **mv = mva[l][ref[l]][mode[l]];
        put_i32(f, ref_index_of(pic)); put_i32(f, pic->chroma_vector_adjustment);
        for (j = block_y; j < block_y + 4; j++) {          /* the two vectors each sample row uses (left / right sample pair) */
          MotionVector *a = &mv[j >> rsy][block_x >> rsx], *b = &mv[j >> rsy][(block_x + 2) >> rsx];
          put_i32(f, a->mv_x); put_i32(f, a->mv_y); put_i32(f, b->mv_x); put_i32(f, b->mv_y);
        }
      } else { int k; put_i32(f, -1); put_i32(f, 0); for (k = 0; k < 16; k++) put_i32(f, 0); }
    }
    put_mc_weights(f, sl, wp, p_dir, l0_ref_idx, l1_ref_idx, uv + 1);
    for (j = 0; j < 4; j++) fwrite(&sl->mb_pred[uv + 1][block_y + j][block_x], sizeof(imgpel), 4, f);
    fclose(f); n++;
  }
}

/* ------------------------------------------------------------------ intra prediction
 *   get_intrapred_4x4     lencod/src/intra4x4.c:521     (predictor samples currMB->intra4x4_pred[pl][0..12]: X, A..H, I..L)
 *   get_intrapred_16x16   lencod/src/intra16x16.c:307   (currMB->intra16x16_pred[pl][0..32]: corner, 16 above, 16 left)
 * record: mode left up max_pel | predictor samples | the prediction left in mpr_4x4[pl][mode] / mpr_16x16[pl][mode] */
extern void __real_get_intrapred_4x4(Macroblock *, ColorPlane, int, int, int, int, int);
void __wrap_get_intrapred_4x4(Macroblock *currMB, ColorPlane pl, int mode, int img_x, int img_y, int left, int up)
{
  static int n = 0, calls = 0;
  __real_get_intrapred_4x4(currMB, pl, mode, img_x, img_y, left, up);
  if (pl == PLANE_Y && calls++ % 23 == 0 && n < tap_max() / 4) {
    int j, i;
    FILE *f = tap_open("intra4x4.bin");
    put_i32(f, mode); put_i32(f, left); put_i32(f, up); put_i32(f, currMB->p_Vid->max_imgpel_value);
    for (i = 0; i < 13; i++) put_i32(f, currMB->intra4x4_pred[pl][i]);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) put_i32(f, currMB->p_Slice->mpr_4x4[pl][mode][j][i]);
    fclose(f); n++;
  }
}
/* get_intrapred_8x8 (lencod/src/intra8x8.c:716): record mode left up | the 25 filtered predictor samples | the 8x8 prediction */
extern void __real_get_intrapred_8x8(Macroblock *, ColorPlane, int, int, int);
void __wrap_get_intrapred_8x8(Macroblock *currMB, ColorPlane pl, int mode, int left, int up)
{
  static int n = 0, calls = 0;
  __real_get_intrapred_8x8(currMB, pl, mode, left, up);
  if (pl == PLANE_Y && calls++ % 7 == 0 && n < tap_max() / 4) {
    int j, i;
    FILE *f = tap_open("intra8x8.bin");
    put_i32(f, mode); put_i32(f, left); put_i32(f, up);
    for (i = 0; i < 25; i++) put_i32(f, currMB->intra8x8_pred[pl][i]);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) put_i32(f, currMB->p_Slice->mpr_8x8[pl][mode][j][i]);
    fclose(f); n++;
  }
}
/* find_sad_16x16_JM (lencod/src/intra16x16.c:463, Slice.find_sad_16x16): the Intra16x16 mode search -- predictor samples, the four
 * predictions (get_intrapred_16x16 is called from inside the same translation unit, so the search is tapped as a whole), the mode cost
 * Slice.distI16x16 and the strict-'<' choice.
 * record: left up upleft mode_mask metric max_pel | predictor samples[33] | source 16x16 | ret(lo,hi) i16mode | 4 x prediction 16x16 */
extern distblk __real_find_sad_16x16_JM(Macroblock *);
distblk __wrap_find_sad_16x16_JM(Macroblock *currMB)
{
  static int n = 0, calls = 0;
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currMB->p_Vid;
  InputParameters *p_Inp = currMB->p_Inp;
  distblk r = __real_find_sad_16x16_JM(currMB);
  if (!currSlice->P444_joined && calls++ % 3 == 0 && n < tap_max() / 16) {
    int left, up, all, k, j, i, mask = 0;
    FILE *f = tap_open("intra16_search.bin");
    currSlice->set_intrapred_16x16(currMB, PLANE_Y, &left, &up, &all);            /* a pure function of the picture: the flags the search used */
    for (k = 0; k < 4; k++) {                                                     /* the modes the search evaluated (intra16x16.c:483-494) */
      int off = 0;
      if (p_Inp->IntraDisableInterOnly == 0 || (currSlice->slice_type != I_SLICE && currSlice->slice_type != SI_SLICE))
        off = (p_Inp->Intra16x16ParDisable && (k == VERT_PRED_16 || k == HOR_PRED_16)) || (p_Inp->Intra16x16PlaneDisable && k == PLANE_16);
      if (!off && !((k == VERT_PRED_16 && !up) || (k == HOR_PRED_16 && !left) || (k == PLANE_16 && (!left || !up || !all)))) mask |= 1 << k;
    }
    put_i32(f, left); put_i32(f, up); put_i32(f, all); put_i32(f, mask); put_i32(f, p_Inp->ModeDecisionMetric); put_i32(f, p_Vid->max_imgpel_value);
    for (i = 0; i < 33; i++) put_i32(f, currMB->intra16x16_pred[0][i]);
    for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) put_i32(f, p_Vid->pCurImg[currMB->opix_y + j][currMB->pix_x + i]);
    put_i32(f, (int)(r & 0xffffffff)); put_i32(f, (int)(r >> 32)); put_i32(f, currMB->i16mode);
    for (k = 0; k < 4; k++) for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) put_i32(f, (mask >> k) & 1 ? currSlice->mpr_16x16[0][k][j][i] : 0);
    fclose(f); n++;
  }
  return r;
}

/* intra_chroma_prediction (lencod/src/intra_chroma.c:530, Slice.intra_chroma_prediction bound in slice.c:1135): the four chroma intra
 * predictions of both planes.  record: yuv_format up left upleft | per plane: up[8] left[16] corner | per plane, per mode: 8 x ch samples
 * (zeros for a mode the function did not write).  The neighbour flags and samples are re-derived the way the function derives them. */
extern void __real_intra_chroma_prediction(Macroblock *, int *, int *, int *);
void __wrap_intra_chroma_prediction(Macroblock *currMB, int *mb_up, int *mb_left, int *mb_up_left)
{
  static int n = 0, calls = 0;
  VideoParameters *p_Vid = currMB->p_Vid;
  Slice *sl = currMB->p_Slice;
  __real_intra_chroma_prediction(currMB, mb_up, mb_left, mb_up_left);
  if (calls++ % 5 == 0 && n < tap_max() / 8 && (p_Vid->yuv_format == YUV420 || p_Vid->yuv_format == YUV422)) {
    PixelPos a, c, d;
    int up, left, ul, uv, k, j, i;
    const int ch = p_Vid->mb_cr_size_y;
    FILE *f = tap_open("intra_chroma.bin");
    p_Vid->getNeighbour(currMB, -1, -1, p_Vid->mb_size[IS_CHROMA], &d);
    p_Vid->getNeighbour(currMB, -1,  0, p_Vid->mb_size[IS_CHROMA], &a);
    p_Vid->getNeighbour(currMB,  0, -1, p_Vid->mb_size[IS_CHROMA], &c);
    up = c.available; left = a.available; ul = d.available;
    if (p_Vid->p_Inp->UseConstrainedIntraPred) {
      up = c.available ? p_Vid->intra_block[c.mb_addr] : 0; left = a.available ? p_Vid->intra_block[a.mb_addr] : 0; ul = d.available ? p_Vid->intra_block[d.mb_addr] : 0;
    }
    put_i32(f, p_Vid->yuv_format); put_i32(f, up); put_i32(f, left); put_i32(f, ul);
    for (uv = 0; uv < 2; uv++) {
      imgpel **img = p_Vid->enc_picture->imgUV[uv];
      for (i = 0; i < 8; i++) put_i32(f, up ? img[c.pos_y][c.pos_x + i] : 0);
      for (j = 0; j < 16; j++) put_i32(f, left && j < ch ? img[a.pos_y + j][a.pos_x] : 0);
      put_i32(f, ul ? img[d.pos_y][d.pos_x] : 0);
    }
    for (uv = 0; uv < 2; uv++)
      for (k = 0; k < 4; k++) {
        const int written = k == DC_PRED_8 || (k == VERT_PRED_8 && up) || (k == HOR_PRED_8 && left) || (k == PLANE_8 && up && left && ul);
        for (j = 0; j < 16; j++) for (i = 0; i < 8; i++) put_i32(f, written && j < ch ? sl->mpr_16x16[uv + 1][k][j][i] : 0);
      }
    fclose(f); n++;
  }
}

/* ------------------------------------------------------------------ deblocking */
extern void __real_DeblockFrame(VideoParameters *, imgpel **, imgpel ***);
void __wrap_DeblockFrame(VideoParameters *p_Vid, imgpel **imgY, imgpel ***imgUV)
{
  static int n = 0;
  FILE *f = NULL;
  int cw = p_Vid->width_cr, chh = p_Vid->height_cr;
  if (n < 4) {
    unsigned i; int y, x;
    StorablePicture *ids[64]; int nids = 0;
    f = tap_open("deblock.bin");
    put_i32(f, n); put_i32(f, p_Vid->width); put_i32(f, p_Vid->height); put_i32(f, p_Vid->yuv_format);
    put_i32(f, p_Vid->max_pel_value_comp[0]); put_i32(f, p_Vid->max_pel_value_comp[1]);
    put_i32(f, p_Vid->active_sps->direct_8x8_inference_flag);
    put_i32(f, (int)p_Vid->PicSizeInMbs);
    for (i = 0; i < p_Vid->PicSizeInMbs; i++) {
      Macroblock *m = &p_Vid->mb_data[i];
      put_i32(f, m->mb_type); put_i32(f, m->p_Slice->slice_type); put_i32(f, m->qp);
      put_i32(f, m->qpc[0]); put_i32(f, m->qpc[1]); put_i32(f, m->cbp);
      put_i32(f, (int)(m->cbp_blk & 0xFFFF)); put_i32(f, m->slice_nr); put_i32(f, m->DFDisableIdc);
      put_i32(f, m->DFAlphaC0Offset); put_i32(f, m->DFBetaOffset); put_i32(f, m->luma_transform_size_8x8_flag);
    }
    for (y = 0; y < p_Vid->height / 4; y++)
      for (x = 0; x < p_Vid->width / 4; x++) {
        PicMotionParams *mp = &p_Vid->enc_picture->mv_info[y][x];
        int l;
        for (l = 0; l < 2; l++) {
          int id = -1;
          if (mp->ref_idx[l] != -1) {
            int k; for (k = 0; k < nids; k++) if (ids[k] == mp->ref_pic[l]) break;
            if (k == nids && nids < 64) ids[nids++] = mp->ref_pic[l];
            id = k;
          }
          put_i32(f, mp->mv[l].mv_x); put_i32(f, mp->mv[l].mv_y); put_i32(f, id);
        }
      }
    put_plane(f, imgY, 0, 0, p_Vid->height, p_Vid->width);
    if (p_Vid->yuv_format != YUV400) { put_plane(f, imgUV[0], 0, 0, chh, cw); put_plane(f, imgUV[1], 0, 0, chh, cw); }
  }
  __real_DeblockFrame(p_Vid, imgY, imgUV);
  if (f) {
    put_plane(f, imgY, 0, 0, p_Vid->height, p_Vid->width);
    if (p_Vid->yuv_format != YUV400) { put_plane(f, imgUV[0], 0, 0, chh, cw); put_plane(f, imgUV[1], 0, 0, chh, cw); }
    fclose(f);
  }
  n++;
}

/* ------------------------------------------------------------------ transform / quant / reconstruct */
extern void __real_forward4x4(int **, int **, int, int);
void __wrap_forward4x4(int **block, int **tblock, int pos_y, int pos_x)
{
  static int n = 0;
  int in[16], j, i;
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) in[j * 4 + i] = block[pos_y + j][pos_x + i];
  __real_forward4x4(block, tblock, pos_y, pos_x);
  if (n < tap_max() && (n % 7) == 0) {
    FILE *f = tap_open("fwd4x4.bin");
    for (j = 0; j < 16; j++) put_i32(f, in[j]);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) put_i32(f, tblock[pos_y + j][pos_x + i]);
    fclose(f);
  }
  n++;
}
extern void __real_inverse4x4(int **, int **, int, int);
void __wrap_inverse4x4(int **tblock, int **block, int pos_y, int pos_x)
{
  static int n = 0;
  int in[16], j, i;
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) in[j * 4 + i] = tblock[pos_y + j][pos_x + i];
  __real_inverse4x4(tblock, block, pos_y, pos_x);
  if (n < tap_max() && (n % 7) == 0) {
    FILE *f = tap_open("inv4x4.bin");
    for (j = 0; j < 16; j++) put_i32(f, in[j]);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) put_i32(f, block[pos_y + j][pos_x + i]);
    fclose(f);
  }
  n++;
}

static void tap_quant4x4(const char *name, int around, Macroblock *currMB, int **tblock, struct quant_methods *q,
                         int (*real)(Macroblock *, int **, struct quant_methods *), int *ret)
{
  static int n[2] = {0, 0};
  VideoParameters *p_Vid = currMB->p_Vid;
  int in[16], j, i, cost_in = *q->coeff_cost, bx = q->block_x;
  int qp_per = p_Vid->p_Quant->qp_per_matrix[q->qp];
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) in[j * 4 + i] = tblock[j][bx + i];
  *ret = real(currMB, tblock, q);
  if (n[around] < tap_max() && (n[around] % 5) == 0) {
    FILE *f = tap_open(name);
    put_i32(f, q->qp); put_i32(f, qp_per); put_i32(f, currMB->p_Slice->symbol_mode == CAVLC);
    put_i32(f, p_Vid->AdaptRndWeight);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) {
      put_i32(f, q->q_params[j][i].OffsetComp); put_i32(f, q->q_params[j][i].ScaleComp); put_i32(f, q->q_params[j][i].InvScaleComp);
    }
    for (j = 0; j < 16; j++) put_i32(f, in[j]);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) put_i32(f, tblock[j][bx + i]);
    for (j = 0; j < 17; j++) put_i32(f, q->ACLevel[j]);       /* valid up to the 0 terminator */
    for (j = 0; j < 17; j++) put_i32(f, q->ACRun[j]);
    put_i32(f, *q->coeff_cost - cost_in); put_i32(f, *ret);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) put_i32(f, around ? q->fadjust[j][bx + i] : 0);
    fclose(f);
  }
  n[around]++;
}
extern int __real_quant_4x4_normal(Macroblock *, int **, struct quant_methods *);
int __wrap_quant_4x4_normal(Macroblock *currMB, int **tblock, struct quant_methods *q)
{
  int r; tap_quant4x4("quant4x4_normal.bin", 0, currMB, tblock, q, __real_quant_4x4_normal, &r); return r;
}
extern int __real_quant_4x4_around(Macroblock *, int **, struct quant_methods *);
int __wrap_quant_4x4_around(Macroblock *currMB, int **tblock, struct quant_methods *q)
{
  int r; tap_quant4x4("quant4x4_around.bin", 1, currMB, tblock, q, __real_quant_4x4_around, &r); return r;
}

extern void __real_sample_reconstruct(imgpel **, imgpel **, int **, int, int, int, int, int, int);
void __wrap_sample_reconstruct(imgpel **curImg, imgpel **mpr, int **mb_rres, int mb_x, int opix_x,
                               int width, int height, int max_imgpel_value, int dq_bits)
{
  static int n = 0;
  __real_sample_reconstruct(curImg, mpr, mb_rres, mb_x, opix_x, width, height, max_imgpel_value, dq_bits);
  if (n < tap_max() && width == 4 && height == 4 && (n % 5) == 0) {
    FILE *f = tap_open("recon4x4.bin");
    int j, i;
    put_i32(f, max_imgpel_value); put_i32(f, dq_bits);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) put_i32(f, mpr[j][mb_x + i]);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) put_i32(f, mb_rres[j][mb_x + i]);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) put_i32(f, curImg[j][opix_x + i]);
    fclose(f);
  }
  n++;
}

/* ================================================================== 8x8 transform / quantisation and the DC transforms
 *   forward8x8 / inverse8x8                      lcommon/src/transform.c:353 / :450
 *   hadamard4x4 / ihadamard4x4 / hadamard4x2 / ihadamard4x2 / hadamard2x2 / ihadamard2x2   transform.c:121-330
 *   quant_8x8_normal / _around / quant_8x8cavlc_normal / _around   lencod/src/quant8x8_normal.c:43,:123 / quant8x8_around.c:43,:136
 *   quant_dc4x4_normal                           lencod/src/quant4x4_normal.c:200
 *   residual_transform_quant_luma_8x8 / _cavlc   lencod/src/transform8x8.c:522 / :604
 */
#include "quant8x8.h"
#include "transform8x8.h"

extern void __real_forward8x8(int **, int **, int, int);
void __wrap_forward8x8(int **block, int **tblock, int pos_y, int pos_x)
{
  static int n = 0;
  int in[64], j, i;
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) in[j * 8 + i] = block[pos_y + j][pos_x + i];
  __real_forward8x8(block, tblock, pos_y, pos_x);
  if (n < tap_max() && (n % 11) == 0) {
    FILE *f = tap_open("fwd8x8.bin");
    for (j = 0; j < 64; j++) put_i32(f, in[j]);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) put_i32(f, tblock[pos_y + j][pos_x + i]);
    fclose(f);
  }
  n++;
}
extern void __real_inverse8x8(int **, int **, int);
void __wrap_inverse8x8(int **tblock, int **block, int pos_x)
{
  static int n = 0;
  int in[64], j, i;
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) in[j * 8 + i] = tblock[j][pos_x + i];
  __real_inverse8x8(tblock, block, pos_x);
  if (n < tap_max() && (n % 11) == 0) {
    FILE *f = tap_open("inv8x8.bin");
    for (j = 0; j < 64; j++) put_i32(f, in[j]);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) put_i32(f, block[j][pos_x + i]);
    fclose(f);
  }
  n++;
}

#define TAP_DC(NAME, ROWS, FILEN)                                                            \
  extern void __real_##NAME(int **, int **);                                                 \
  void __wrap_##NAME(int **a, int **b)                                                       \
  {                                                                                          \
    static int n = 0;                                                                        \
    int in[16], j, i;                                                                        \
    for (j = 0; j < ROWS; j++) for (i = 0; i < 4; i++) in[j * 4 + i] = a[j][i];              \
    __real_##NAME(a, b);                                                                     \
    if (n < tap_max()) {                                                                     \
      FILE *f = tap_open(FILEN);                                                             \
      for (j = 0; j < ROWS * 4; j++) put_i32(f, in[j]);                                      \
      for (j = 0; j < ROWS; j++) for (i = 0; i < 4; i++) put_i32(f, b[j][i]);                \
      fclose(f);                                                                             \
    }                                                                                        \
    n++;                                                                                     \
  }
TAP_DC(hadamard4x4, 4, "hadamard4x4.bin")
TAP_DC(ihadamard4x4, 4, "ihadamard4x4.bin")
TAP_DC(hadamard4x2, 2, "hadamard4x2.bin")
/* ihadamard4x2 writes its result transposed: block[0..3][0..1] (transform.c:258-298) */
extern void __real_ihadamard4x2(int **, int **);
void __wrap_ihadamard4x2(int **tblock, int **block)
{
  static int n = 0;
  int in[8], j, i;
  for (j = 0; j < 2; j++) for (i = 0; i < 4; i++) in[j * 4 + i] = tblock[j][i];
  __real_ihadamard4x2(tblock, block);
  if (n < tap_max()) {
    FILE *f = tap_open("ihadamard4x2.bin");
    for (j = 0; j < 8; j++) put_i32(f, in[j]);
    for (j = 0; j < 4; j++) for (i = 0; i < 2; i++) put_i32(f, block[j][i]);
    fclose(f);
  }
  n++;
}

extern void __real_hadamard2x2(int **, int *);
void __wrap_hadamard2x2(int **block, int tblock[4])
{
  static int n = 0;
  int in[4] = {block[0][0], block[0][4], block[4][0], block[4][4]}, k;
  __real_hadamard2x2(block, tblock);
  if (n < tap_max()) {
    FILE *f = tap_open("hadamard2x2.bin");
    for (k = 0; k < 4; k++) put_i32(f, in[k]);
    for (k = 0; k < 4; k++) put_i32(f, tblock[k]);
    fclose(f);
  }
  n++;
}
extern void __real_ihadamard2x2(int *, int *);
void __wrap_ihadamard2x2(int tblock[4], int block[4])
{
  static int n = 0;
  int in[4] = {tblock[0], tblock[1], tblock[2], tblock[3]}, k;
  __real_ihadamard2x2(tblock, block);
  if (n < tap_max()) {
    FILE *f = tap_open("ihadamard2x2.bin");
    for (k = 0; k < 4; k++) put_i32(f, in[k]);
    for (k = 0; k < 4; k++) put_i32(f, block[k]);
    fclose(f);
  }
  n++;
}

/* record: qp qp_per AdaptRndWeight variant | 64 x {Offset,Scale,InvScale} | 64 x (i,j) scan | 64 c_cost | in[64] | out[64] |
 *         4 x 17 levels | 4 x 17 runs | cost delta | return | fadjust[64]        (non-CAVLC variants use the first 65 level/run slots) */
static void tap_quant8x8(int variant, Macroblock *currMB, int **tblock, struct quant_methods *q, int ***cofAC, int *ret,
                         int (*real3)(Macroblock *, int **, struct quant_methods *),
                         int (*real4)(Macroblock *, int **, struct quant_methods *, int ***))
{
  static int n[4] = {0, 0, 0, 0};
  VideoParameters *p_Vid = currMB->p_Vid;
  int in[64], j, i, k, cost_in = *q->coeff_cost, bx = q->block_x;
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) in[j * 8 + i] = tblock[j][bx + i];
  *ret = cofAC ? real4(currMB, tblock, q, cofAC) : real3(currMB, tblock, q);
  if (n[variant] < tap_max() / 4 && (n[variant] % 3) == 0) {
    FILE *f = tap_open("quant8x8.bin");
    put_i32(f, q->qp); put_i32(f, p_Vid->p_Quant->qp_per_matrix[q->qp]); put_i32(f, p_Vid->AdaptRndWeight); put_i32(f, variant);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) {
      put_i32(f, q->q_params[j][i].OffsetComp); put_i32(f, q->q_params[j][i].ScaleComp); put_i32(f, q->q_params[j][i].InvScaleComp);
    }
    for (k = 0; k < 64; k++) { put_i32(f, q->pos_scan[k][0]); put_i32(f, q->pos_scan[k][1]); }
    for (k = 0; k < 64; k++) put_i32(f, q->c_cost[k]);
    for (k = 0; k < 64; k++) put_i32(f, in[k]);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) put_i32(f, tblock[j][bx + i]);
    if (cofAC) {
      for (k = 0; k < 4; k++) for (j = 0; j < 17; j++) put_i32(f, cofAC[k][0][j]);
      for (k = 0; k < 4; k++) for (j = 0; j < 17; j++) put_i32(f, cofAC[k][1][j]);
    } else {
      for (j = 0; j < 68; j++) put_i32(f, j < 65 ? q->ACLevel[j] : 0);
      for (j = 0; j < 68; j++) put_i32(f, j < 65 ? q->ACRun[j] : 0);
    }
    put_i32(f, *q->coeff_cost - cost_in); put_i32(f, *ret);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) put_i32(f, (variant & 1) ? q->fadjust[j][bx + i] : 0);
    fclose(f);
  }
  n[variant]++;
}
extern int __real_quant_8x8_normal(Macroblock *, int **, struct quant_methods *);
int __wrap_quant_8x8_normal(Macroblock *m, int **t, struct quant_methods *q) { int r; tap_quant8x8(0, m, t, q, NULL, &r, __real_quant_8x8_normal, NULL); return r; }
extern int __real_quant_8x8_around(Macroblock *, int **, struct quant_methods *);
int __wrap_quant_8x8_around(Macroblock *m, int **t, struct quant_methods *q) { int r; tap_quant8x8(1, m, t, q, NULL, &r, __real_quant_8x8_around, NULL); return r; }
extern int __real_quant_8x8cavlc_normal(Macroblock *, int **, struct quant_methods *, int ***);
int __wrap_quant_8x8cavlc_normal(Macroblock *m, int **t, struct quant_methods *q, int ***c) { int r; tap_quant8x8(2, m, t, q, c, &r, NULL, __real_quant_8x8cavlc_normal); return r; }
extern int __real_quant_8x8cavlc_around(Macroblock *, int **, struct quant_methods *, int ***);
int __wrap_quant_8x8cavlc_around(Macroblock *m, int **t, struct quant_methods *q, int ***c) { int r; tap_quant8x8(3, m, t, q, c, &r, NULL, __real_quant_8x8cavlc_around); return r; }

extern int __real_quant_dc4x4_normal(Macroblock *, int **, int, int *, int *, LevelQuantParams *, const byte (*)[2]);
int __wrap_quant_dc4x4_normal(Macroblock *currMB, int **tblock, int qp, int *DCLevel, int *DCRun, LevelQuantParams *qp44, const byte (*pos_scan)[2])
{
  static int n = 0;
  int in[16], j, i, r;
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) in[j * 4 + i] = tblock[j][i];
  r = __real_quant_dc4x4_normal(currMB, tblock, qp, DCLevel, DCRun, qp44, pos_scan);
  if (n < tap_max()) {
    FILE *f = tap_open("quant_dc4x4.bin");
    put_i32(f, qp); put_i32(f, currMB->p_Vid->p_Quant->qp_per_matrix[qp]); put_i32(f, currMB->p_Slice->symbol_mode == CAVLC);
    put_i32(f, qp44->OffsetComp); put_i32(f, qp44->ScaleComp); put_i32(f, qp44->InvScaleComp);
    for (j = 0; j < 16; j++) put_i32(f, in[j]);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) put_i32(f, tblock[j][i]);
    for (j = 0; j < 17; j++) put_i32(f, DCLevel[j]);
    for (j = 0; j < 17; j++) put_i32(f, DCRun[j]);
    put_i32(f, r);
    fclose(f);
  }
  n++;
  return r;
}

/* record: variant b8 intra qp qp_per AdaptRndWeight AdaptiveRounding max_pel | 64 x q_params | pred[64] | ores[64] |
 *         return | cost delta | rec[64] | 4 x 17 levels | 4 x 17 runs */
static int tap_rtq8x8(int variant, Macroblock *currMB, ColorPlane pl, int b8, int *coeff_cost, int intra,
                      int (*real)(Macroblock *, ColorPlane, int, int *, int))
{
  static int n[2] = {0, 0};
  VideoParameters *p_Vid = currMB->p_Vid;
  Slice *currSlice = currMB->p_Slice;
  const int bx = 8 * (b8 & 1), by = 8 * (b8 >> 1), qp = currMB->qp_scaled[pl], pl_off = b8 + (pl << 2);
  int pred[64], ores[64], j, i, k, r, cost_in = *coeff_cost;
  LevelQuantParams **qpar = p_Vid->p_Quant->q_params_8x8[pl][intra][qp];
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) { pred[j * 8 + i] = currSlice->mb_pred[pl][by + j][bx + i]; ores[j * 8 + i] = currSlice->mb_ores[pl][by + j][bx + i]; }
  r = real(currMB, pl, b8, coeff_cost, intra);
  if (pl == PLANE_Y && n[variant] < tap_max() / 4 && (n[variant] % 3) == 0) {
    FILE *f = tap_open("rtq8x8.bin");
    imgpel **img = p_Vid->enc_picture->p_curr_img;
    put_i32(f, variant); put_i32(f, b8); put_i32(f, intra); put_i32(f, qp); put_i32(f, p_Vid->p_Quant->qp_per_matrix[qp]);
    put_i32(f, p_Vid->AdaptRndWeight); put_i32(f, p_Vid->AdaptiveRounding); put_i32(f, p_Vid->max_imgpel_value);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) { put_i32(f, qpar[j][i].OffsetComp); put_i32(f, qpar[j][i].ScaleComp); put_i32(f, qpar[j][i].InvScaleComp); }
    for (k = 0; k < 64; k++) put_i32(f, pred[k]);
    for (k = 0; k < 64; k++) put_i32(f, ores[k]);
    put_i32(f, r); put_i32(f, *coeff_cost - cost_in);
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) put_i32(f, img[currMB->pix_y + by + j][currMB->pix_x + bx + i]);
    if (variant) {
      for (k = 0; k < 4; k++) for (j = 0; j < 17; j++) put_i32(f, currSlice->cofAC[pl_off][k][0][j]);
      for (k = 0; k < 4; k++) for (j = 0; j < 17; j++) put_i32(f, currSlice->cofAC[pl_off][k][1][j]);
    } else {
      for (j = 0; j < 68; j++) put_i32(f, j < 65 ? currSlice->cofAC[pl_off][0][0][j] : 0);
      for (j = 0; j < 68; j++) put_i32(f, j < 65 ? currSlice->cofAC[pl_off][0][1][j] : 0);
    }
    fclose(f);
  }
  n[variant]++;
  return r;
}
extern int __real_residual_transform_quant_luma_8x8(Macroblock *, ColorPlane, int, int *, int);
int __wrap_residual_transform_quant_luma_8x8(Macroblock *m, ColorPlane pl, int b8, int *cc, int intra) { return tap_rtq8x8(0, m, pl, b8, cc, intra, __real_residual_transform_quant_luma_8x8); }
extern int __real_residual_transform_quant_luma_8x8_cavlc(Macroblock *, ColorPlane, int, int *, int);
int __wrap_residual_transform_quant_luma_8x8_cavlc(Macroblock *m, ColorPlane pl, int b8, int *cc, int intra) { return tap_rtq8x8(1, m, pl, b8, cc, intra, __real_residual_transform_quant_luma_8x8_cavlc); }

/* ================================================================== chroma residual: residual_transform_quant_chroma_4x4 (block.c:954)
 * The slot Macroblock.residual_transform_quant_chroma_4x4[uv] is filled by select_transform() in the translation unit that also
 * defines the function, so there is no link-time reference to wrap: select_transform is wrapped and the slot rebound to the tap.
 * record: uv cr_cbp_in intra yuv cur_qp qp_per_ac qp_per_dc cavlc AdaptRndWeight AdaptiveRounding max_pel cbp_blk_in(lo,hi) |
 *         16 x q_params AC | q_params DC | pred[128] | ores[128] | ret cbp_blk_out(lo,hi) | rec[128] | DC level[9] run[9] |
 *         8 x (AC level[16], run[16]) | fadjust[128]                       (rows of 8 samples; 4:2:0 uses the first 64)      */
extern int residual_transform_quant_chroma_4x4(Macroblock *, int, int);
static int tap_rtq_chroma(Macroblock *currMB, int uv, int cr_cbp)
{
  static int n = 0;
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currSlice->p_Vid;
  const int yuv = p_Vid->yuv_format, H = p_Vid->mb_cr_size_y, intra = is_intra(currMB);
  const int cur_qp = currMB->qpc[uv] + currSlice->bitdepth_chroma_qp_scale, qp_dc = yuv == YUV422 ? cur_qp + 3 : cur_qp;
  const int uv_scale = uv * (p_Vid->num_blk8x8_uv >> 1);
  int pred[128], ores[128], j, i, k, r;
  int64 cbp_in = currMB->cbp_blk;
  LevelQuantParams **qa = p_Vid->p_Quant->q_params_4x4[uv + 1][intra][cur_qp], *qd = &p_Vid->p_Quant->q_params_4x4[uv + 1][intra][qp_dc][0][0];
  int **fadj = NULL;
  if (yuv != YUV420 && yuv != YUV422) return residual_transform_quant_chroma_4x4(currMB, uv, cr_cbp);
  memset(pred, 0, sizeof pred); memset(ores, 0, sizeof ores);
  for (j = 0; j < H; j++) for (i = 0; i < 8; i++) { pred[j * 8 + i] = currSlice->mb_pred[uv + 1][j][i]; ores[j * 8 + i] = currSlice->mb_ores[uv + 1][j][i]; }
  r = residual_transform_quant_chroma_4x4(currMB, uv, cr_cbp);
  if (n < tap_max() / 2 && (n % 3) == 0) {
    FILE *f = tap_open("rtq_chroma.bin");
    put_i32(f, uv); put_i32(f, cr_cbp); put_i32(f, intra); put_i32(f, yuv); put_i32(f, cur_qp);
    put_i32(f, p_Vid->p_Quant->qp_per_matrix[cur_qp]); put_i32(f, p_Vid->p_Quant->qp_per_matrix[qp_dc]);
    put_i32(f, currSlice->symbol_mode == CAVLC); put_i32(f, p_Vid->AdaptRndWeight); put_i32(f, p_Vid->AdaptiveRounding); put_i32(f, p_Vid->max_pel_value_comp[uv + 1]);
    put_i32(f, (int)(cbp_in & 0xffffffff)); put_i32(f, (int)(cbp_in >> 32));
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) { put_i32(f, qa[j][i].OffsetComp); put_i32(f, qa[j][i].ScaleComp); put_i32(f, qa[j][i].InvScaleComp); }
    put_i32(f, qd->OffsetComp); put_i32(f, qd->ScaleComp); put_i32(f, qd->InvScaleComp);
    for (k = 0; k < 128; k++) put_i32(f, pred[k]);
    for (k = 0; k < 128; k++) put_i32(f, ores[k]);
    put_i32(f, r); put_i32(f, (int)(currMB->cbp_blk & 0xffffffff)); put_i32(f, (int)(currMB->cbp_blk >> 32));
    for (j = 0; j < 16; j++) for (i = 0; i < 8; i++) put_i32(f, j < H ? p_Vid->enc_picture->imgUV[uv][currMB->pix_c_y + j][currMB->pix_c_x + i] : 0);
    for (k = 0; k < 9; k++) put_i32(f, currSlice->cofDC[uv + 1][0][k]);
    for (k = 0; k < 9; k++) put_i32(f, currSlice->cofDC[uv + 1][1][k]);
    for (k = 0; k < 8; k++) {
      const int b8 = k >> 2, b4 = k & 3, live = b8 < (p_Vid->num_blk8x8_uv >> 1);
      for (j = 0; j < 16; j++) put_i32(f, live ? currSlice->cofAC[4 + b8 + uv_scale][b4][0][j] : 0);
      for (j = 0; j < 16; j++) put_i32(f, live ? currSlice->cofAC[4 + b8 + uv_scale][b4][1][j] : 0);
    }
    if (p_Vid->AdaptiveRounding)
      fadj = (currMB->mb_type == P8x8 && currMB->luma_transform_size_8x8_flag) ? p_Vid->ARCofAdj4x4[uv + 1][4] : p_Vid->ARCofAdj4x4[uv + 1][currMB->ar_mode];
    for (j = 0; j < 16; j++) for (i = 0; i < 8; i++) put_i32(f, (fadj && j < H) ? fadj[j][i] : 0);
    fclose(f);
  }
  n++;
  return r;
}
/* ================================================================== Intra16x16 luma: residual_transform_quant_luma_16x16 (block.c:208)
 * (slot rebound through select_transform, as above)
 * record: pl qp qp_per cavlc AdaptiveRounding AdaptRndWeight max_pel i16mode | 16 x q_params (intra) | orig[256] | pred[256] |
 *         ret | DC level[17] run[17] | 16 x (AC level[16], run[16]) in (b8, b4) order | rec[256] | fadjust rows 0..3 [4][16] after the call */
extern int residual_transform_quant_luma_16x16(Macroblock *, ColorPlane);
static int tap_rtq_luma_16x16(Macroblock *currMB, ColorPlane pl)
{
  static int n = 0;
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currSlice->p_Vid;
  int orig[256], pred[256], j, i, k, b, r;
  const int mode = currMB->i16mode, qp = currMB->qp_scaled[pl];
  if (pl != PLANE_Y || p_Vid->yuv_format == YUV444) return residual_transform_quant_luma_16x16(currMB, pl);
  for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) {
    orig[j * 16 + i] = p_Vid->pCurImg[currMB->opix_y + j][currMB->pix_x + i];
    pred[j * 16 + i] = currSlice->mpr_16x16[pl][mode][j][i];
  }
  r = residual_transform_quant_luma_16x16(currMB, pl);
  if (n < tap_max() / 8 && (n % 2) == 0) {
    LevelQuantParams **q = p_Vid->p_Quant->q_params_4x4[pl][1][qp];
    FILE *f = tap_open("rtq16x16.bin");
    put_i32(f, pl); put_i32(f, qp); put_i32(f, p_Vid->p_Quant->qp_per_matrix[qp]); put_i32(f, currSlice->symbol_mode == CAVLC);
    put_i32(f, p_Vid->AdaptiveRounding); put_i32(f, p_Vid->AdaptRndWeight); put_i32(f, p_Vid->max_imgpel_value); put_i32(f, mode);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) { put_i32(f, q[j][i].OffsetComp); put_i32(f, q[j][i].ScaleComp); put_i32(f, q[j][i].InvScaleComp); }
    for (k = 0; k < 256; k++) put_i32(f, orig[k]);
    for (k = 0; k < 256; k++) put_i32(f, pred[k]);
    put_i32(f, r);
    for (k = 0; k < 17; k++) put_i32(f, currSlice->cofDC[pl][0][k]);
    for (k = 0; k < 17; k++) put_i32(f, currSlice->cofDC[pl][1][k]);
    for (b = 0; b < 16; b++) {
      for (k = 0; k < 16; k++) put_i32(f, currSlice->cofAC[(pl << 2) + (b >> 2)][b & 3][0][k]);
      for (k = 0; k < 16; k++) put_i32(f, currSlice->cofAC[(pl << 2) + (b >> 2)][b & 3][1][k]);
    }
    for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) put_i32(f, p_Vid->enc_picture->p_curr_img[currMB->pix_y + j][currMB->pix_x + i]);
    for (j = 0; j < 4; j++) for (i = 0; i < 16; i++) put_i32(f, p_Vid->AdaptiveRounding ? p_Vid->ARCofAdj4x4[pl][I16MB][j][i] : 0);
    fclose(f);
  }
  n++;
  return r;
}
extern void __real_select_transform(Macroblock *);
void __wrap_select_transform(Macroblock *currMB)
{
  __real_select_transform(currMB);
  if (currMB->residual_transform_quant_luma_16x16 == residual_transform_quant_luma_16x16) currMB->residual_transform_quant_luma_16x16 = tap_rtq_luma_16x16;
  if (currMB->residual_transform_quant_chroma_4x4[0] == residual_transform_quant_chroma_4x4) currMB->residual_transform_quant_chroma_4x4[0] = tap_rtq_chroma;
  if (currMB->residual_transform_quant_chroma_4x4[1] == residual_transform_quant_chroma_4x4) currMB->residual_transform_quant_chroma_4x4[1] = tap_rtq_chroma;
}

/* ------------------------------------------------------------------ weighted / bi-predictive candidate distortions: call counts
 *   compute{SAD,SATD,SSE}WP, computeBiPred{SAD,SATD,SSE}1 / 2     lencod/src/me_distortion.c:434-1530
 * Which of them a configuration reaches, and with which weights (the distinct (w1, w2, offset, denominator) tuples, first 64), written
 * to dist_calls.txt at exit.  Their arithmetic is pinned by direct calls (oracle/ref_call.c); this tap tells which configurations are
 * worth running end to end. */
#include "me_distortion.h"
static long g_dist_calls[9];
static int g_wp_seen[64][5], g_wp_n;
static void dist_report(void)
{
  static const char *nm[9] = {"computeSADWP", "computeSATDWP", "computeSSEWP", "computeBiPredSAD1", "computeBiPredSATD1", "computeBiPredSSE1",
                              "computeBiPredSAD2", "computeBiPredSATD2", "computeBiPredSSE2"};
  FILE *f = tap_open("dist_calls.txt");
  int k;
  for (k = 0; k < 9; k++) fprintf(f, "%s %ld\n", nm[k], g_dist_calls[k]);
  for (k = 0; k < g_wp_n; k++) fprintf(f, "weights kind=%d w1=%d w2=%d offset=%d log_denom=%d\n", g_wp_seen[k][0], g_wp_seen[k][1], g_wp_seen[k][2], g_wp_seen[k][3], g_wp_seen[k][4]);
  fclose(f);
}
static void dist_count(int which, MEBlock *mb, int bi)
{
  static int hooked = 0;
  int t[5], k;
  if (!hooked) { hooked = 1; atexit(dist_report); }
  g_dist_calls[which]++;
  t[0] = bi; t[1] = bi ? mb->weight1 : mb->weight_luma; t[2] = bi ? mb->weight2 : 0; t[3] = bi ? mb->offsetBi : mb->offset_luma;
  t[4] = mb->p_Slice->luma_log_weight_denom;
  if (bi == 2) return;                                                        /* the un-weighted average reads no weights */
  for (k = 0; k < g_wp_n; k++) if (!memcmp(g_wp_seen[k], t, sizeof t)) return;
  if (g_wp_n < 64) memcpy(g_wp_seen[g_wp_n++], t, sizeof t);
}
#define TAP_UNI(fn, idx) \
  extern distblk __real_##fn(StorablePicture *, MEBlock *, distblk, MotionVector *); \
  distblk __wrap_##fn(StorablePicture *r, MEBlock *mb, distblk m, MotionVector *c) { dist_count(idx, mb, 0); return __real_##fn(r, mb, m, c); }
#define TAP_BI(fn, idx, kind) \
  extern distblk __real_##fn(StorablePicture *, StorablePicture *, MEBlock *, distblk, MotionVector *, MotionVector *); \
  distblk __wrap_##fn(StorablePicture *r1, StorablePicture *r2, MEBlock *mb, distblk m, MotionVector *c1, MotionVector *c2) \
  { dist_count(idx, mb, kind); return __real_##fn(r1, r2, mb, m, c1, c2); }
TAP_UNI(computeSADWP, 0) TAP_UNI(computeSATDWP, 1) TAP_UNI(computeSSEWP, 2)
TAP_BI(computeBiPredSAD1, 3, 2) TAP_BI(computeBiPredSATD1, 4, 2) TAP_BI(computeBiPredSSE1, 5, 2)
TAP_BI(computeBiPredSAD2, 6, 1) TAP_BI(computeBiPredSATD2, 7, 1) TAP_BI(computeBiPredSSE2, 8, 1)
