/*
 * oracle/ref_tap_mb.c -- TEST INFRASTRUCTURE: a tap on the REAL reference encoder's macroblock mode decision.
 *
 * Linked (oracle/Makefile.ref, target `tapmb`) with the unmodified JM 19.0 lencod objects using GNU ld's --wrap, it lets the
 * reference's own encode_one_macroblock_low (lencod/src/md_low.c:104) run and then dumps what that call left behind for
 * write_macroblock (lencod/src/macroblock.c:2810): macroblock type, sub-modes, vectors, reference indices, coded block pattern,
 * intra modes, the coefficient (level, run) lists and the reconstructed samples -- plus the motion costs of every searched
 * partition, which localise a divergence to one search.  tests/golden/make_mb_golden.py turns the dump into the committed
 * fixtures that pin oracle/jmo_mbenc.c.
 *
 * Own code against JM's public headers; no reference source.  Builds only where /root/reference is present, only into oracle/_ref/.
 * Output: $JM_TAP_DIR/mb_low.bin (default "."), fixed-size little-endian records (layout: MBREC below, mirrored in make_mb_golden.py).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "global.h"
#include "image.h"
#include "mbuffer.h"
#include "mv_search.h"
#include "macroblock.h"

#pragma pack(push, 1)
typedef struct {
  int32_t frame_no, mb_addr, slice_type, slice_nr;
  int32_t best_mode, mb_type, cbp, c_ipred_mode, i16mode, i16offset, transform8x8, qp;
  int32_t lambda_mf[3], lambda_mdfp, num_ref, max_mvd, mv_limit[4], qpc, search_range;   /* the slice's parameters, repeated per record */
  int64_t cbp_blk, min_rdcost;
  int8_t  b8mode[4], b8pdir[4];
  int8_t  ipred_syntax[16];                 /* currMB->intra_pred_modes[4*b8+b4] */
  int8_t  ipredmode[16];                    /* p_Vid->ipredmode, 4x4 raster */
  int16_t mv[16][2];                        /* enc_picture->mv_info[..].mv[LIST_0], 4x4 raster */
  int8_t  ref_idx[16];
  int64_t motion_cost[8][4];                /* p_Vid->motion_cost[mode][LIST_0][0][block] (reference 0) */
  int16_t all_mv[8][16][2];                 /* currSlice->all_mv[LIST_0][0][mode][by][bx] */
  int32_t luma_level[16][17], luma_run[16][17];      /* cofAC[b8][b4][0/1][k], b = 4*b8+b4 */
  int32_t dc_level[3][18], dc_run[3][18];            /* cofDC[pl][0/1][k] */
  int32_t chroma_level[8][17], chroma_run[8][17];    /* cofAC[4 + uv*2.. ][b4] for 4:2:0: b8 = 4, 5 hold U, V? see make_mb_golden.py */
  uint8_t rec_y[256], rec_u[64], rec_v[64];
  int32_t poc, ref_poc[16];                 /* enc_picture->poc, listX[LIST_0][r]->poc (EPZS scales its predictors by picture distances) */
  int64_t motion_cost_ref[8][4][4];         /* p_Vid->motion_cost[mode][LIST_0][ref 1..4][block] */
  int32_t luma8_level[4][65], luma8_run[4][65];      /* cofAC[b8][0][0/1][k]: the 64-entry list of an 8x8 transform block with CABAC (CAVLC: four lists of 16 in luma_level) */
  /* 4:2:2: chroma_level / _run [uv * 4 + b4] = cofAC[4 + 2 uv][b4] (the plane's blocks 0..3), chroma2_* [uv * 4 + b4] = cofAC[5 + 2 uv][b4] (blocks 4..7, block.c:1096-1105);
   * rec_u / rec_v rows 0..7, rec_u2 / rec_v2 rows 8..15 */
  int32_t chroma2_level[8][17], chroma2_run[8][17];
  uint8_t rec_u2[64], rec_v2[64];
  int32_t yuv_format;
  int32_t qpc_v;                            /* currMB->qpc[1] (qpc above is qpc[0]) */
  /* B slices (zero otherwise) */
  int16_t mv1[16][2];                       /* enc_picture->mv_info[..].mv[LIST_1] */
  int8_t  ref_idx1[16];
  int8_t  b8bipred[4];                      /* currMB->b8x8[k].bipred */
  int32_t num_ref1;                         /* listXsize[LIST_1] */
  int64_t motion_cost1[8][2][4];            /* p_Vid->motion_cost[mode][LIST_1][ref 0..1][block] */
  int32_t poc_l1[4];                        /* listX[LIST_1][r]->poc */
  int32_t direct_8x8_inference, pad_;
} MBREC;
#pragma pack(pop)

/* B slices: what Get_Direct_MV_Spatial_Normal (lencod/src/mv_direct.c:522) read and what it left -- $JM_TAP_DIR/mb_low_b.bin, one record per macroblock of a B slice.
 * The neighbours A, B, C (C replaced by D where get_neighbors does that) are earlier macroblocks: their mv_info is final when the current macroblock's direct vectors are made,
 * and still is when this tap runs.  Layout mirrored in tests/golden/make_direct_b.py. */
typedef struct {
  int32_t frame_no, mb_addr, slice_nr, direct_8x8_inference, weighted_bipred_idc, num_ref[2], col_long_term;
  int8_t  nb_avail[4];                      /* get_neighbors(currMB, mb, 0, 0, 16): A, B, C (after the replacement), D */
  int8_t  nb_ref[3][2];                     /* mv_info[..].ref_idx[list] of A, B, C (valid where available) */
  int16_t nb_mv[3][2][2];                   /* ... .mv[list] {x, y} */
  int8_t  direct_ref_idx[16][2], direct_pdir[16];        /* currSlice->direct_ref_idx / direct_pdir, 4x4 raster of the macroblock */
  int16_t direct_mv[16][2][2];              /* currSlice->all_mv[list][max(direct_ref_idx, 0)][0][by][bx] */
  int8_t  col_ref[16][2];                   /* listX[LIST_1][0]->mv_info at the co-located position get_colocated_info reads (RSD corners with direct_8x8_inference) */
  int16_t col_mv[16][2][2];
} MBREC_B;

static void tap_b_slice(Macroblock *currMB)
{
  static FILE *f = NULL;
  VideoParameters *p_Vid = currMB->p_Vid;
  Slice *currSlice = currMB->p_Slice;
  PicMotionParams **motion = p_Vid->enc_picture->mv_info;
  StorablePicture *col = currSlice->listX[LIST_1][0];
  const int inf = p_Vid->active_sps->direct_8x8_inference_flag;
  PixelPos mb[4];
  MBREC_B r;
  int k, l, i, j;
  if (!f) {
    char path[1024];
    const char *d = getenv("JM_TAP_DIR");
    snprintf(path, sizeof path, "%s/mb_low_b.bin", d ? d : ".");
    f = fopen(path, "wb");
    if (!f) { perror(path); exit(3); }
  }
  memset(&r, 0, sizeof r);
  r.frame_no = p_Vid->frame_no; r.mb_addr = currMB->mbAddrX; r.slice_nr = currMB->slice_nr; r.direct_8x8_inference = inf;
  r.weighted_bipred_idc = p_Vid->active_pps->weighted_bipred_idc; r.num_ref[0] = currSlice->listXsize[LIST_0]; r.num_ref[1] = currSlice->listXsize[LIST_1];
  r.col_long_term = col->is_long_term;
  get_neighbors(currMB, mb, 0, 0, 16);
  for (k = 0; k < 4; k++) r.nb_avail[k] = (int8_t)mb[k].available;
  for (k = 0; k < 3; k++)
    for (l = 0; l < 2; l++) {
      r.nb_ref[k][l] = -1;
      if (mb[k].available) {
        const PicMotionParams *mp = &motion[mb[k].pos_y][mb[k].pos_x];
        r.nb_ref[k][l] = mp->ref_idx[l]; r.nb_mv[k][l][0] = mp->mv[l].mv_x; r.nb_mv[k][l][1] = mp->mv[l].mv_y;
      }
    }
  for (j = 0; j < 4; j++)
    for (i = 0; i < 4; i++) {
      const int by = currMB->block_y + j, bx = currMB->block_x + i, b = j * 4 + i;
      const int cy = inf ? RSD(by) : by, cx = inf ? RSD(bx) : bx;
      r.direct_pdir[b] = currSlice->direct_pdir[by][bx];
      for (l = 0; l < 2; l++) {
        const int ref = currSlice->direct_ref_idx[by][bx][l];
        r.direct_ref_idx[b][l] = (int8_t)ref;
        r.direct_mv[b][l][0] = currSlice->all_mv[l][ref < 0 ? 0 : ref][0][j][i].mv_x; r.direct_mv[b][l][1] = currSlice->all_mv[l][ref < 0 ? 0 : ref][0][j][i].mv_y;
        r.col_ref[b][l] = col->mv_info[cy][cx].ref_idx[l];
        r.col_mv[b][l][0] = col->mv_info[cy][cx].mv[l].mv_x; r.col_mv[b][l][1] = col->mv_info[cy][cx].mv[l].mv_y;
      }
    }
  fwrite(&r, sizeof r, 1, f);
  fflush(f);
}

extern void __real_encode_one_macroblock_low(Macroblock *currMB);
void __wrap_encode_one_macroblock_low(Macroblock *currMB)
{
  static FILE *f = NULL;
  static int n = 0, maxn = -1;
  VideoParameters *p_Vid = currMB->p_Vid;
  Slice *currSlice = currMB->p_Slice;
  PicMotionParams **motion = p_Vid->enc_picture->mv_info;
  MBREC r;
  int i, j, k, b8, b4, m;

  __real_encode_one_macroblock_low(currMB);
  if (currSlice->slice_type == B_SLICE && getenv("JM_TAPMB_B")) tap_b_slice(currMB);

  if (maxn < 0) { const char *e = getenv("JM_TAPMB_MAX"); maxn = e ? atoi(e) : 1 << 30; }
  if (n >= maxn) return;
  if (!f) {
    char path[1024];
    const char *d = getenv("JM_TAP_DIR");
    snprintf(path, sizeof path, "%s/mb_low.bin", d ? d : ".");
    f = fopen(path, "wb");
    if (!f) { perror(path); exit(3); }
  }
  memset(&r, 0, sizeof r);
  r.frame_no = p_Vid->frame_no; r.mb_addr = currMB->mbAddrX; r.slice_type = currSlice->slice_type; r.slice_nr = currMB->slice_nr;
  r.best_mode = currMB->best_mode; r.mb_type = currMB->mb_type; r.cbp = currMB->cbp; r.c_ipred_mode = currMB->c_ipred_mode;
  r.i16mode = currMB->i16mode; r.i16offset = currMB->i16offset; r.transform8x8 = currMB->luma_transform_size_8x8_flag; r.qp = currMB->qp;
  r.cbp_blk = currMB->cbp_blk; r.min_rdcost = currMB->min_rdcost;
  for (i = 0; i < 3; i++) r.lambda_mf[i] = p_Vid->lambda_mf[currSlice->slice_type][p_Vid->masterQP][i];
  r.lambda_mdfp = LAMBDA_FACTOR(p_Vid->lambda_md[currSlice->slice_type][p_Vid->masterQP]);
  r.num_ref = currSlice->listXsize[LIST_0]; r.max_mvd = p_Vid->max_mvd; r.qpc = currMB->qpc[0]; r.search_range = p_Vid->searchRange.max_x >> 2;
  r.mv_limit[0] = p_Vid->MaxHmvR[4]; r.mv_limit[1] = p_Vid->MaxHmvR[5]; r.mv_limit[2] = p_Vid->MaxVmvR[4]; r.mv_limit[3] = p_Vid->MaxVmvR[5];
  for (i = 0; i < 4; i++) { r.b8mode[i] = currMB->b8x8[i].mode; r.b8pdir[i] = currMB->b8x8[i].pdir; }
  memcpy(r.ipred_syntax, currMB->intra_pred_modes, 16);
  for (j = 0; j < 4; j++)
    for (i = 0; i < 4; i++) {
      PicMotionParams *mp = &motion[currMB->block_y + j][currMB->block_x + i];
      r.ipredmode[j * 4 + i] = p_Vid->ipredmode[currMB->block_y + j][currMB->block_x + i];
      r.mv[j * 4 + i][0] = mp->mv[LIST_0].mv_x; r.mv[j * 4 + i][1] = mp->mv[LIST_0].mv_y;
      r.ref_idx[j * 4 + i] = mp->ref_idx[LIST_0];
    }
  if (currSlice->slice_type != I_SLICE) {
    for (m = 1; m < 8; m++) {
      for (k = 0; k < 4; k++) r.motion_cost[m][k] = p_Vid->motion_cost[m][LIST_0][0][k];
      for (j = 0; j < 4; j++)
        for (i = 0; i < 4; i++) {
          r.all_mv[m][j * 4 + i][0] = currSlice->all_mv[LIST_0][0][m][j][i].mv_x;
          r.all_mv[m][j * 4 + i][1] = currSlice->all_mv[LIST_0][0][m][j][i].mv_y;
        }
    }
  }
  r.poc = p_Vid->enc_picture->poc;
  if (currSlice->slice_type != I_SLICE) {
    for (i = 0; i < 16 && i < currSlice->listXsize[LIST_0]; i++) r.ref_poc[i] = currSlice->listX[LIST_0][i]->poc;
    for (m = 1; m < 8; m++)
      for (i = 1; i < 5 && i < currSlice->listXsize[LIST_0]; i++)
        for (k = 0; k < 4; k++) r.motion_cost_ref[m][i - 1][k] = p_Vid->motion_cost[m][LIST_0][i][k];
  }
  for (b8 = 0; b8 < 4; b8++)
    for (b4 = 0; b4 < 4; b4++)
      for (k = 0; k < 17; k++) {
        r.luma_level[b8 * 4 + b4][k] = currSlice->cofAC[b8][b4][0][k];
        r.luma_run[b8 * 4 + b4][k] = currSlice->cofAC[b8][b4][1][k];
      }
  for (b8 = 0; b8 < 4; b8++)
    for (k = 0; k < 65; k++) { r.luma8_level[b8][k] = currSlice->cofAC[b8][0][0][k]; r.luma8_run[b8][k] = currSlice->cofAC[b8][0][1][k]; }
  for (m = 0; m < 3; m++)
    for (k = 0; k < 18; k++) { r.dc_level[m][k] = currSlice->cofDC[m][0][k]; r.dc_run[m][k] = currSlice->cofDC[m][1][k]; }
  if (p_Vid->yuv_format == YUV420)
    for (b8 = 4; b8 < 6; b8++)                   /* 4:2:0: cofAC[4] = U, cofAC[5] = V (block.c:1107 `b8 = 4 + uv`) */
      for (b4 = 0; b4 < 4; b4++)
        for (k = 0; k < 17; k++) {
          r.chroma_level[(b8 - 4) * 4 + b4][k] = currSlice->cofAC[b8][b4][0][k];
          r.chroma_run[(b8 - 4) * 4 + b4][k] = currSlice->cofAC[b8][b4][1][k];
        }
  if (p_Vid->yuv_format == YUV422)
    for (m = 0; m < 2; m++)                      /* plane */
      for (b4 = 0; b4 < 4; b4++)
        for (k = 0; k < 17; k++) {
          r.chroma_level[m * 4 + b4][k] = currSlice->cofAC[4 + 2 * m][b4][0][k];
          r.chroma_run[m * 4 + b4][k] = currSlice->cofAC[4 + 2 * m][b4][1][k];
          r.chroma2_level[m * 4 + b4][k] = currSlice->cofAC[5 + 2 * m][b4][0][k];
          r.chroma2_run[m * 4 + b4][k] = currSlice->cofAC[5 + 2 * m][b4][1][k];
        }
  r.yuv_format = p_Vid->yuv_format; r.qpc_v = currMB->qpc[1];
  r.direct_8x8_inference = p_Vid->active_sps->direct_8x8_inference_flag;
  if (currSlice->slice_type == B_SLICE) {
    r.num_ref1 = currSlice->listXsize[LIST_1];
    for (i = 0; i < 4; i++) r.b8bipred[i] = currMB->b8x8[i].bipred;
    for (j = 0; j < 4; j++)
      for (i = 0; i < 4; i++) {
        PicMotionParams *mp = &motion[currMB->block_y + j][currMB->block_x + i];
        r.mv1[j * 4 + i][0] = mp->mv[LIST_1].mv_x; r.mv1[j * 4 + i][1] = mp->mv[LIST_1].mv_y; r.ref_idx1[j * 4 + i] = mp->ref_idx[LIST_1];
      }
    for (m = 1; m < 8; m++)
      for (i = 0; i < 2 && i < currSlice->listXsize[LIST_1]; i++)
        for (k = 0; k < 4; k++) r.motion_cost1[m][i][k] = p_Vid->motion_cost[m][LIST_1][i][k];
    for (i = 0; i < 4 && i < currSlice->listXsize[LIST_1]; i++) r.poc_l1[i] = currSlice->listX[LIST_1][i]->poc;
  }
  for (j = 0; j < 16; j++)
    for (i = 0; i < 16; i++) r.rec_y[j * 16 + i] = (uint8_t)p_Vid->enc_picture->imgY[currMB->pix_y + j][currMB->pix_x + i];
  if (p_Vid->yuv_format == YUV420)
    for (j = 0; j < 8; j++)
      for (i = 0; i < 8; i++) {
        r.rec_u[j * 8 + i] = (uint8_t)p_Vid->enc_picture->imgUV[0][currMB->pix_c_y + j][currMB->pix_c_x + i];
        r.rec_v[j * 8 + i] = (uint8_t)p_Vid->enc_picture->imgUV[1][currMB->pix_c_y + j][currMB->pix_c_x + i];
      }
  if (p_Vid->yuv_format == YUV422)
    for (j = 0; j < 8; j++)
      for (i = 0; i < 8; i++) {
        r.rec_u[j * 8 + i] = (uint8_t)p_Vid->enc_picture->imgUV[0][currMB->pix_c_y + j][currMB->pix_c_x + i];
        r.rec_v[j * 8 + i] = (uint8_t)p_Vid->enc_picture->imgUV[1][currMB->pix_c_y + j][currMB->pix_c_x + i];
        r.rec_u2[j * 8 + i] = (uint8_t)p_Vid->enc_picture->imgUV[0][currMB->pix_c_y + 8 + j][currMB->pix_c_x + i];
        r.rec_v2[j * 8 + i] = (uint8_t)p_Vid->enc_picture->imgUV[1][currMB->pix_c_y + 8 + j][currMB->pix_c_x + i];
      }
  fwrite(&r, sizeof r, 1, f);
  fflush(f);
  n++;
}
