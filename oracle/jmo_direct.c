/*
 * oracle/jmo_direct.c -- TEST INFRASTRUCTURE (parity oracle, see jmo.h).
 *
 * CPU restatement of the spatial direct mode of B slices for frame macroblocks (the first building block of the B-picture path):
 *
 *   Get_Direct_MV_Spatial_Normal                 lencod/src/mv_direct.c:522-685
 *   set_direct_references :330, get_colocated_info :428 / get_colocated_info_4x4 :492 (frame pictures, frame_mbs_only: the plain branch)
 *   GetMotionVectorPredictorNormal               lcommon/src/mv_prediction.c:194-325 for the 16x16 block of list 0 / list 1
 *
 * Scope: no weighted bi-prediction (test_valid_direct is not restated: weighted_bipred_idc 1 is rejected by the caller), no MBAFF, no fields.
 * Pinned by tests/golden/direct_b.npz: per-macroblock dumps of the real encoder's B slices (oracle/ref_tap_mb.c, tap_b_slice).
 */
#include <stdint.h>
#include "jmo.h"

static int iabs_(int x) { return x < 0 ? -x : x; }
static int imedian(int a, int b, int c) { return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b)); }

/* the 16x16 predictor of one list: neighbours A, B, C (C already replaced by D where get_neighbors does that) */
static void pred16(const int8_t avail[3], const int8_t nref[3][2], const int16_t nmv[3][2][2], int list, int ref, int16_t out[2])
{
  int r[3], k, type = 0;
  int mv[3][2];
  for (k = 0; k < 3; k++) {
    r[k] = avail[k] ? nref[k][list] : -1;
    mv[k][0] = avail[k] ? nmv[k][list][0] : 0; mv[k][1] = avail[k] ? nmv[k][list][1] : 0;
  }
  if (r[0] == ref && r[1] != ref && r[2] != ref) type = 1;
  else if (r[0] != ref && r[1] == ref && r[2] != ref) type = 2;
  else if (r[0] != ref && r[1] != ref && r[2] == ref) type = 3;
  if (type == 0) {
    if (!(avail[1] || avail[2])) { out[0] = (int16_t)mv[0][0]; out[1] = (int16_t)mv[0][1]; return; }
    out[0] = (int16_t)imedian(mv[0][0], mv[1][0], mv[2][0]); out[1] = (int16_t)imedian(mv[0][1], mv[1][1], mv[2][1]);
    return;
  }
  out[0] = (int16_t)mv[type - 1][0]; out[1] = (int16_t)mv[type - 1][1];
}

/* avail / nref / nmv: the neighbours A, B, C.  col_ref / col_mv: per 4x4 block of the macroblock (raster) what listX[LIST_1][0]->mv_info holds at the position
 * get_colocated_info reads for it.  Out, per 4x4 block: direct_ref_idx[2], direct_pdir, and the vector of each list (zero where the list is not used). */
void jmo_direct_spatial(const int8_t avail[3], const int8_t nref[3][2], const int16_t nmv[3][2][2], int col_long_term,
                        const int8_t col_ref[16][2], const int16_t col_mv[16][2][2],
                        int8_t ref_out[16][2], int8_t pdir_out[16], int16_t mv_out[16][2][2])
{
  int refX[2], l, b;
  int16_t pmv[2][2] = {{0, 0}, {0, 0}};
  for (l = 0; l < 2; l++) {
    /* imin over unsigned char: -1 (no reference / not available) is the largest value */
    unsigned m = 255;
    int k;
    for (k = 0; k < 3; k++) { const unsigned v = (unsigned)(uint8_t)(avail[k] ? nref[k][l] : -1); if (v < m) m = v; }
    refX[l] = (int)(int8_t)(uint8_t)m;
    if (refX[l] >= 0) pred16(avail, nref, nmv, l, refX[l], pmv[l]);
  }
  for (b = 0; b < 16; b++) {
    int8_t *ri = ref_out[b];
    mv_out[b][0][0] = mv_out[b][0][1] = mv_out[b][1][0] = mv_out[b][1][1] = 0;
    if (refX[0] == -1 && refX[1] == -1) { ri[0] = ri[1] = 0; pdir_out[b] = 2; continue; }
    if (refX[0] == 0 || refX[1] == 0) {
      int moving = 1;                                         /* get_colocated_info: 1 = the co-located block moves */
      if (!col_long_term)
        moving = !((col_ref[b][0] == 0 && (iabs_(col_mv[b][0][0]) >> 1) == 0 && (iabs_(col_mv[b][0][1]) >> 1) == 0) ||
                   (col_ref[b][0] == -1 && col_ref[b][1] == 0 && (iabs_(col_mv[b][1][0]) >> 1) == 0 && (iabs_(col_mv[b][1][1]) >> 1) == 0));
      for (l = 0; l < 2; l++) {
        if (refX[l] < 0) ri[l] = -1;
        else if (refX[l] == 0 && !moving) ri[l] = 0;          /* JM's is_moving_block is "the co-located block does NOT move" (mv_direct.c:593) */
        else { ri[l] = (int8_t)refX[l]; mv_out[b][l][0] = pmv[l][0]; mv_out[b][l][1] = pmv[l][1]; }
      }
    } else {
      for (l = 0; l < 2; l++) {
        if (refX[l] > 0) { ri[l] = (int8_t)refX[l]; mv_out[b][l][0] = pmv[l][0]; mv_out[b][l][1] = pmv[l][1]; }
        else ri[l] = -1;
      }
    }
    pdir_out[b] = ri[1] == -1 ? 0 : (ri[0] == -1 ? 1 : 2);
  }
}
