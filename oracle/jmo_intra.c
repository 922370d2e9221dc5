/* jmo_intra.c -- TEST INFRASTRUCTURE (oracle): plain-C restatement of JM 19.0's luma intra prediction and of its Intra16x16 mode search.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 *   get_intrapred_4x4      lencod/src/intra4x4.c:521-561 (nine modes :72-308) over the predictor samples set_intrapred_4x4 :421-519 gathers:
 *                          e[0] = X (above left), e[1..8] = A..H (above, above right), e[9..12] = I..L (left)
 *   get_intrapred_16x16    lencod/src/intra16x16.c:307-328 (four modes :28-146): e[0] = above left, e[1..16] above, e[17..32] left
 *   find_sad_16x16_JM      :463-517 with distI16x16_sad / _sse / _satd :331-452
 * Written from the prediction formulas of H.264 8.3.1.2 / 8.3.3 (which JM's per-sample assignments implement); pinned by the records of
 * the real encoder's calls in tests/golden/qcif_intra.npz. */
#include <string.h>
#include "jmo.h"

static int top4(const jmo_pel *e, int x) { return x < 0 ? e[0] : e[1 + x]; }       /* p[x, -1], x = -1 .. 7 */
static int left4(const jmo_pel *e, int y) { return y < 0 ? e[0] : e[9 + y]; }      /* p[-1, y], y = -1 .. 3 */

void jmo_intrapred_4x4(const jmo_pel e[13], int mode, int left_available, int up_available, jmo_pel out[16])
{
  int x, y;
  for (y = 0; y < 4; y++)
    for (x = 0; x < 4; x++) {
      int v = 0;
      switch (mode) {
      case 0: v = top4(e, x); break;                                                /* vertical */
      case 1: v = left4(e, y); break;                                               /* horizontal */
      case 2:                                                                       /* DC */
        if (up_available && left_available) v = (e[1] + e[2] + e[3] + e[4] + e[9] + e[10] + e[11] + e[12] + 4) >> 3;
        else if (left_available) v = (e[9] + e[10] + e[11] + e[12] + 2) >> 2;
        else if (up_available) v = (e[1] + e[2] + e[3] + e[4] + 2) >> 2;
        else v = e[1];                                                              /* already the DC value (set_intrapred_4x4) */
        break;
      case 3:                                                                       /* diagonal down left */
        v = (x == 3 && y == 3) ? (top4(e, 6) + 3 * top4(e, 7) + 2) >> 2 : (top4(e, x + y) + 2 * top4(e, x + y + 1) + top4(e, x + y + 2) + 2) >> 2;
        break;
      case 4:                                                                       /* diagonal down right */
        if (x > y) v = (top4(e, x - y - 2) + 2 * top4(e, x - y - 1) + top4(e, x - y) + 2) >> 2;
        else if (x < y) v = (left4(e, y - x - 2) + 2 * left4(e, y - x - 1) + left4(e, y - x) + 2) >> 2;
        else v = (top4(e, 0) + 2 * e[0] + left4(e, 0) + 2) >> 2;
        break;
      case 5: {                                                                     /* vertical right */
        const int z = 2 * x - y, k = x - (y >> 1);
        if (z >= 0 && !(z & 1)) v = (top4(e, k - 1) + top4(e, k) + 1) >> 1;
        else if (z > 0) v = (top4(e, k - 2) + 2 * top4(e, k - 1) + top4(e, k) + 2) >> 2;
        else if (z == -1) v = (left4(e, 0) + 2 * e[0] + top4(e, 0) + 2) >> 2;
        else v = (left4(e, y - 1) + 2 * left4(e, y - 2) + left4(e, y - 3) + 2) >> 2;
        break; }
      case 6: {                                                                     /* horizontal down */
        const int z = 2 * y - x, k = y - (x >> 1);
        if (z >= 0 && !(z & 1)) v = (left4(e, k - 1) + left4(e, k) + 1) >> 1;
        else if (z > 0) v = (left4(e, k - 2) + 2 * left4(e, k - 1) + left4(e, k) + 2) >> 2;
        else if (z == -1) v = (left4(e, 0) + 2 * e[0] + top4(e, 0) + 2) >> 2;
        else v = (top4(e, x - 1) + 2 * top4(e, x - 2) + top4(e, x - 3) + 2) >> 2;
        break; }
      case 7: {                                                                     /* vertical left */
        const int k = x + (y >> 1);
        v = (y & 1) ? (top4(e, k) + 2 * top4(e, k + 1) + top4(e, k + 2) + 2) >> 2 : (top4(e, k) + top4(e, k + 1) + 1) >> 1;
        break; }
      default: {                                                                    /* 8: horizontal up */
        const int z = x + 2 * y, k = y + (x >> 1);
        if (z > 5) v = left4(e, 3);
        else if (z == 5) v = (left4(e, 2) + 3 * left4(e, 3) + 2) >> 2;
        else if (z & 1) v = (left4(e, k) + 2 * left4(e, k + 1) + left4(e, k + 2) + 2) >> 2;
        else v = (left4(e, k) + left4(e, k + 1) + 1) >> 1;
        break; }
      }
      out[4 * y + x] = (jmo_pel)v;
    }
}

void jmo_intrapred_16x16(const jmo_pel e[33], int mode, int left_available, int up_available, int max_pel, jmo_pel out[256])
{
  int x, y, dc = 0, a = 0, b = 0, c = 0;
  if (mode == 2) {
    int s1 = 0, s2 = 0;
    for (x = 0; x < 16; x++) { s1 += e[1 + x]; s2 += e[17 + x]; }
    if (up_available && left_available) dc = (s1 + s2 + 16) >> 5;
    else if (up_available) dc = (s1 + 8) >> 4;
    else if (left_available) dc = (s2 + 8) >> 4;
    else dc = e[1];
  } else if (mode == 3) {
    int H = 0, V = 0;
    for (x = 0; x < 8; x++) {                                                       /* p[-1, -1] stands in for index -1 */
      H += (x + 1) * (e[1 + 8 + x] - (x == 7 ? e[0] : e[1 + 6 - x]));
      V += (x + 1) * (e[17 + 8 + x] - (x == 7 ? e[0] : e[17 + 6 - x]));
    }
    b = (5 * H + 32) >> 6; c = (5 * V + 32) >> 6; a = 16 * (e[16] + e[32]);
  }
  for (y = 0; y < 16; y++)
    for (x = 0; x < 16; x++) {
      int v;
      if (mode == 0) v = e[1 + x];
      else if (mode == 1) v = e[17 + y];
      else if (mode == 2) v = dc;
      else { v = (a + b * (x - 7) + c * (y - 7) + 16) >> 5; v = v < 0 ? 0 : (v > max_pel ? max_pel : v); }
      out[16 * y + x] = (jmo_pel)v;
    }
}

/* Slice.distI16x16 of a 16x16 prediction: metric 0 SAD, 1 SSE, 2 (and anything else) SATD; the value JM reaches without early exit, << 5 */
jmo_dist jmo_dist_i16x16(const jmo_pel orig[256], const jmo_pel pred[256], int metric)
{
  int64_t cost = 0;
  int j, i, jj, ii;
  if (metric == 0 || metric == 1) {
    for (j = 0; j < 256; j++) { const int d = (int)orig[j] - (int)pred[j]; cost += metric == 0 ? (d < 0 ? -d : d) : d * d; }
    return cost << 5;
  }
  {
    int dc[16], hd[16];
    for (jj = 0; jj < 4; jj++)
      for (ii = 0; ii < 4; ii++) {
        int in[16], out[16];
        for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) in[4*j+i] = (int)orig[(4*jj+j)*16 + 4*ii+i] - (int)pred[(4*jj+j)*16 + 4*ii+i];
        jmo_hadamard4x4(in, out);                                                   /* JM's hadamard4x4: second pass >> 1 */
        for (j = 1; j < 16; j++) cost += out[j] < 0 ? -out[j] : out[j];
        dc[jj * 4 + ii] = out[0] >> 1;
      }
    jmo_hadamard4x4(dc, hd);
    for (j = 0; j < 16; j++) cost += hd[j] < 0 ? -hd[j] : hd[j];
  }
  return cost << 5;
}

/* find_sad_16x16_JM: modes of mode_mask in ascending order, strict '<' against the running best (DISTBLK_MAX first); pred4 receives the
 * predictions of the evaluated modes.  Returns the best cost, *best_mode as JM leaves currMB->i16mode (DC when nothing was evaluated). */
jmo_dist jmo_intra16_search(const jmo_pel e[33], int left_available, int up_available, int mode_mask, int metric, int max_pel,
                            const jmo_pel orig[256], jmo_pel pred4[4][256], int *best_mode)
{
  jmo_dist best = JMO_DIST_MAX;
  int k;
  *best_mode = 2;
  for (k = 0; k < 4; k++)
    if ((mode_mask >> k) & 1) {
      jmo_dist c;
      jmo_intrapred_16x16(e, k, left_available, up_available, max_pel, pred4[k]);
      c = jmo_dist_i16x16(orig, pred4[k], metric);
      if (c < best) { best = c; *best_mode = k; }
    }
  return best;
}

/* intra_chroma_prediction, lencod/src/intra_chroma.c:530-778 (frame macroblocks): the four chroma prediction modes of one plane.
 *   up[cw] = image[pix_c.pos_y][pix_c.pos_x ...], left[ch] = image[pix_a.pos_y ...][pix_a.pos_x], corner = image[pix_d.pos_y][pix_d.pos_x]
 *   cw = 8, ch = 8 (4:2:0) or 16 (4:2:2);  pred[mode][j * 8 + i], modes DC_PRED_8 0, HOR_PRED_8 1, VERT_PRED_8 2, PLANE_8 3 (defines.h:280-283)
 * DC: per 4x4 block, by the block's place (:590-686): top-left both sums, top-right the upper sum first, bottom-left the left sum first,
 * bottom-right both; 128 (dc_pred_value) without neighbours.  Vertical / horizontal only with their neighbour, plane only with all three
 * (:690-747; the modes without their neighbours are left alone: zeros here).  Returns the mask of the modes written. */
int jmo_intra_chroma_pred(const jmo_pel *up, const jmo_pel *left, int corner, int up_avail, int left_avail, int upleft_avail,
                          int ch, int max_pel, jmo_pel pred[4][128])
{
  const int cw = 8;
  int bx, by, i, j, mask = 1;
  memset(pred, 0, 4 * 128 * sizeof(jmo_pel));
  for (by = 0; by < ch; by += 4)
    for (bx = 0; bx < cw; bx += 4) {
      int su = 0, sl = 0, s = (max_pel + 1) >> 1;
      for (i = 0; i < 4; i++) { su += up_avail ? up[bx + i] : 0; sl += left_avail ? left[by + i] : 0; }
      if ((by == 0) == (bx == 0)) {                                 /* top-left, bottom-right: both sums when both are there */
        if (up_avail && left_avail) s = (su + sl + 4) >> 3;
        else if (up_avail) s = (su + 2) >> 2;
        else if (left_avail) s = (sl + 2) >> 2;
      } else if (by == 0) {                                         /* top-right: the samples above first */
        if (up_avail) s = (su + 2) >> 2;
        else if (left_avail) s = (sl + 2) >> 2;
      } else {                                                      /* bottom-left: the samples to the left first */
        if (left_avail) s = (sl + 2) >> 2;
        else if (up_avail) s = (su + 2) >> 2;
      }
      for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) pred[0][(by + j) * 8 + bx + i] = (jmo_pel)s;
    }
  if (up_avail) { mask |= 4; for (j = 0; j < ch; j++) for (i = 0; i < cw; i++) pred[2][j * 8 + i] = up[i]; }
  if (left_avail) { mask |= 2; for (j = 0; j < ch; j++) for (i = 0; i < cw; i++) pred[1][j * 8 + i] = left[j]; }
  if (left_avail && up_avail && upleft_avail) {
    const int cr_x = cw >> 1, cr_y = ch >> 1;
    int ih = cr_x * ((int)up[cw - 1] - corner), iv = cr_y * ((int)left[ch - 1] - corner), ib, ic, iaa;
    for (i = 0; i < cr_x - 1; i++) ih += (i + 1) * ((int)up[cr_x + i] - (int)up[cr_x - 2 - i]);
    for (i = 0; i < cr_y - 1; i++) iv += (i + 1) * ((int)left[cr_y + i] - (int)left[cr_y - 2 - i]);
    ib = (17 * ih + 2 * cw) >> 5;                                   /* cr_MB_x == 8 */
    ic = ch == 8 ? (17 * iv + 2 * ch) >> 5 : (5 * iv + 2 * ch) >> 6;
    iaa = 16 * ((int)up[cw - 1] + (int)left[ch - 1]) + (1 - cr_x) * ib + (1 - cr_y) * ic;
    for (j = 0; j < ch; j++)
      for (i = 0; i < cw; i++) {
        int v = (iaa + i * ib + j * ic + 16) >> 5;
        pred[3][j * 8 + i] = (jmo_pel)(v < 0 ? 0 : (v > max_pel ? max_pel : v));
      }
    mask |= 8;
  }
  return mask;
}

/* get_intrapred_8x8, lencod/src/intra8x8.c:716-760 (the nine modes :148-495): 8x8 luma intra prediction from the 25 predictor samples
 * currMB->intra8x8_pred[pl] = Z, A..P (16 above incl. above-right), Q..X (8 left) AFTER LowPassForIntra8x8Pred (:85-140, applied by
 * set_intrapred_8x8 :497-601).  H.264 8.3.2.2.2-8.3.2.2.10 with p'[x,-1] = top8(x), p'[-1,y] = left8(y); JM's per-sample assignments implement them. */
static int top8(const jmo_pel *e, int x) { return x < 0 ? e[0] : e[1 + x]; }        /* p'[x, -1], x = -1 .. 15 */
static int left8(const jmo_pel *e, int y) { return y < 0 ? e[0] : e[17 + y]; }      /* p'[-1, y], y = -1 .. 7 */
void jmo_intrapred_8x8(const jmo_pel e[25], int mode, int left_available, int up_available, jmo_pel out[64])
{
  int x, y, i;
  for (y = 0; y < 8; y++)
    for (x = 0; x < 8; x++) {
      int v = 0;
      switch (mode) {
      case 0: v = top8(e, x); break;
      case 1: v = left8(e, y); break;
      case 2: {
        int su = 0, sl = 0;
        for (i = 0; i < 8; i++) { su += e[1 + i]; sl += e[17 + i]; }
        if (up_available && left_available) v = (su + sl + 8) >> 4;
        else if (left_available) v = (sl + 4) >> 3;
        else if (up_available) v = (su + 4) >> 3;
        else v = e[1];                                                              /* P_A: holds dc_pred_value then (:216) */
        break; }
      case 3: v = (x == 7 && y == 7) ? (top8(e, 14) + 3 * top8(e, 15) + 2) >> 2 : (top8(e, x + y) + 2 * top8(e, x + y + 1) + top8(e, x + y + 2) + 2) >> 2; break;
      case 4:
        if (x > y) v = (top8(e, x - y - 2) + 2 * top8(e, x - y - 1) + top8(e, x - y) + 2) >> 2;
        else if (x < y) v = (left8(e, y - x - 2) + 2 * left8(e, y - x - 1) + left8(e, y - x) + 2) >> 2;
        else v = (top8(e, 0) + 2 * e[0] + left8(e, 0) + 2) >> 2;
        break;
      case 5: {
        const int z = 2 * x - y, k = x - (y >> 1);
        if (z >= 0 && !(z & 1)) v = (top8(e, k - 1) + top8(e, k) + 1) >> 1;
        else if (z > 0) v = (top8(e, k - 2) + 2 * top8(e, k - 1) + top8(e, k) + 2) >> 2;
        else if (z == -1) v = (left8(e, 0) + 2 * e[0] + top8(e, 0) + 2) >> 2;
        else v = (left8(e, y - 2 * x - 1) + 2 * left8(e, y - 2 * x - 2) + left8(e, y - 2 * x - 3) + 2) >> 2;
        break; }
      case 6: {
        const int z = 2 * y - x, k = y - (x >> 1);
        if (z >= 0 && !(z & 1)) v = (left8(e, k - 1) + left8(e, k) + 1) >> 1;
        else if (z > 0) v = (left8(e, k - 2) + 2 * left8(e, k - 1) + left8(e, k) + 2) >> 2;
        else if (z == -1) v = (left8(e, 0) + 2 * e[0] + top8(e, 0) + 2) >> 2;
        else v = (top8(e, x - 2 * y - 1) + 2 * top8(e, x - 2 * y - 2) + top8(e, x - 2 * y - 3) + 2) >> 2;
        break; }
      case 7: {
        const int k = x + (y >> 1);
        v = (y & 1) ? (top8(e, k) + 2 * top8(e, k + 1) + top8(e, k + 2) + 2) >> 2 : (top8(e, k) + top8(e, k + 1) + 1) >> 1;
        break; }
      default: {
        const int z = x + 2 * y, k = y + (x >> 1);
        if (z > 13) v = left8(e, 7);
        else if (z == 13) v = (left8(e, 6) + 3 * left8(e, 7) + 2) >> 2;
        else if (z & 1) v = (left8(e, k) + 2 * left8(e, k + 1) + left8(e, k + 2) + 2) >> 2;
        else v = (left8(e, k) + left8(e, k + 1) + 1) >> 1;
        break; }
      }
      out[y * 8 + x] = (jmo_pel)v;
    }
}
