/* jmo_mc.c -- TEST INFRASTRUCTURE (oracle): plain-C restatement of JM 19.0's motion-compensated prediction of one block,
 * the un-weighted paths.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 *   luma_prediction                           lencod/src/mc_prediction.c:144-236
 *   OneComponentLumaPrediction                :122-136   (one UMVLine4X origin, block_size_x samples per row, padded pitch)
 *   mc_prediction / bi_prediction             :100-110 / :82-93   ((a + b + 1) >> 1)
 *   chroma_prediction_4x4                     :568-650
 *   OneComponentChromaPrediction4x4_retrieve  :361-411   (ChromaMCBuffer = 1: two samples from the chroma sub-image of the
 *                                             vector's phase, UMVLine8X_chroma origin clamp, lencod/inc/refbuf.h:61-65)
 *   getSubImagesChroma / generateChroma*      lencod/src/img_chroma.c:26-437: sub-image (k, l) at (Y, X) =
 *                                             (w00 S[Y][X] + w01 S[Y][X+1] + w10 S[Y+1][X] + w11 S[Y+1][X+1] + 32) >> 6 with
 *                                             every coordinate clamped into the picture (that is what its edge code amounts to)
 * Pinned by tests/golden/qcif_mc.npz (records of the real encoder's calls, tests/test_oracle_golden.py). */
#include "jmo.h"

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

static void one_luma(const jmo_refpic *r, int qx, int qy, int bsx, int bsy, jmo_pel *out)
{
  /* UMVLine4X: plane by the two low bits, origin clamped to [-PAD, size + 2 PAD - 1 - 16 - PAD] (mbuffer.c:564-565) */
  const int yy = clip3(-JMO_PAD_Y, r->height + JMO_PAD_Y - 1 - 16, qy >> 2), xx = clip3(-JMO_PAD_X, r->width + JMO_PAD_X - 1 - 16, qx >> 2);
  const jmo_pel *line = r->sub[qy & 3][qx & 3] + (long)yy * r->pitch + xx;
  int j, i;
  for (j = 0; j < bsy; j++, line += r->pitch)
    for (i = 0; i < bsx; i++) out[j * bsx + i] = line[i];
}

void jmo_luma_pred(const jmo_refpic *r0, const jmo_refpic *r1, int p_dir, int x, int y, int bsx, int bsy, jmo_mv mv0, jmo_mv mv1, jmo_pel *out)
{
  jmo_pel a[256], b[256];
  int k;
  if (p_dir != 1) one_luma(r0, (x << 2) + mv0.x, (y << 2) + mv0.y, bsx, bsy, a);
  if (p_dir != 0) one_luma(r1, (x << 2) + mv1.x, (y << 2) + mv1.y, bsx, bsy, b);
  for (k = 0; k < bsx * bsy; k++) out[k] = p_dir == 0 ? a[k] : (p_dir == 1 ? b[k] : (jmo_pel)((a[k] + b[k] + 1) >> 1));
}

/* one sample of the chroma sub-image of phase (jj & my, ii & mx) at integer position (Y, X) of a W x H plane */
static int chroma_sub(const jmo_pel *pl, int pitch, int W, int H, int yuv, int ph_y, int ph_x, int Y, int X)
{
  /* weights of getSubImagesChroma: l = ph_x * mul_x, k = ph_y * mul_y in eighths (4:2:0 and the x direction: mul 1; 4:2:2 y: mul 2) */
  const int l = ph_x, k = ph_y * (yuv == 2 ? 2 : 1), m = 8 - k;
  const int w01 = m * l, w00 = (m << 3) - w01, w11 = k * l, w10 = (k << 3) - w11;
  const int y0 = clip3(0, H - 1, Y), y1 = clip3(0, H - 1, Y + 1), x0 = clip3(0, W - 1, X), x1 = clip3(0, W - 1, X + 1);
  return (w00 * pl[y0 * pitch + x0] + w01 * pl[y0 * pitch + x1] + w10 * pl[y1 * pitch + x0] + w11 * pl[y1 * pitch + x1] + 32) >> 6;
}

/* one list of chroma_prediction_4x4: mv[row][pair] = the vector of sample row j, sample pair (0,1) / (2,3) */
static void one_chroma(const jmo_pel *pl, int pitch, int W, int H, int yuv, int xc, int yc, const jmo_mv mv[4][2], jmo_pel *out)
{
  const int sx = 3, sy = yuv == 2 ? 2 : 3;                       /* chroma_shift_x / _y: vector units per chroma sample = 1 << shift */
  const int mx = 7, my = yuv == 2 ? 3 : 7;                       /* chroma_mask_mv_x / _y, lencod.c:2366-2378 */
  const int pad_x = JMO_PAD_X >> 1, pad_y = yuv == 2 ? JMO_PAD_Y : JMO_PAD_Y >> 1;
  const int mbw = 8, mbh = yuv == 2 ? 16 : 8;
  const int max_x = W - 1 + pad_x - mbw, max_y = H - 1 + pad_y - mbh;   /* size_x_cr_pad, size_y_cr_pad, mbuffer.c:568-569 */
  int j, h, o;
  for (j = 0; j < 4; j++)
    for (h = 0; h < 2; h++) {
      const int ii = ((xc + 2 * h) << sx) + mv[j][h].x, jj = ((yc + j) << sy) + mv[j][h].y;
      const int X = clip3(-pad_x, max_x, ii >> sx), Y = clip3(-pad_y, max_y, jj >> sy);
      for (o = 0; o < 2; o++) out[j * 4 + 2 * h + o] = (jmo_pel)chroma_sub(pl, pitch, W, H, yuv, jj & my, ii & mx, Y, X + o);
    }
}

void jmo_chroma_pred4x4(const jmo_pel *p0, const jmo_pel *p1, int pitch, int W, int H, int yuv, int p_dir, int xc, int yc,
                        const jmo_mv mv0[4][2], const jmo_mv mv1[4][2], jmo_pel out[16])
{
  jmo_pel a[16], b[16];
  int k;
  if (p_dir != 1) one_chroma(p0, pitch, W, H, yuv, xc, yc, mv0, a);
  if (p_dir != 0) one_chroma(p1, pitch, W, H, yuv, xc, yc, mv1, b);
  for (k = 0; k < 16; k++) out[k] = p_dir == 0 ? a[k] : (p_dir == 1 ? b[k] : (jmo_pel)((a[k] + b[k] + 1) >> 1));
}

/* getSubImagesChroma (lencod/src/img_chroma.c:338-437) of one plane: every sub-image with its padding.
 * dst: [suby][subx][H + 2 pad_y][W + 2 pad_x], suby < 8 (4:2:0) or 4 (4:2:2), subx < 8; pads (IMG_PAD_SIZE >> 1 each at 4:2:0,
 * IMG_PAD_SIZE_Y rows at 4:2:2; lencod.c:2366-2376). */
void jmo_sub_images_chroma(const jmo_pel *src, int pitch, int W, int H, int yuv, jmo_pel *dst)
{
  const int ny = yuv == 2 ? 4 : 8, pad_x = JMO_PAD_X >> 1, pad_y = yuv == 2 ? JMO_PAD_Y : JMO_PAD_Y >> 1;
  const int Wp = W + 2 * pad_x, Hp = H + 2 * pad_y;
  int sy, sx, Y, X;
  for (sy = 0; sy < ny; sy++)
    for (sx = 0; sx < 8; sx++) {
      jmo_pel *d = dst + (long)(sy * 8 + sx) * Wp * Hp;
      for (Y = -pad_y; Y < H + pad_y; Y++)
        for (X = -pad_x; X < W + pad_x; X++) d[(long)(Y + pad_y) * Wp + X + pad_x] = (jmo_pel)chroma_sub(src, pitch, W, H, yuv, sy, sx, Y, X);
    }
}

/* weighted_mc_prediction / weighted_bi_prediction, lencod/src/mc_prediction.c:38-73, applied to the per-list predictions p0 / p1 of n samples
 * (luma_prediction :203-228 and chroma_prediction_4x4 :615-640 hand them over with the parameters documented at jmo_wp):
 *   p_dir 0 / 1  clip1(((weight[p_dir] * p + round) >> shift) + offset)        p_dir 2  clip1(((w0 * p0 + w1 * p1 + round) >> shift) + offset) */
void jmo_weighted_samples(const jmo_pel *p0, const jmo_pel *p1, int n, int p_dir, const jmo_wp *wp, int max_pel, jmo_pel *out)
{
  int i;
  for (i = 0; i < n; i++) {
    int v;
    if (p_dir == 2) v = ((wp->weight[0] * p0[i] + wp->weight[1] * p1[i] + wp->round) >> wp->shift) + wp->offset;
    else v = ((wp->weight[p_dir] * (p_dir ? p1[i] : p0[i]) + wp->round) >> wp->shift) + wp->offset;
    out[i] = (jmo_pel)(v < 0 ? 0 : (v > max_pel ? max_pel : v));
  }
}
