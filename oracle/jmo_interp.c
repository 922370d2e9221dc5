/*
 * oracle/jmo_interp.c -- TEST INFRASTRUCTURE (parity oracle, see jmo.h).
 * CPU restatement of getSubImagesLuma (lencod/src/img_luma.c:611-679): the 16
 * quarter-pel luma planes of a reference picture.
 *
 * Restated in padded coordinates (x in [0,Wp), y in [0,Hp), Wp = W+2*32, Hp = H+2*20).
 * Every border special case in img_luma.c (:169-232 left/right columns, :274-289 /
 * :312-329 top/bottom rows, :480-497 / :519-546 / :565-603 last column/row of the
 * bilinear planes) is an index clamp into the padded plane, so one clamped stencil
 * reproduces them all.
 */
#include <stdlib.h>
#include "jmo.h"

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int clip1(int hi, int x) { return x < 0 ? 0 : (x > hi ? hi : x); }
static inline int tap6(int c, int b, int a, int d, int e, int f)      /* ONE_FOURTH_TAP[0] = {20,-5,1}, img_luma.h:21-25 */
{
  return 20 * (a + d) - 5 * (b + e) + (c + f);
}

void jmo_sub_images_luma(const jmo_pel *src, int src_pitch, int W, int H, int max_pel,
                         jmo_pel *dst, int pitch, long plane_stride)
{
  const int Wp = W + 2 * JMO_PAD_X, Hp = H + 2 * JMO_PAD_Y;
  int x, y;
  int *tmp = (int *)malloc(sizeof(int) * (size_t)Wp * Hp);            /* imgY_sub_tmp */
#define P(j, i) (dst + ((j) * 4 + (i)) * plane_stride)
#define AT(pl, yy, xx) (pl)[(long)(yy) * pitch + (xx)]
  jmo_pel *p00 = P(0, 0), *p02 = P(0, 2), *p20 = P(2, 0), *p22 = P(2, 2);

  /* [0][0]: edge-replicated copy, getSubImageInteger :40-86 */
  for (y = 0; y < Hp; y++) {
    int sy = clampi(y - JMO_PAD_Y, 0, H - 1);
    for (x = 0; x < Wp; x++)
      AT(p00, y, x) = src[(long)sy * src_pitch + clampi(x - JMO_PAD_X, 0, W - 1)];
  }
  /* [0][2]: horizontal six-tap, keeps the unclipped sum; getHorSubImageSixTap :151-236 */
  for (y = 0; y < Hp; y++)
    for (x = 0; x < Wp; x++) {
      int s = tap6(AT(p00, y, clampi(x - 2, 0, Wp - 1)), AT(p00, y, clampi(x - 1, 0, Wp - 1)), AT(p00, y, x),
                   AT(p00, y, clampi(x + 1, 0, Wp - 1)), AT(p00, y, clampi(x + 2, 0, Wp - 1)), AT(p00, y, clampi(x + 3, 0, Wp - 1)));
      tmp[(long)y * Wp + x] = s;
      AT(p02, y, x) = (jmo_pel)clip1(max_pel, (s + 16) >> 5);
    }
  /* [2][0]: vertical six-tap on [0][0]; getVerSubImageSixTap :257-331 */
  for (y = 0; y < Hp; y++) {
    int ym2 = clampi(y - 2, 0, Hp - 1), ym1 = clampi(y - 1, 0, Hp - 1);
    int yp1 = clampi(y + 1, 0, Hp - 1), yp2 = clampi(y + 2, 0, Hp - 1), yp3 = clampi(y + 3, 0, Hp - 1);
    for (x = 0; x < Wp; x++) {
      int s = tap6(AT(p00, ym2, x), AT(p00, ym1, x), AT(p00, y, x), AT(p00, yp1, x), AT(p00, yp2, x), AT(p00, yp3, x));
      AT(p20, y, x) = (jmo_pel)clip1(max_pel, (s + 16) >> 5);
      /* [2][2]: vertical six-tap on the unclipped horizontal sums; getVerSubImageSixTapTmp :347-423 */
      s = tap6(tmp[(long)ym2 * Wp + x], tmp[(long)ym1 * Wp + x], tmp[(long)y * Wp + x],
               tmp[(long)yp1 * Wp + x], tmp[(long)yp2 * Wp + x], tmp[(long)yp3 * Wp + x]);
      AT(p22, y, x) = (jmo_pel)clip1(max_pel, (s + 512) >> 10);
    }
  }
  /* quarter-pel planes: rounded averages, getSubImagesLuma :653-678 */
  for (y = 0; y < Hp; y++) {
    int y1 = clampi(y + 1, 0, Hp - 1);
    for (x = 0; x < Wp; x++) {
      int x1 = clampi(x + 1, 0, Wp - 1);
#define AVG(a, b) (jmo_pel)(((int)(a) + (int)(b) + 1) >> 1)
      AT(P(0, 1), y, x) = AVG(AT(p00, y, x), AT(p02, y, x));
      AT(P(1, 0), y, x) = AVG(AT(p00, y, x), AT(p20, y, x));
      AT(P(1, 1), y, x) = AVG(AT(p02, y, x), AT(p20, y, x));
      AT(P(1, 2), y, x) = AVG(AT(p02, y, x), AT(p22, y, x));
      AT(P(2, 1), y, x) = AVG(AT(p20, y, x), AT(p22, y, x));
      AT(P(0, 3), y, x) = AVG(AT(p02, y, x), AT(p00, y, x1));       /* getHorSubImageBiLinear :477-500 */
      AT(P(1, 3), y, x) = AVG(AT(p02, y, x), AT(p20, y, x1));
      AT(P(2, 3), y, x) = AVG(AT(p22, y, x), AT(p20, y, x1));
      AT(P(3, 0), y, x) = AVG(AT(p20, y, x), AT(p00, y1, x));       /* getVerSubImageBiLinear :519-550 */
      AT(P(3, 1), y, x) = AVG(AT(p20, y, x), AT(p02, y1, x));
      AT(P(3, 2), y, x) = AVG(AT(p22, y, x), AT(p02, y1, x));
      AT(P(3, 3), y, x) = AVG(AT(p02, y1, x), AT(p20, y, x1));      /* getDiagSubImageBiLinear :569-603 */
#undef AVG
    }
  }
#undef AT
#undef P
  free(tmp);
}

/* A frame as it lies in the source file (planar Y, U, V, 8-bit, source size == output size) -> the coded-size planes:
 *   buf2img_basic    lcommon/src/input.c:552-600  one byte per sample widened to imgpel, straight copy when the sizes agree
 *   pad_borders      lcommon/src/input.c:880-925  right border: every sample repeats its left neighbour; bottom border: every row repeats the row above
 * (read_one_frame :792-868 hands the three planes of p_Vid->buf to buf2img; image.c:1243-1244 pads).  yuv: 0 4:0:0, 1 4:2:0, 2 4:2:2. */
static void load_plane(const uint8_t *raw, int sw, int sh, int W, int H, jmo_pel *out)
{
  int x, y;
  for (y = 0; y < sh; y++) {
    for (x = 0; x < sw; x++) out[(long)y * W + x] = raw[(long)y * sw + x];
    for (x = sw; x < W; x++) out[(long)y * W + x] = out[(long)y * W + x - 1];
  }
  for (y = sh; y < H; y++) for (x = 0; x < W; x++) out[(long)y * W + x] = out[(long)(y - 1) * W + x];
}
void jmo_load_frame(const uint8_t *raw, int src_w, int src_h, int W, int H, int yuv, jmo_pel *y, jmo_pel *u, jmo_pel *v)
{
  const int scw = src_w >> 1, sch = yuv == 1 ? src_h >> 1 : src_h, cw = W >> 1, ch = yuv == 1 ? H >> 1 : H;
  load_plane(raw, src_w, src_h, W, H, y);
  if (yuv == 1 || yuv == 2) {
    load_plane(raw + (long)src_w * src_h, scw, sch, cw, ch, u);
    load_plane(raw + (long)src_w * src_h + (long)scw * sch, scw, sch, cw, ch, v);
  }
}
