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
#include <string.h>
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

/* ---- the general source picture reader: read_one_frame's buf2img calls (lcommon/src/input.c:822-853, planar Y, U, V) with buf2img_basic (:552-650) when the
 * source and output bit depths agree and buf2img_bitshift (:440-540, rshift_rnd: rounding shift down, plain shift up) otherwise (initInput :41-53), little endian
 * samples of one or two bytes, the centred / cropped copy when the file's size differs from the picture's, then pad_borders (:880-925).
 * Planes: tight, coded size (W x H luma; chroma by yuv: 0 none, 1 4:2:0, 2 4:2:2, 3 4:4:4); what the copy does not reach stays 0, as in freshly allocated planes. */
static void load_plane_ex(const uint8_t *buf, int w, int h, int ow, int oh, int cw, int ch, int sb, int shift, int bitshift_fn, jmo_pel *out)
{
  const int same = w == ow && h == oh;
  const int iw = same ? ow : (w < ow ? w : ow), ih = same ? oh : (h < oh ? h : oh);
  const int dx = (!same && ow >= w) ? (ow - w) >> 1 : 0, dy = (!same && oh >= h) ? (oh - h) >> 1 : 0;
  int x, y;
  memset(out, 0, (size_t)cw * ch * sizeof(jmo_pel));
  if (!bitshift_fn && sb == (int)sizeof(uint16_t) && same) {
    /* buf2img_basic :568-570: imgpel-sized samples of an equal-sized picture are copied with ONE memcpy into &imgX[0][0] -- w * h samples back to back, although
     * the plane's rows are the CODED width apart: a picture whose width is not a multiple of 16 arrives sheared.  That is what the reference encodes. */
    for (x = 0; x < w * h; x++) out[x] = (jmo_pel)(buf[2 * x] | (buf[2 * x + 1] << 8));
  } else
  for (y = 0; y < ih; y++)
    for (x = 0; x < iw; x++) {
      /* buf2img_basic :586-588 (imgpel-sized samples, sizes differ): row y is taken at temp_buf[y * size_x] -- an unsigned char pointer, so the row starts y * size_x BYTES
       * into the plane, not y * size_x samples; that, too, is what the reference encodes */
      const uint8_t *p = (!bitshift_fn && sb == 2) ? buf + (long)y * w + 2 * x : buf + ((long)y * w + x) * sb;
      int v = sb == 1 ? p[0] : p[0] | (p[1] << 8);
      if (bitshift_fn) v = shift > 0 ? (v + (1 << (shift - 1))) >> shift : v << (-shift);
      out[(long)(y + dy) * cw + x + dx] = (jmo_pel)v;
    }
  for (y = 0; y < oh; y++) for (x = ow; x < cw; x++) out[(long)y * cw + x] = out[(long)y * cw + x - 1];
  for (y = oh; y < ch; y++) for (x = 0; x < cw; x++) out[(long)y * cw + x] = out[(long)(y - 1) * cw + x];
}
void jmo_load_frame_ex(const uint8_t *raw, int yuv, int src_w, int src_h, int out_w, int out_h, int W, int H, int symbol_bytes,
                       const int src_depth[3], const int out_depth[3], jmo_pel *y, jmo_pel *u, jmo_pel *v)
{
  const int sx = (yuv == 1 || yuv == 2) ? 1 : 0, sy = yuv == 1 ? 1 : 0;
  const int fn = !(src_depth[0] == out_depth[0] && src_depth[1] == out_depth[1]);
  const long by = (long)src_w * src_h * symbol_bytes, bc = (long)(src_w >> sx) * (src_h >> sy) * symbol_bytes;
  load_plane_ex(raw, src_w, src_h, out_w, out_h, W, H, symbol_bytes, src_depth[0] - out_depth[0], fn, y);
  if (yuv) {
    load_plane_ex(raw + by, src_w >> sx, src_h >> sy, out_w >> sx, out_h >> sy, W >> sx, H >> sy, symbol_bytes, src_depth[1] - out_depth[1], fn, u);
    load_plane_ex(raw + by + bc, src_w >> sx, src_h >> sy, out_w >> sx, out_h >> sy, W >> sx, H >> sy, symbol_bytes, src_depth[2] - out_depth[2], fn, v);
  }
}
