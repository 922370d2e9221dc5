/*
 * oracle/ref_call.c -- TEST INFRASTRUCTURE: direct calls into the REAL reference's candidate-distortion functions.
 *
 * Linked (by oracle/Makefile.ref, target `call`) with the unmodified JM 19.0 lencod objects into oracle/_ref/libjmrefcall.so.
 * The encoder's own configurations only ever reach a few weight values (foreman has no fades); this shim dresses caller-supplied
 * planes, block and weights in the structures the reference's functions read and calls them, so that the fixture
 * tests/golden/pred_dist.npz (made by tests/golden/make_pred_dist.py) pins the oracle's -- and through it the kernel's -- weighted and
 * bi-predictive distortions on arbitrary weights, offsets, denominators, block shapes and out-of-picture candidates.
 *
 * It is our own code written against JM's public headers; it contains no reference source.  It only builds where /root/reference is
 * present and only into oracle/_ref/.
 *
 * Called (reference file:line, lencod/src/me_distortion.c):
 *   computeSAD :349   computeSADWP :434   computeBiPredSAD1 :525   computeBiPredSAD2 :624
 *   computeSATD :745  computeSATDWP :833  computeBiPredSATD1 :943  computeBiPredSATD2 :1038
 *   computeSSE :1190  computeSSEWP :1261  computeBiPredSSE1 :1353  computeBiPredSSE2 :1438
 */
#include <stdlib.h>
#include <string.h>
#include "global.h"
#include "image.h"
#include "mbuffer.h"
#include "mv_search.h"
#include "me_distortion.h"

typedef struct {
  StorablePicture pic;
  imgpel **rows[4][4];         /* row pointers of plane (j,i) */
  imgpel ***subrow[4];         /* subrow[j][i] = rows[j][i] + IMG_PAD_SIZE_Y: row -IMG_PAD_SIZE_Y is the first one */
} refcall_pic;

/* planes: 16 planes of (H + 2*IMG_PAD_SIZE_Y) rows x (W + 2*IMG_PAD_SIZE_X) samples, plane (y&3)*4 + (x&3), rows back to back: the
 * layout get_mem4Dpel_pad gives p_curr_img_sub (consecutive rows padded_size_x apart, which the distortion loops rely on) */
static void dress_picture(refcall_pic *r, const imgpel *planes, int W, int H)
{
  const int pitch = W + 2 * IMG_PAD_SIZE_X, hp = H + 2 * IMG_PAD_SIZE_Y;
  int j, i, y;
  memset(&r->pic, 0, sizeof r->pic);
  for (j = 0; j < 4; j++) {
    r->subrow[j] = (imgpel ***)malloc(4 * sizeof(imgpel **));
    for (i = 0; i < 4; i++) {
      imgpel **rows = (imgpel **)malloc((size_t)hp * sizeof(imgpel *));
      for (y = 0; y < hp; y++) rows[y] = (imgpel *)planes + ((size_t)(j * 4 + i) * hp + y) * pitch + IMG_PAD_SIZE_X;
      r->rows[j][i] = rows;
      r->subrow[j][i] = rows + IMG_PAD_SIZE_Y;
    }
  }
  r->pic.p_curr_img_sub = r->subrow;
  r->pic.size_x = W; r->pic.size_y = H;
  r->pic.size_x_pad = W + 2 * IMG_PAD_SIZE_X - 1 - MB_BLOCK_SIZE - IMG_PAD_SIZE_X;      /* as alloc_storable_picture sets them */
  r->pic.size_y_pad = H + 2 * IMG_PAD_SIZE_Y - 1 - MB_BLOCK_SIZE - IMG_PAD_SIZE_Y;
}
static void undress_picture(refcall_pic *r)
{
  int j, i;
  for (j = 0; j < 4; j++) { for (i = 0; i < 4; i++) free(r->rows[j][i]); free(r->subrow[j]); }
}

/* pred: 0 = computeBiPred*1, 1 = computeBiPred*2, 2 = compute*WP, 3 = compute*; metric: 0 SAD, 1 SSE, 2 SATD.
 * cand*: absolute quarter-pel positions (block position << 2 already added), as the MotionVector the reference's callers pass. */
long long refcall_pred_dist(int pred, int metric, int W, int H, const unsigned short *planes1, const unsigned short *planes2,
                            const unsigned short *orig, int bsx, int bsy, int test8x8,
                            int weight1, int weight2, int offset, int luma_log_weight_denom, int wp_luma_round,
                            long long min_mcost, int cand1_x, int cand1_y, int cand2_x, int cand2_y)
{
  static VideoParameters vid;
  static Slice slice;
  static MEBlock mb;
  refcall_pic r1, r2;
  MotionVector c1, c2;
  imgpel *orig_rows[3];
  distblk d = -1;
  memset(&vid, 0, sizeof vid); memset(&slice, 0, sizeof slice); memset(&mb, 0, sizeof mb);
  dress_picture(&r1, (const imgpel *)planes1, W, H);
  dress_picture(&r2, (const imgpel *)(planes2 ? planes2 : planes1), W, H);
  vid.width = W; vid.height = H;
  vid.padded_size_x = W + 2 * IMG_PAD_SIZE_X;
  vid.padded_size_x_m8x8 = vid.padded_size_x - BLOCK_SIZE_8x8;
  vid.padded_size_x_m4x4 = vid.padded_size_x - BLOCK_SIZE;
  vid.max_imgpel_value = 255;
  vid.dc_pred_value_comp[0] = 128;
  slice.luma_log_weight_denom = (short)luma_log_weight_denom;
  slice.wp_luma_round = wp_luma_round;
  mb.p_Vid = &vid; mb.p_Slice = &slice;
  mb.blocksize_x = (short)bsx; mb.blocksize_y = (short)bsy;
  orig_rows[0] = (imgpel *)orig; orig_rows[1] = orig_rows[2] = NULL;
  mb.orig_pic = orig_rows;
  mb.test8x8 = (short)test8x8;
  mb.ChromaMEEnable = 0;
  mb.weight_luma = (short)weight1; mb.offset_luma = (short)offset;
  mb.weight1 = (short)weight1; mb.weight2 = (short)weight2; mb.offsetBi = (short)offset;
  c1.mv_x = (short)cand1_x; c1.mv_y = (short)cand1_y; c2.mv_x = (short)cand2_x; c2.mv_y = (short)cand2_y;
  switch (pred * 3 + metric) {
  case 0:  d = computeBiPredSAD1(&r1.pic, &r2.pic, &mb, min_mcost, &c1, &c2); break;
  case 1:  d = computeBiPredSSE1(&r1.pic, &r2.pic, &mb, min_mcost, &c1, &c2); break;
  case 2:  d = computeBiPredSATD1(&r1.pic, &r2.pic, &mb, min_mcost, &c1, &c2); break;
  case 3:  d = computeBiPredSAD2(&r1.pic, &r2.pic, &mb, min_mcost, &c1, &c2); break;
  case 4:  d = computeBiPredSSE2(&r1.pic, &r2.pic, &mb, min_mcost, &c1, &c2); break;
  case 5:  d = computeBiPredSATD2(&r1.pic, &r2.pic, &mb, min_mcost, &c1, &c2); break;
  case 6:  d = computeSADWP(&r1.pic, &mb, min_mcost, &c1); break;
  case 7:  d = computeSSEWP(&r1.pic, &mb, min_mcost, &c1); break;
  case 8:  d = computeSATDWP(&r1.pic, &mb, min_mcost, &c1); break;
  case 9:  d = computeSAD(&r1.pic, &mb, min_mcost, &c1); break;
  case 10: d = computeSSE(&r1.pic, &mb, min_mcost, &c1); break;
  case 11: d = computeSATD(&r1.pic, &mb, min_mcost, &c1); break;
  }
  undress_picture(&r1); undress_picture(&r2);
  return (long long)d;
}

/* ---- the source picture reader: read_one_frame's buf2img calls (lcommon/src/input.c:822-853, non-RGB order) with the function initInput (:41-53) selects,
 * then pad_borders (:880-925).  buf = one frame as ReadFrameSeparate leaves it in p_Vid->buf (planar Y, U, V; symbol_bytes per sample, little endian).
 * y / u / v: tight planes of the coded size (coded_w x coded_h, chroma per yuv), zero-filled by the caller like freshly allocated imgpel planes. */
#include "input.h"
void buf2img_basic(imgpel **imgX, unsigned char *buf, int size_x, int size_y, int o_size_x, int o_size_y, int symbol_size_in_bytes, int bitshift);
void buf2img_bitshift(imgpel **imgX, unsigned char *buf, int size_x, int size_y, int o_size_x, int o_size_y, int symbol_size_in_bytes, int bitshift);
int refcall_load_frame(unsigned char *buf, int yuv, int src_w, int src_h, int out_w, int out_h, int coded_w, int coded_h, int symbol_bytes,
                       const int src_depth[3], const int out_depth[3], imgpel *y, imgpel *u, imgpel *v)
{
  const int sx = (yuv == YUV420 || yuv == YUV422) ? 1 : 0, sy = yuv == YUV420 ? 1 : 0;
  const int w[2] = {src_w, yuv == YUV400 ? 0 : src_w >> sx}, h[2] = {src_h, yuv == YUV400 ? 0 : src_h >> sy};
  const int ow[2] = {out_w, yuv == YUV400 ? 0 : out_w >> sx}, oh[2] = {out_h, yuv == YUV400 ? 0 : out_h >> sy};
  const int cw[2] = {coded_w, yuv == YUV400 ? 0 : coded_w >> sx}, ch[2] = {coded_h, yuv == YUV400 ? 0 : coded_h >> sy};
  imgpel *planes[3] = {y, u, v};
  imgpel **rows[3];
  void (*b2i)(imgpel **, unsigned char *, int, int, int, int, int, int) =
      (src_depth[0] == out_depth[0] && src_depth[1] == out_depth[1]) ? buf2img_basic : buf2img_bitshift;
  const long bytes_y = (long)w[0] * h[0] * symbol_bytes, bytes_uv = (long)w[1] * h[1] * symbol_bytes;
  FrameFormat out;
  int k, j;
  for (k = 0; k < 3; k++) {
    const int c = k ? 1 : 0;
    rows[k] = (imgpel **)malloc((size_t)(ch[c] > 0 ? ch[c] : 1) * sizeof(imgpel *));
    for (j = 0; j < ch[c]; j++) rows[k][j] = planes[k] + (size_t)j * cw[c];
  }
  b2i(rows[0], buf, w[0], h[0], ow[0], oh[0], symbol_bytes, src_depth[0] - out_depth[0]);
  if (yuv != YUV400) {
    b2i(rows[1], buf + bytes_y, w[1], h[1], ow[1], oh[1], symbol_bytes, src_depth[1] - out_depth[1]);
    b2i(rows[2], buf + bytes_y + bytes_uv, w[1], h[1], ow[1], oh[1], symbol_bytes, src_depth[2] - out_depth[2]);
  }
  memset(&out, 0, sizeof out);
  out.yuv_format = (ColorFormat)yuv;
  out.width[0] = ow[0]; out.height[0] = oh[0]; out.width[1] = out.width[2] = ow[1]; out.height[1] = out.height[2] = oh[1];
  pad_borders(out, cw[0], ch[0], cw[1], ch[1], rows);
  for (k = 0; k < 3; k++) free(rows[k]);
  return 0;
}
