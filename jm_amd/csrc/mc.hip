// mc.hip -- motion-compensated prediction of block lists, un-weighted, frame pictures (gfx950).  SURVEY.md 8f row 2.
//
// Same samples, bit for bit, as
//   luma_prediction         lencod/src/mc_prediction.c:144-236  (OneComponentLumaPrediction :122-136, mc_prediction :100-110,
//                           bi_prediction :82-93) reading the sixteen quarter-pel planes k_subplanes made, and
//   chroma_prediction_4x4   :568-650 with OneComponentChromaPrediction4x4_retrieve :361-411 (ChromaMCBuffer = 1).
// The reference keeps 64 (4:2:0) / 32 (4:2:2) pre-interpolated chroma sub-images per plane (getSubImagesChroma,
// lencod/src/img_chroma.c:338-437) and reads two samples per (row, pair) through UMVLine8X_chroma (refbuf.h:61-65); every
// such sample is (w00 S[Y][X] + w01 S[Y][X+1] + w10 S[Y+1][X] + w11 S[Y+1][X+1] + 32) >> 6 of the integer plane S with all
// coordinates clamped into the picture, so the kernel interpolates on the fly from the integer planes: 4 byte gathers per
// sample from an L2-resident plane instead of 32-64x the plane in HBM.
//
// Both kernels are gathers: algorithmic bytes = w*h in per list + w*h out (luma), 16 out + 4..36 in (chroma); HBM/L2-bound.
#include "jmhip_internal.h"

struct McSlots { const uint8_t *p[32]; };
static_assert(sizeof(jmhip_mc_weights) == 12, "jmhip_mc_weights is 12 bytes in include/jmhip.h");

struct __attribute__((packed)) u32un { uint32_t v; };
__device__ __forceinline__ uint32_t ld4u(const uint8_t *p) { return ((const u32un *)p)->v; }
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }   // v_lerp_u8, round bit set: per byte (a + b + 1) >> 1

// 16 lanes per block; a lane copies every 16th group of four samples (one group for 4x4 .. 8x8, four for 16x16)
// weighted sample prediction (mc_prediction.c:38-73): one list clip1(((w * p + round) >> shift) + offset), both lists
// clip1(((w0 * p0 + w1 * p1 + round) >> shift) + offset); four samples of a 32-bit word at a time
__device__ __forceinline__ int wp1(int p, const jmhip_mc_weights &w, int list) { return min(max(((w.weight[list] * p + w.round) >> w.shift) + w.offset, 0), 255); }
__device__ __forceinline__ int wp2(int p0, int p1, const jmhip_mc_weights &w) { return min(max(((w.weight[0] * p0 + w.weight[1] * p1 + w.round) >> w.shift) + w.offset, 0), 255); }
__device__ __forceinline__ uint32_t wp_word(uint32_t a, uint32_t c, int dir, const jmhip_mc_weights &w)
{
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int p0 = (a >> (8 * k)) & 255, p1 = (c >> (8 * k)) & 255;
    r |= (uint32_t)(dir == 2 ? wp2(p0, p1, w) : (dir == 0 ? wp1(p0, w, 0) : wp1(p1, w, 1))) << (8 * k);
  }
  return r;
}

__global__ __launch_bounds__(256) void k_mc_luma(const jmhip_mc_luma_blk *__restrict__ blocks, const jmhip_mc_weights *__restrict__ weights, int n,
                                                 McSlots slots, int nslots, int pitch, long plane_stride, int W, int H, uint8_t *__restrict__ out)
{
  const int t = blockIdx.x * 256 + threadIdx.x, b = t >> 4, l = t & 15;
  if (b >= n) return;
  const jmhip_mc_luma_blk q = blocks[b];
  if (q.dir > 2) return;
  // the origin of each list: UMVLine4X clamps the block's origin once (refbuf.h:22-26)
  long o0 = 0, o1 = 0;
  int s0 = 0, s1 = 0;
#pragma unroll
  for (int list = 0; list < 2; list++) {
    if (q.dir != list && q.dir != 2) continue;
    const int s = q.slot[list];
    if (s < 0 || s >= nslots) return;
    const int qx = (q.x << 2) + q.mv[list][0], qy = (q.y << 2) + q.mv[list][1];
    const int yy = min(max(qy >> 2, -JMHIP_PAD_Y), H + 3), xx = min(max(qx >> 2, -JMHIP_PAD_X), W + 15);
    const long o = ((qy & 3) * 4 + (qx & 3)) * plane_stride + (long)(yy + JMHIP_PAD_Y) * pitch + xx + JMHIP_PAD_X;
    if (list == 0) { o0 = o; s0 = s; } else { o1 = o; s1 = s; }
  }
  const uint8_t *__restrict__ src0 = slots.p[s0] + o0, *__restrict__ src1 = slots.p[s1] + o1;
  jmhip_mc_weights wt = {};
  if (weights) wt = weights[b];
  const int w4 = q.w >> 2, groups = w4 * q.h;
  for (int g = l; g < groups; g += 16) {
    const int row = g / w4, c4 = g - row * w4;
    const long off = (long)row * pitch + 4 * c4;
    const uint32_t a = q.dir != 1 ? ld4u(src0 + off) : 0u, c = q.dir != 0 ? ld4u(src1 + off) : 0u;
    *(uint32_t *)(out + (long)b * 256 + row * q.w + 4 * c4) = weights ? wp_word(a, c, q.dir, wt) : (q.dir == 0 ? a : (q.dir == 1 ? c : avg4(a, c)));
  }
}

// 16 lanes per block: lane -> sample (row j, column i); the sample pair i >> 1 shares a vector
__global__ __launch_bounds__(256) void k_mc_chroma(const jmhip_mc_chroma_blk *__restrict__ blocks, const jmhip_mc_weights *__restrict__ weights, int n,
                                                   McSlots slots, int nslots, int cw, int ch, int yuv, uint8_t *__restrict__ out)
{
  const int t = blockIdx.x * 256 + threadIdx.x, b = t >> 4, smp = t & 15, j = smp >> 2, i = smp & 3, hp = i >> 1, o = i & 1;
  if (b >= n) return;
  const jmhip_mc_chroma_blk *q = blocks + b;
  const int dir = q->dir, plane = q->plane;
  if (dir > 2 || plane > 1) return;
  const int sy = yuv == 2 ? 2 : 3, my = yuv == 2 ? 3 : 7, ky = yuv == 2 ? 2 : 1;      // chroma_shift_y, chroma_mask_mv_y (lencod.c:2366-2380), eighths per phase
  const int pad_x = JMHIP_PAD_X >> 1, pad_y = yuv == 2 ? JMHIP_PAD_Y : JMHIP_PAD_Y >> 1;
  const int max_x = cw - 1 + pad_x - 8, max_y = ch - 1 + pad_y - (yuv == 2 ? 16 : 8);  // size_x_cr_pad, size_y_cr_pad (mbuffer.c:568-569)
  int v[2] = {0, 0};
#pragma unroll
  for (int list = 0; list < 2; list++) {
    if (dir != list && dir != 2) continue;
    const int s = q->slot[list];
    if (s < 0 || s >= nslots || !slots.p[s]) return;
    const uint8_t *pl = slots.p[s] + (long)plane * cw * ch;
    const int ii = ((q->x + 2 * hp) << 3) + q->mv[list][j][hp][0], jj = ((q->y + j) << sy) + q->mv[list][j][hp][1];
    const int X = min(max(ii >> 3, -pad_x), max_x) + o, Y = min(max(jj >> sy, -pad_y), max_y);
    const int lx = ii & 7, k = (jj & my) * ky, m = 8 - k;
    const int w01 = m * lx, w00 = (m << 3) - w01, w11 = k * lx, w10 = (k << 3) - w11;
    const int y0 = min(max(Y, 0), ch - 1), y1 = min(max(Y + 1, 0), ch - 1), x0 = min(max(X, 0), cw - 1), x1 = min(max(X + 1, 0), cw - 1);
    v[list] = (w00 * pl[y0 * cw + x0] + w01 * pl[y0 * cw + x1] + w10 * pl[y1 * cw + x0] + w11 * pl[y1 * cw + x1] + 32) >> 6;
  }
  if (weights) {
    const jmhip_mc_weights wt = weights[b];
    out[(long)b * 16 + smp] = (uint8_t)(dir == 2 ? wp2(v[0], v[1], wt) : wp1(v[dir], wt, dir));
  } else out[(long)b * 16 + smp] = (uint8_t)(dir == 0 ? v[0] : (dir == 1 ? v[1] : (v[0] + v[1] + 1) >> 1));
}

static McSlots luma_slots(jmhip_ctx *ctx) { McSlots s; for (int k = 0; k < 32; k++) s.p[k] = k < ctx->cfg.num_ref_slots ? ctx->d_sub[k] : nullptr; return s; }
static McSlots chroma_slots(jmhip_ctx *ctx) { McSlots s; for (int k = 0; k < 32; k++) s.p[k] = k < ctx->cfg.num_ref_slots ? ctx->d_refc[k] : nullptr; return s; }

extern "C" int jmhip_mc_luma_wp_dev(jmhip_ctx *ctx, const jmhip_mc_luma_blk *d_blocks, const jmhip_mc_weights *d_weights, int32_t n, uint8_t *d_out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || (n > 0 && (!d_blocks || !d_out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_luma_dev: bad argument");
  if (n == 0) return JMHIP_OK;
  hipLaunchKernelGGL(k_mc_luma, dim3((n + 15) / 16), dim3(256), 0, ctx->stream, d_blocks, d_weights, n, luma_slots(ctx), ctx->cfg.num_ref_slots,
                     ctx->pitch, (long)ctx->plane_stride, ctx->W, ctx->H, d_out);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}
extern "C" int jmhip_mc_luma_dev(jmhip_ctx *ctx, const jmhip_mc_luma_blk *d_blocks, int32_t n, uint8_t *d_out)
{
  return jmhip_mc_luma_wp_dev(ctx, d_blocks, nullptr, n, d_out);
}

// the weights of a host batch: shift 0..8 (log_weight_denom 0..7, + 1 for two lists)
static int check_weights(jmhip_ctx *ctx, const char *who, const jmhip_mc_weights *weights, int n)
{
  for (int i = 0; weights && i < n; i++)
    if (weights[i].shift < 0 || weights[i].shift > 8) return jmhip_fail(ctx, JMHIP_EINVAL, "%s: block %d: weight shift %d outside 0..8", who, i, weights[i].shift);
  return JMHIP_OK;
}
// weights (may be NULL) -> a device copy behind the block records in scratch buffer 0 (`din`, sized by blocks_bytes())
static size_t blocks_bytes(size_t n, size_t rec, bool with_weights) { return ((n * rec + 15) & ~(size_t)15) + (with_weights ? n * sizeof(jmhip_mc_weights) : 0); }
static int stage_weights(jmhip_ctx *ctx, const jmhip_mc_weights *weights, int n, void *din, size_t rec, const jmhip_mc_weights **d_weights)
{
  *d_weights = nullptr;
  if (!weights) return JMHIP_OK;
  void *dw = (char *)din + (((size_t)n * rec + 15) & ~(size_t)15);
  HIPCHK(ctx, hipMemcpyAsync(dw, weights, (size_t)n * sizeof(jmhip_mc_weights), hipMemcpyHostToDevice, ctx->stream));
  *d_weights = (const jmhip_mc_weights *)dw;
  return JMHIP_OK;
}

extern "C" int jmhip_mc_luma_wp(jmhip_ctx *ctx, const jmhip_mc_luma_blk *blocks, const jmhip_mc_weights *weights, int32_t n, uint8_t *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || (n > 0 && (!blocks || !out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_luma: bad argument");
  { int r_ = check_weights(ctx, "jmhip_mc_luma_wp", weights, n); if (r_) return r_; }
  for (int i = 0; i < n; i++) {
    const jmhip_mc_luma_blk *q = blocks + i;
    const bool size_ok = (q->w == 4 || q->w == 8 || q->w == 16) && (q->h == 4 || q->h == 8 || q->h == 16);
    if (!size_ok || q->dir > 2) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_luma: block %d: size %dx%d, direction %d", i, q->w, q->h, q->dir);
    for (int l = 0; l < 2; l++)
      if ((q->dir == l || q->dir == 2) && (q->slot[l] < 0 || q->slot[l] >= ctx->cfg.num_ref_slots))
        return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_luma: block %d: list %d reference slot %d", i, l, q->slot[l]);
  }
  if (n == 0) return JMHIP_OK;
  int r; void *din, *dout;
  if ((r = jmhip_scratch(ctx, 0, blocks_bytes((size_t)n, sizeof(jmhip_mc_luma_blk), weights != nullptr), &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * 256, &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(din, blocks, (size_t)n * sizeof(jmhip_mc_luma_blk), hipMemcpyHostToDevice, ctx->stream));
  const jmhip_mc_weights *dw;
  if ((r = stage_weights(ctx, weights, n, din, sizeof(jmhip_mc_luma_blk), &dw))) return r;
  if ((r = jmhip_mc_luma_wp_dev(ctx, (const jmhip_mc_luma_blk *)din, dw, n, (uint8_t *)dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * 256, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}
extern "C" int jmhip_mc_luma(jmhip_ctx *ctx, const jmhip_mc_luma_blk *blocks, int32_t n, uint8_t *out)
{
  return jmhip_mc_luma_wp(ctx, blocks, nullptr, n, out);
}

extern "C" int jmhip_mc_chroma_wp_dev(jmhip_ctx *ctx, const jmhip_mc_chroma_blk *d_blocks, const jmhip_mc_weights *d_weights, int32_t n, uint8_t *d_out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || (n > 0 && (!d_blocks || !d_out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_chroma_dev: bad argument");
  if (ctx->cfg.yuv_format != 1 && ctx->cfg.yuv_format != 2) return jmhip_fail(ctx, JMHIP_EUNSUPPORTED, "jmhip_mc_chroma: yuv_format %d (4:2:0 and 4:2:2 only)", ctx->cfg.yuv_format);
  if (n == 0) return JMHIP_OK;
  hipLaunchKernelGGL(k_mc_chroma, dim3((n + 15) / 16), dim3(256), 0, ctx->stream, d_blocks, d_weights, n, chroma_slots(ctx), ctx->cfg.num_ref_slots,
                     ctx->cw, ctx->ch, ctx->cfg.yuv_format, d_out);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}
extern "C" int jmhip_mc_chroma_dev(jmhip_ctx *ctx, const jmhip_mc_chroma_blk *d_blocks, int32_t n, uint8_t *d_out)
{
  return jmhip_mc_chroma_wp_dev(ctx, d_blocks, nullptr, n, d_out);
}

extern "C" int jmhip_mc_chroma_wp(jmhip_ctx *ctx, const jmhip_mc_chroma_blk *blocks, const jmhip_mc_weights *weights, int32_t n, uint8_t *out)
{
  if (!ctx) return JMHIP_EINVAL;
  { int r_ = check_weights(ctx, "jmhip_mc_chroma_wp", weights, n); if (r_) return r_; }
  if (n < 0 || (n > 0 && (!blocks || !out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_chroma: bad argument");
  if (ctx->cfg.yuv_format != 1 && ctx->cfg.yuv_format != 2) return jmhip_fail(ctx, JMHIP_EUNSUPPORTED, "jmhip_mc_chroma: yuv_format %d (4:2:0 and 4:2:2 only)", ctx->cfg.yuv_format);
  for (int i = 0; i < n; i++) {
    const jmhip_mc_chroma_blk *q = blocks + i;
    if (q->dir > 2 || q->plane > 1) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_chroma: block %d: direction %d, plane %d", i, q->dir, q->plane);
    for (int l = 0; l < 2; l++)
      if ((q->dir == l || q->dir == 2) && (q->slot[l] < 0 || q->slot[l] >= ctx->cfg.num_ref_slots || !ctx->d_refc[q->slot[l]]))
        return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_chroma: block %d: list %d reference slot %d has no chroma planes", i, l, q->slot[l]);
  }
  if (n == 0) return JMHIP_OK;
  int r; void *din, *dout;
  if ((r = jmhip_scratch(ctx, 0, blocks_bytes((size_t)n, sizeof(jmhip_mc_chroma_blk), weights != nullptr), &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * 16, &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(din, blocks, (size_t)n * sizeof(jmhip_mc_chroma_blk), hipMemcpyHostToDevice, ctx->stream));
  const jmhip_mc_weights *dw;
  if ((r = stage_weights(ctx, weights, n, din, sizeof(jmhip_mc_chroma_blk), &dw))) return r;
  if ((r = jmhip_mc_chroma_wp_dev(ctx, (const jmhip_mc_chroma_blk *)din, dw, n, (uint8_t *)dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}
extern "C" int jmhip_mc_chroma(jmhip_ctx *ctx, const jmhip_mc_chroma_blk *blocks, int32_t n, uint8_t *out)
{
  return jmhip_mc_chroma_wp(ctx, blocks, nullptr, n, out);
}


// ---- frame-level glue for P pictures coded as 16x16 macroblocks: prediction straight from the refinement's results into the
//      block order the transform/quant kernel reads; 64 lanes per macroblock, lane -> (row, four samples)
__global__ __launch_bounds__(256) void k_mc_mb16(const jmhip_me_job *__restrict__ jobs, const jmhip_me_result *__restrict__ results, int n,
                                                 const uint8_t *__restrict__ planes, int pitch, long plane_stride, int W, int H,
                                                 int y_offset, int blocks_per_row, uint8_t *__restrict__ pred)
{
  const int t = blockIdx.x * 256 + threadIdx.x, b = t >> 6, l = t & 63, row = l >> 2, c4 = l & 3;
  if (b >= n) return;
  const int mb_x = jobs[b].mb_x, mb_y = jobs[b].mb_y;
  const jmhip_me_best mv = results[b].best[0];
  const int qx = (mb_x << 2) + mv.mv_x, qy = (mb_y << 2) + mv.mv_y;
  const int yy = min(max(qy >> 2, -JMHIP_PAD_Y), H + 3), xx = min(max(qx >> 2, -JMHIP_PAD_X), W + 15);
  const uint32_t v = ld4u(planes + ((qy & 3) * 4 + (qx & 3)) * plane_stride + (long)(yy + JMHIP_PAD_Y + row) * pitch + xx + JMHIP_PAD_X + 4 * c4);
  const long blk = (long)(((mb_y - y_offset) >> 2) + (row >> 2)) * blocks_per_row + (mb_x >> 2) + c4;
  *(uint32_t *)(pred + blk * 16 + (row & 3) * 4) = v;
}

__global__ __launch_bounds__(256) void k_tq_rec_to_plane(const jmhip_tq_out *__restrict__ out, int n, int blocks_per_row, uint8_t *__restrict__ plane, int pitch)
{
  const int t = blockIdx.x * 256 + threadIdx.x, b = t >> 2, r = t & 3;
  if (b >= n) return;
  const int by = b / blocks_per_row, bx = b - by * blocks_per_row;
  *(uint32_t *)(plane + (long)(4 * by + r) * pitch + 4 * bx) = *(const uint32_t *)(out[b].rec + 4 * r);
}

extern "C" int jmhip_mc_mb16_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *d_jobs, const jmhip_me_result *d_results, int32_t n,
                                 int32_t y_offset, int32_t blocks_per_row, uint8_t *d_pred)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || slot < 0 || slot >= ctx->cfg.num_ref_slots || blocks_per_row < 4 || (n > 0 && (!d_jobs || !d_results || !d_pred)))
    return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_mb16_dev: bad argument");
  if (n == 0) return JMHIP_OK;
  hipLaunchKernelGGL(k_mc_mb16, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, d_jobs, d_results, n, (const uint8_t *)ctx->d_sub[slot], ctx->pitch,
                     (long)ctx->plane_stride, ctx->W, ctx->H, y_offset, blocks_per_row, d_pred);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_tq_rec_to_plane_dev(jmhip_ctx *ctx, const jmhip_tq_out *d_out, int32_t n, int32_t blocks_per_row, uint8_t *d_plane, int32_t pitch)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || blocks_per_row < 1 || pitch < 4 * blocks_per_row || (pitch & 3) || (n > 0 && (!d_out || !d_plane)))
    return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_rec_to_plane_dev: bad argument");
  if (n == 0) return JMHIP_OK;
  hipLaunchKernelGGL(k_tq_rec_to_plane, dim3((n + 63) / 64), dim3(256), 0, ctx->stream, d_out, n, blocks_per_row, d_plane, pitch);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

// chroma of the same macroblocks: 2 planes x (8 x RH) samples per job, one lane per sample pair row segment of four samples:
// lane -> (plane, row, half); every sample of the macroblock uses the job's 16x16 vector (chroma_prediction_4x4's per-pair vectors coincide)
__global__ __launch_bounds__(256) void k_mc_mb16_chroma(const jmhip_me_job *__restrict__ jobs, const jmhip_me_result *__restrict__ results, int n,
                                                        const uint8_t *__restrict__ planes, int cw, int ch, int yuv, uint8_t *__restrict__ pred)
{
  const int RH = yuv == 2 ? 16 : 8, per = 2 * RH * 2;                        // lanes per job: plane x row x half row
  const int t = blockIdx.x * 256 + threadIdx.x, b = t / per, l = t - b * per;
  if (b >= n) return;
  const int plane = l / (RH * 2), row = (l >> 1) % RH, half = l & 1;
  const int mb_cx = jobs[b].mb_x >> 1, mb_cy = yuv == 2 ? jobs[b].mb_y : jobs[b].mb_y >> 1;
  const jmhip_me_best mv = results[b].best[0];
  const int sy = yuv == 2 ? 2 : 3, my = yuv == 2 ? 3 : 7, ky = yuv == 2 ? 2 : 1;
  const int pad_x = JMHIP_PAD_X >> 1, pad_y = yuv == 2 ? JMHIP_PAD_Y : JMHIP_PAD_Y >> 1;
  const int max_x = cw - 1 + pad_x - 8, max_y = ch - 1 + pad_y - (yuv == 2 ? 16 : 8);
  const uint8_t *pl = planes + (long)plane * cw * ch;
  uint32_t w = 0;
#pragma unroll
  for (int hp = 0; hp < 2; hp++) {                                           // the two sample pairs of this half row
    const int ii = ((mb_cx + 4 * half + 2 * hp) << 3) + mv.mv_x, jj = ((mb_cy + row) << sy) + mv.mv_y;
    const int X0 = min(max(ii >> 3, -pad_x), max_x), Y = min(max(jj >> sy, -pad_y), max_y);
    const int lx = ii & 7, k = (jj & my) * ky, m = 8 - k;
    const int w01 = m * lx, w00 = (m << 3) - w01, w11 = k * lx, w10 = (k << 3) - w11;
    const int y0 = min(max(Y, 0), ch - 1), y1 = min(max(Y + 1, 0), ch - 1);
#pragma unroll
    for (int o = 0; o < 2; o++) {
      const int x0 = min(max(X0 + o, 0), cw - 1), x1 = min(max(X0 + o + 1, 0), cw - 1);
      const int v = (w00 * pl[y0 * cw + x0] + w01 * pl[y0 * cw + x1] + w10 * pl[y1 * cw + x0] + w11 * pl[y1 * cw + x1] + 32) >> 6;
      w |= (uint32_t)v << (8 * (2 * hp + o));
    }
  }
  *(uint32_t *)(pred + ((long)b * 2 + plane) * 128 + row * 8 + 4 * half) = w;
}

__global__ __launch_bounds__(256) void k_tqc_rec_to_planes(const jmhip_me_job *__restrict__ jobs, const jmhip_tqc_out *__restrict__ out, int n, int yuv,
                                                           int y_offset, uint8_t *__restrict__ u, uint8_t *__restrict__ v, int pitch)
{
  const int RH = yuv == 2 ? 16 : 8, per = 2 * RH * 2;
  const int t = blockIdx.x * 256 + threadIdx.x, b = t / per, l = t - b * per;
  if (b >= n) return;
  const int plane = l / (RH * 2), row = (l >> 1) % RH, half = l & 1;
  const int cx = jobs[b].mb_x >> 1, cy = yuv == 2 ? jobs[b].mb_y - y_offset : (jobs[b].mb_y - y_offset) >> 1;
  *(uint32_t *)((plane ? v : u) + (long)(cy + row) * pitch + cx + 4 * half) = *(const uint32_t *)(out[(long)b * 2 + plane].rec + row * 8 + 4 * half);
}

extern "C" int jmhip_mc_mb16_chroma_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *d_jobs, const jmhip_me_result *d_results, int32_t n, uint8_t *d_pred)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || slot < 0 || slot >= ctx->cfg.num_ref_slots || (n > 0 && (!d_jobs || !d_results || !d_pred)))
    return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_mb16_chroma_dev: bad argument");
  if (ctx->cfg.yuv_format != 1 && ctx->cfg.yuv_format != 2) return jmhip_fail(ctx, JMHIP_EUNSUPPORTED, "jmhip_mc_mb16_chroma_dev: yuv_format %d", ctx->cfg.yuv_format);
  if (!ctx->d_refc[slot]) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mc_mb16_chroma_dev: reference slot %d has no chroma planes", slot);
  if (n == 0) return JMHIP_OK;
  const int per = (ctx->cfg.yuv_format == 2 ? 16 : 8) * 4;
  hipLaunchKernelGGL(k_mc_mb16_chroma, dim3(((long)n * per + 255) / 256), dim3(256), 0, ctx->stream, d_jobs, d_results, n, (const uint8_t *)ctx->d_refc[slot],
                     ctx->cw, ctx->ch, ctx->cfg.yuv_format, d_pred);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_tqc_rec_to_planes_dev(jmhip_ctx *ctx, const jmhip_me_job *d_jobs, const jmhip_tqc_out *d_out, int32_t n, int32_t y_offset,
                                           uint8_t *d_u, uint8_t *d_v, int32_t pitch)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || pitch < ctx->cw || (pitch & 3) || (n > 0 && (!d_jobs || !d_out || !d_u || !d_v)))
    return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tqc_rec_to_planes_dev: bad argument");
  if (ctx->cfg.yuv_format != 1 && ctx->cfg.yuv_format != 2) return jmhip_fail(ctx, JMHIP_EUNSUPPORTED, "jmhip_tqc_rec_to_planes_dev: yuv_format %d", ctx->cfg.yuv_format);
  if (n == 0) return JMHIP_OK;
  const int per = (ctx->cfg.yuv_format == 2 ? 16 : 8) * 4;
  hipLaunchKernelGGL(k_tqc_rec_to_planes, dim3(((long)n * per + 255) / 256), dim3(256), 0, ctx->stream, d_jobs, d_out, n, ctx->cfg.yuv_format, y_offset, d_u, d_v, pitch);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

// ---- K6: the chroma sub-images themselves (getSubImagesChroma, lencod/src/img_chroma.c:338-437), for callers that keep JM's host-side
//      chroma prediction: sub-image (sy, sx) of a plane = the clamped bilinear blend above at every position of the padded plane.
//      One thread per four horizontally adjacent samples; the integer plane (one per launch) stays in L2, the 64 (4:2:0) / 32 (4:2:2)
//      output planes are a stream: HBM-bound on the stores.
__global__ __launch_bounds__(256) void k_chroma_subplanes(const uint8_t *__restrict__ pl, int cw, int ch, int yuv, uint8_t *__restrict__ out)
{
  const int pad_x = JMHIP_PAD_X >> 1, pad_y = yuv == 2 ? JMHIP_PAD_Y : JMHIP_PAD_Y >> 1, ky = yuv == 2 ? 2 : 1;
  const int Wp = cw + 2 * pad_x, Hp = ch + 2 * pad_y, W4 = Wp >> 2;                  // Wp is a multiple of 4 (cw is a multiple of 8)
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int x4 = (int)(t % W4), row = (int)((t / W4) % Hp), sub = (int)(t / ((long)W4 * Hp));
  if (sub >= (yuv == 2 ? 32 : 64)) return;
  const int sy = sub >> 3, sx = sub & 7, k = sy * ky, m = 8 - k;
  const int w01 = m * sx, w00 = (m << 3) - w01, w11 = k * sx, w10 = (k << 3) - w11;
  const int Y = row - pad_y, y0 = min(max(Y, 0), ch - 1), y1 = min(max(Y + 1, 0), ch - 1);
  uint32_t w = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int X = 4 * x4 + i - pad_x, x0 = min(max(X, 0), cw - 1), x1 = min(max(X + 1, 0), cw - 1);
    const int v = (w00 * pl[y0 * cw + x0] + w01 * pl[y0 * cw + x1] + w10 * pl[y1 * cw + x0] + w11 * pl[y1 * cw + x1] + 32) >> 6;
    w |= (uint32_t)v << (8 * i);
  }
  *(uint32_t *)(out + ((long)sub * Hp + row) * Wp + 4 * x4) = w;
}

extern "C" int jmhip_get_chroma_subplanes(jmhip_ctx *ctx, int32_t slot, int32_t plane, uint16_t *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!out || slot < 0 || slot >= ctx->cfg.num_ref_slots || plane < 0 || plane > 1) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_get_chroma_subplanes: bad argument");
  const int yuv = ctx->cfg.yuv_format;
  if (yuv != 1 && yuv != 2) return jmhip_fail(ctx, JMHIP_EUNSUPPORTED, "jmhip_get_chroma_subplanes: yuv_format %d (4:2:0 and 4:2:2 only)", yuv);
  if (!ctx->d_refc[slot]) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_get_chroma_subplanes: reference slot %d has no chroma planes", slot);
  const int pad_x = JMHIP_PAD_X >> 1, pad_y = yuv == 2 ? JMHIP_PAD_Y : JMHIP_PAD_Y >> 1, nsub = yuv == 2 ? 32 : 64;
  const int Wp = ctx->cw + 2 * pad_x, Hp = ctx->ch + 2 * pad_y;
  const size_t plane_px = (size_t)Wp * Hp;
  if (plane_px > ctx->h_stage_bytes) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_get_chroma_subplanes: plane larger than staging");
  void *dout; int r;
  if ((r = jmhip_scratch(ctx, 1, plane_px * nsub, &dout))) return r;
  const long threads = (long)(Wp >> 2) * Hp * nsub;
  hipLaunchKernelGGL(k_chroma_subplanes, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const uint8_t *)ctx->d_refc[slot] + (size_t)plane * ctx->cw * ctx->ch, ctx->cw, ctx->ch, yuv, (uint8_t *)dout);
  HIPCHK(ctx, hipGetLastError());
  for (int k = 0; k < nsub; k++) {                         // back through the pinned staging area as u8, widened to imgpel on the host
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_stage, (const uint8_t *)dout + (size_t)k * plane_px, plane_px, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    uint16_t *o = out + (size_t)k * plane_px;
    for (size_t i = 0; i < plane_px; i++) o[i] = ctx->h_stage[i];
  }
  return JMHIP_OK;
}
