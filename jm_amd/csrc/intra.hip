// intra.hip -- luma intra prediction of 4x4 blocks and the Intra16x16 mode search of whole macroblocks (gfx950).  SURVEY.md 8f row 1.
//
// Same values, bit for bit, as
//   get_intrapred_4x4      lencod/src/intra4x4.c:521-561 (the nine modes :72-308) over the predictor samples set_intrapred_4x4 :421 gathers
//                          (edge[0] = above left, [1..8] = above and above right, [9..12] = left), and
//   find_sad_16x16_JM      lencod/src/intra16x16.c:463-517: the four predictions of get_intrapred_16x16 :307 (:28-146; edge[0] = above left,
//                          [1..16] above, [17..32] left), the mode cost Slice.distI16x16 (distI16x16_sad / _sse / _satd :331-452; the SATD
//                          form sums the AC terms of hadamard4x4 of every 4x4 difference block and the hadamard4x4 of the halved DC terms),
//                          and the choice by strict '<' in ascending mode order.
// The formulas are those of H.264 8.3.1.2 / 8.3.3, which JM's per-sample assignments implement.
// k_intrapred4x4: 16 lanes per block, one sample each.  k_intra16_search: 16 lanes per macroblock, lane = 4x4 block: prediction,
// difference and Hadamard in registers, the sixteen DC terms by wave shuffles, the cost by a group reduction.
#include "jmhip_internal.h"
static_assert(sizeof(jmhip_ip4_blk) == 16 && sizeof(jmhip_i16_mb) == 40 && sizeof(jmhip_i16_out) == 1040, "intra records");

__device__ __forceinline__ int e4t(const uint8_t *e, int x) { return x < 0 ? e[0] : e[1 + x]; }     // p[x, -1], x = -1 .. 7
__device__ __forceinline__ int e4l(const uint8_t *e, int y) { return y < 0 ? e[0] : e[9 + y]; }     // p[-1, y], y = -1 .. 3

__global__ __launch_bounds__(256) void k_intrapred4x4(const jmhip_ip4_blk *__restrict__ blks, int n, uint8_t *__restrict__ out)
{
  const int t = blockIdx.x * 256 + threadIdx.x, b = t >> 4, x = t & 3, y = (t >> 2) & 3;
  if (b >= n) return;
  const uint8_t *e = blks[b].edge;
  const int mode = blks[b].mode, left = blks[b].left, up = blks[b].up;
  int v;
  switch (mode) {
  case 0: v = e4t(e, x); break;
  case 1: v = e4l(e, y); break;
  case 2:
    if (up && left) v = (e[1] + e[2] + e[3] + e[4] + e[9] + e[10] + e[11] + e[12] + 4) >> 3;
    else if (left) v = (e[9] + e[10] + e[11] + e[12] + 2) >> 2;
    else if (up) v = (e[1] + e[2] + e[3] + e[4] + 2) >> 2;
    else v = e[1];
    break;
  case 3: v = (x == 3 && y == 3) ? (e4t(e, 6) + 3 * e4t(e, 7) + 2) >> 2 : (e4t(e, x + y) + 2 * e4t(e, x + y + 1) + e4t(e, x + y + 2) + 2) >> 2; break;
  case 4:
    if (x > y) v = (e4t(e, x - y - 2) + 2 * e4t(e, x - y - 1) + e4t(e, x - y) + 2) >> 2;
    else if (x < y) v = (e4l(e, y - x - 2) + 2 * e4l(e, y - x - 1) + e4l(e, y - x) + 2) >> 2;
    else v = (e4t(e, 0) + 2 * e[0] + e4l(e, 0) + 2) >> 2;
    break;
  case 5: {
    const int z = 2 * x - y, k = x - (y >> 1);
    if (z >= 0 && !(z & 1)) v = (e4t(e, k - 1) + e4t(e, k) + 1) >> 1;
    else if (z > 0) v = (e4t(e, k - 2) + 2 * e4t(e, k - 1) + e4t(e, k) + 2) >> 2;
    else if (z == -1) v = (e4l(e, 0) + 2 * e[0] + e4t(e, 0) + 2) >> 2;
    else v = (e4l(e, y - 1) + 2 * e4l(e, y - 2) + e4l(e, y - 3) + 2) >> 2;
    break; }
  case 6: {
    const int z = 2 * y - x, k = y - (x >> 1);
    if (z >= 0 && !(z & 1)) v = (e4l(e, k - 1) + e4l(e, k) + 1) >> 1;
    else if (z > 0) v = (e4l(e, k - 2) + 2 * e4l(e, k - 1) + e4l(e, k) + 2) >> 2;
    else if (z == -1) v = (e4l(e, 0) + 2 * e[0] + e4t(e, 0) + 2) >> 2;
    else v = (e4t(e, x - 1) + 2 * e4t(e, x - 2) + e4t(e, x - 3) + 2) >> 2;
    break; }
  case 7: {
    const int k = x + (y >> 1);
    v = (y & 1) ? (e4t(e, k) + 2 * e4t(e, k + 1) + e4t(e, k + 2) + 2) >> 2 : (e4t(e, k) + e4t(e, k + 1) + 1) >> 1;
    break; }
  default: {
    const int z = x + 2 * y, k = y + (x >> 1);
    if (z > 5) v = e4l(e, 3);
    else if (z == 5) v = (e4l(e, 2) + 3 * e4l(e, 3) + 2) >> 2;
    else if (z & 1) v = (e4l(e, k) + 2 * e4l(e, k + 1) + e4l(e, k + 2) + 2) >> 2;
    else v = (e4l(e, k) + e4l(e, k + 1) + 1) >> 1;
    break; }
  }
  out[(long)b * 16 + 4 * y + x] = (uint8_t)v;
}

__device__ __forceinline__ int iabsi_(int v) { return v < 0 ? -v : v; }
// JM's hadamard4x4 (transform.c:121-168): rows, then columns with >> 1; in place on a row-major 4x4
__device__ __forceinline__ void hadamard4x4_jm(int (&m)[16])
{
  int u[16];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int e0 = m[4 * i] + m[4 * i + 3], e1 = m[4 * i + 1] + m[4 * i + 2], o0 = m[4 * i + 1] - m[4 * i + 2], o1 = m[4 * i] - m[4 * i + 3];
    u[4 * i] = e0 + e1; u[4 * i + 1] = o1 + o0; u[4 * i + 2] = e0 - e1; u[4 * i + 3] = o1 - o0;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int e0 = u[i] + u[12 + i], e1 = u[4 + i] + u[8 + i], o0 = u[4 + i] - u[8 + i], o1 = u[i] - u[12 + i];
    m[i] = (e0 + e1) >> 1; m[4 + i] = (o0 + o1) >> 1; m[8 + i] = (e0 - e1) >> 1; m[12 + i] = (o1 - o0) >> 1;
  }
}

__global__ __launch_bounds__(256) void k_intra16_search(const jmhip_i16_mb *__restrict__ mbs, const uint8_t *__restrict__ orig, int n, int max_pel,
                                                        jmhip_i16_out *__restrict__ out)
{
  const int lane = threadIdx.x & 63, b = lane & 15, gb = lane & ~15;
  const int item = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = item < n;
  const int it = live ? item : 0;
  const jmhip_i16_mb *mb = mbs + it;
  const uint8_t *e = mb->edge;
  const int jj = b >> 2, ii = b & 3, left = mb->left, up = mb->up, mask = mb->mode_mask, metric = mb->metric;
  // the block's source samples
  int src[16];
  {
    const uint8_t *po = orig + (long)it * 256 + (4 * jj) * 16 + 4 * ii;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t w = *(const uint32_t *)(po + 16 * j);
#pragma unroll
      for (int i = 0; i < 4; i++) src[4 * j + i] = (w >> (8 * i)) & 255;
    }
  }
  // uniform parameters of the DC and plane modes
  int dc, pa, pb, pc;
  {
    int s1 = 0, s2 = 0, H = 0, V = 0;
#pragma unroll
    for (int x = 0; x < 16; x++) { s1 += e[1 + x]; s2 += e[17 + x]; }
    dc = (up && left) ? (s1 + s2 + 16) >> 5 : (up ? (s1 + 8) >> 4 : (left ? (s2 + 8) >> 4 : e[1]));
#pragma unroll
    for (int x = 0; x < 8; x++) {
      H += (x + 1) * ((int)e[9 + x] - (int)(x == 7 ? e[0] : e[7 - x]));
      V += (x + 1) * ((int)e[25 + x] - (int)(x == 7 ? e[0] : e[23 - x]));
    }
    pb = (5 * H + 32) >> 6; pc = (5 * V + 32) >> 6; pa = 16 * (e[16] + e[32]);
  }
  long long best = 0x7fffffffLL << 5;                       // DISTBLK_MAX
  int best_mode = 2;
#pragma unroll 1
  for (int k = 0; k < 4; k++) {
    if (!((mask >> k) & 1)) continue;                       // uniform inside the 16-lane group, and the shuffles below stay inside it
    int m[16];
    uint32_t pw[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      pw[j] = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int x = 4 * ii + i, y = 4 * jj + j;
        int v;
        if (k == 0) v = e[1 + x];
        else if (k == 1) v = e[17 + y];
        else if (k == 2) v = dc;
        else { v = (pa + pb * (x - 7) + pc * (y - 7) + 16) >> 5; v = v < 0 ? 0 : (v > max_pel ? max_pel : v); }
        pw[j] |= (uint32_t)v << (8 * i);
        m[4 * j + i] = src[4 * j + i] - v;
      }
    }
    if (live) {
#pragma unroll
      for (int j = 0; j < 4; j++) *(uint32_t *)(out[it].pred[k] + (4 * jj + j) * 16 + 4 * ii) = pw[j];
    }
    long long part = 0;
    if (metric == JMHIP_METRIC_SAD) {
#pragma unroll
      for (int c = 0; c < 16; c++) part += iabsi_(m[c]);
    } else if (metric == JMHIP_METRIC_SSE) {
#pragma unroll
      for (int c = 0; c < 16; c++) part += m[c] * m[c];
    } else {
      hadamard4x4_jm(m);
#pragma unroll
      for (int c = 1; c < 16; c++) part += iabsi_(m[c]);
      int t[16];
#pragma unroll
      for (int q = 0; q < 16; q++) t[q] = __shfl(m[0] >> 1, gb + q, 64);
      hadamard4x4_jm(t);
      int v = 0;
#pragma unroll
      for (int w = 0; w < 16; w++) v = (w == b) ? t[w] : v;      // each lane adds one term of the DC transform
      part += iabsi_(v);
    }
    int lo = (int)part;                                    // a block's share stays far below 2^31 (SSE: 16 * 255^2)
    lo += __shfl_xor(lo, 1, 64); lo += __shfl_xor(lo, 2, 64); lo += __shfl_xor(lo, 4, 64); lo += __shfl_xor(lo, 8, 64);
    const long long cost = (long long)lo << 5;
    if (cost < best) { best = cost; best_mode = k; }
  }
  if (live && b == 0) { out[it].cost = best; out[it].mode = best_mode; out[it].reserved_ = 0; }
}

extern "C" int jmhip_intrapred4x4(jmhip_ctx *ctx, const jmhip_ip4_blk *blks, int32_t n, uint8_t *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || (n > 0 && (!blks || !out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_intrapred4x4: bad argument");
  for (int i = 0; i < n; i++) if (blks[i].mode > 8) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_intrapred4x4: block %d: mode %d", i, blks[i].mode);
  if (n == 0) return JMHIP_OK;
  int r; void *din, *dout;
  if ((r = jmhip_scratch(ctx, 0, (size_t)n * sizeof(jmhip_ip4_blk), &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * 16, &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(din, blks, (size_t)n * sizeof(jmhip_ip4_blk), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_intrapred4x4, dim3((n + 15) / 16), dim3(256), 0, ctx->stream, (const jmhip_ip4_blk *)din, n, (uint8_t *)dout);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_intra16_search_dev(jmhip_ctx *ctx, const jmhip_i16_mb *d_mbs, const uint8_t *d_orig, int32_t n, jmhip_i16_out *d_out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || (n > 0 && (!d_mbs || !d_orig || !d_out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_intra16_search_dev: bad argument");
  if (n == 0) return JMHIP_OK;
  hipLaunchKernelGGL(k_intra16_search, dim3((n + 15) / 16), dim3(256), 0, ctx->stream, d_mbs, d_orig, n, 255, d_out);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_intra16_search(jmhip_ctx *ctx, const jmhip_i16_mb *mbs, const uint8_t *orig, int32_t n, jmhip_i16_out *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || (n > 0 && (!mbs || !orig || !out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_intra16_search: bad argument");
  for (int i = 0; i < n; i++)
    if (mbs[i].mode_mask > 15 || mbs[i].metric > 2) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_intra16_search: macroblock %d: mode mask %d, metric %d", i, mbs[i].mode_mask, mbs[i].metric);
  if (n == 0) return JMHIP_OK;
  int r; void *din, *dout;
  if ((r = jmhip_scratch(ctx, 0, (size_t)n * (sizeof(jmhip_i16_mb) + 256) + 64, &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * sizeof(jmhip_i16_out), &dout))) return r;
  uint8_t *d_orig = (uint8_t *)din;
  jmhip_i16_mb *d_mbs = (jmhip_i16_mb *)(d_orig + (size_t)n * 256);
  HIPCHK(ctx, hipMemcpyAsync(d_orig, orig, (size_t)n * 256, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_mbs, mbs, (size_t)n * sizeof(jmhip_i16_mb), hipMemcpyHostToDevice, ctx->stream));
  if ((r = jmhip_intra16_search_dev(ctx, d_mbs, d_orig, n, (jmhip_i16_out *)dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * sizeof(jmhip_i16_out), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}


// ---- chroma: intra_chroma_prediction (lencod/src/intra_chroma.c:530-778, frame macroblocks), all four modes of both planes of a macroblock.
// 64 lanes per macroblock: lane -> (plane, sample row j < 16, half row of four samples); a lane writes its four samples of every mode.
// DC per 4x4 block by the block's place (:590-686): top-left and bottom-right use both neighbour sums, top-right prefers the samples above,
// bottom-left the samples to the left; 128 without neighbours.  Modes whose neighbours are missing are written as zeros (the reference
// leaves them alone).
static_assert(sizeof(jmhip_ic_mb) == 56, "jmhip_ic_mb is 56 bytes in include/jmhip.h");
__global__ __launch_bounds__(256) void k_intra_chroma(const jmhip_ic_mb *__restrict__ mbs, int n, int ch, uint8_t *__restrict__ out)
{
  const int t = blockIdx.x * 256 + threadIdx.x, b = t >> 6, l = t & 63, uv = l >> 5, j = (l >> 1) & 15, bx = (l & 1) * 4;
  if (b >= n) return;
  const jmhip_ic_mb *q = mbs + b;
  const uint8_t *up = q->up[uv], *left = q->left[uv];
  const int ua = q->up_avail, la = q->left_avail, da = q->upleft_avail, corner = q->corner[uv];
  uint8_t *o = out + (long)b * 1024 + uv * 128 + j * 8 + bx;                 // out[b][mode][plane][16 x 8]
  uint32_t dc = 0, hor = 0, ver = 0, pl = 0;
  if (j < ch) {
    const int by = j & ~3;
    int su = 0, sl = 0, s = 128;
#pragma unroll
    for (int i = 0; i < 4; i++) { su += up[bx + i]; sl += left[by + i]; }
    if ((by == 0) == (bx == 0)) s = (ua && la) ? (su + sl + 4) >> 3 : (ua ? (su + 2) >> 2 : (la ? (sl + 2) >> 2 : 128));
    else if (by == 0) s = ua ? (su + 2) >> 2 : (la ? (sl + 2) >> 2 : 128);
    else s = la ? (sl + 2) >> 2 : (ua ? (su + 2) >> 2 : 128);
    dc = 0x01010101u * (uint32_t)s;
    if (la) hor = 0x01010101u * (uint32_t)left[j];
    if (ua) ver = (uint32_t)up[bx] | ((uint32_t)up[bx + 1] << 8) | ((uint32_t)up[bx + 2] << 16) | ((uint32_t)up[bx + 3] << 24);
    if (ua && la && da) {
      const int cr_y = ch >> 1;
      int ih = 4 * ((int)up[7] - corner), iv = cr_y * ((int)left[ch - 1] - corner);
#pragma unroll
      for (int i = 0; i < 3; i++) ih += (i + 1) * ((int)up[4 + i] - (int)up[2 - i]);
      for (int i = 0; i < cr_y - 1; i++) iv += (i + 1) * ((int)left[cr_y + i] - (int)left[cr_y - 2 - i]);
      const int ib = (17 * ih + 16) >> 5, ic = ch == 8 ? (17 * iv + 16) >> 5 : (5 * iv + 32) >> 6;
      const int iaa = 16 * ((int)up[7] + (int)left[ch - 1]) - 3 * ib + (1 - cr_y) * ic;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        // the shifted value is pinned before the clamp: left alone, hipcc (ROCm 7.2) fuses the first two "clamp(x >> 5, 0, 255)" into one
        // v_ashr_pk_u8_i32 and ORs the other two bytes into bits 16..31 of its result -- which the instruction does not clear on the
        // MI355X (it keeps what the destination register held: here the first source, 0xffff.... when that was negative), so samples
        // 2 and 3 came out as 255 (profiles/microbench/ashr_pk_u8.hip).  The only place in the library where the pattern occurs.
        int v = (iaa + (bx + i) * ib + j * ic + 16) >> 5;
        asm volatile("" : "+v"(v));
        pl |= (uint32_t)min(max(v, 0), 255) << (8 * i);
      }
    }
  }
  *(uint32_t *)(o + 0 * 256) = dc; *(uint32_t *)(o + 1 * 256) = hor; *(uint32_t *)(o + 2 * 256) = ver; *(uint32_t *)(o + 3 * 256) = pl;
}

extern "C" int jmhip_intra_chroma_dev(jmhip_ctx *ctx, const jmhip_ic_mb *d_mbs, int32_t n, uint8_t *d_out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || (n > 0 && (!d_mbs || !d_out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_intra_chroma_dev: bad argument");
  if (ctx->cfg.yuv_format != 1 && ctx->cfg.yuv_format != 2) return jmhip_fail(ctx, JMHIP_EUNSUPPORTED, "jmhip_intra_chroma: yuv_format %d (4:2:0 and 4:2:2 only)", ctx->cfg.yuv_format);
  if (n == 0) return JMHIP_OK;
  hipLaunchKernelGGL(k_intra_chroma, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, d_mbs, n, ctx->cfg.yuv_format == 2 ? 16 : 8, d_out);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_intra_chroma(jmhip_ctx *ctx, const jmhip_ic_mb *mbs, int32_t n, uint8_t *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || (n > 0 && (!mbs || !out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_intra_chroma: bad argument");
  if (ctx->cfg.yuv_format != 1 && ctx->cfg.yuv_format != 2) return jmhip_fail(ctx, JMHIP_EUNSUPPORTED, "jmhip_intra_chroma: yuv_format %d (4:2:0 and 4:2:2 only)", ctx->cfg.yuv_format);
  if (n == 0) return JMHIP_OK;
  int r; void *din, *dout;
  if ((r = jmhip_scratch(ctx, 0, (size_t)n * sizeof(jmhip_ic_mb), &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * 1024, &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(din, mbs, (size_t)n * sizeof(jmhip_ic_mb), hipMemcpyHostToDevice, ctx->stream));
  if ((r = jmhip_intra_chroma_dev(ctx, (const jmhip_ic_mb *)din, n, (uint8_t *)dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * 1024, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}


// ---- Intra8x8: get_intrapred_8x8 (lencod/src/intra8x8.c:716-760, the nine modes :148-495) from the 25 low-pass filtered predictor samples
// (Z, A..P, Q..X: LowPassForIntra8x8Pred :85-140 is applied by set_intrapred_8x8, which stays with the caller).  64 lanes per block, one
// sample each; H.264 8.3.2.2.2 - 8.3.2.2.10 with p'[x,-1] = e8t(x), p'[-1,y] = e8l(y).
static_assert(sizeof(jmhip_ip8_blk) == 28, "jmhip_ip8_blk is 28 bytes in include/jmhip.h");
__device__ __forceinline__ int e8t(const uint8_t *e, int x) { return x < 0 ? e[0] : e[1 + x]; }      // x = -1 .. 15
__device__ __forceinline__ int e8l(const uint8_t *e, int y) { return y < 0 ? e[0] : e[17 + y]; }     // y = -1 .. 7
__global__ __launch_bounds__(256) void k_intrapred8x8(const jmhip_ip8_blk *__restrict__ blks, int n, uint8_t *__restrict__ out)
{
  const int t = blockIdx.x * 256 + threadIdx.x, b = t >> 6, x = t & 7, y = (t >> 3) & 7;
  if (b >= n) return;
  const uint8_t *e = blks[b].edge;
  const int mode = blks[b].mode, left = blks[b].left, up = blks[b].up;
  int v;
  switch (mode) {
  case 0: v = e8t(e, x); break;
  case 1: v = e8l(e, y); break;
  case 2: {
    int su = 0, sl = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { su += e[1 + i]; sl += e[17 + i]; }
    v = (up && left) ? (su + sl + 8) >> 4 : (left ? (sl + 4) >> 3 : (up ? (su + 4) >> 3 : e[1]));
    break; }
  case 3: v = (x == 7 && y == 7) ? (e8t(e, 14) + 3 * e8t(e, 15) + 2) >> 2 : (e8t(e, x + y) + 2 * e8t(e, x + y + 1) + e8t(e, x + y + 2) + 2) >> 2; break;
  case 4:
    if (x > y) v = (e8t(e, x - y - 2) + 2 * e8t(e, x - y - 1) + e8t(e, x - y) + 2) >> 2;
    else if (x < y) v = (e8l(e, y - x - 2) + 2 * e8l(e, y - x - 1) + e8l(e, y - x) + 2) >> 2;
    else v = (e8t(e, 0) + 2 * e[0] + e8l(e, 0) + 2) >> 2;
    break;
  case 5: {
    const int z = 2 * x - y, k = x - (y >> 1);
    if (z >= 0 && !(z & 1)) v = (e8t(e, k - 1) + e8t(e, k) + 1) >> 1;
    else if (z > 0) v = (e8t(e, k - 2) + 2 * e8t(e, k - 1) + e8t(e, k) + 2) >> 2;
    else if (z == -1) v = (e8l(e, 0) + 2 * e[0] + e8t(e, 0) + 2) >> 2;
    else v = (e8l(e, y - 2 * x - 1) + 2 * e8l(e, y - 2 * x - 2) + e8l(e, y - 2 * x - 3) + 2) >> 2;
    break; }
  case 6: {
    const int z = 2 * y - x, k = y - (x >> 1);
    if (z >= 0 && !(z & 1)) v = (e8l(e, k - 1) + e8l(e, k) + 1) >> 1;
    else if (z > 0) v = (e8l(e, k - 2) + 2 * e8l(e, k - 1) + e8l(e, k) + 2) >> 2;
    else if (z == -1) v = (e8l(e, 0) + 2 * e[0] + e8t(e, 0) + 2) >> 2;
    else v = (e8t(e, x - 2 * y - 1) + 2 * e8t(e, x - 2 * y - 2) + e8t(e, x - 2 * y - 3) + 2) >> 2;
    break; }
  case 7: {
    const int k = x + (y >> 1);
    v = (y & 1) ? (e8t(e, k) + 2 * e8t(e, k + 1) + e8t(e, k + 2) + 2) >> 2 : (e8t(e, k) + e8t(e, k + 1) + 1) >> 1;
    break; }
  default: {
    const int z = x + 2 * y, k = y + (x >> 1);
    if (z > 13) v = e8l(e, 7);
    else if (z == 13) v = (e8l(e, 6) + 3 * e8l(e, 7) + 2) >> 2;
    else if (z & 1) v = (e8l(e, k) + 2 * e8l(e, k + 1) + e8l(e, k + 2) + 2) >> 2;
    else v = (e8l(e, k) + e8l(e, k + 1) + 1) >> 1;
    break; }
  }
  out[(long)b * 64 + y * 8 + x] = (uint8_t)v;
}

extern "C" int jmhip_intrapred8x8(jmhip_ctx *ctx, const jmhip_ip8_blk *blks, int32_t n, uint8_t *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || (n > 0 && (!blks || !out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_intrapred8x8: bad argument");
  for (int i = 0; i < n; i++) if (blks[i].mode > 8) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_intrapred8x8: block %d: mode %d", i, blks[i].mode);
  if (n == 0) return JMHIP_OK;
  int r; void *din, *dout;
  if ((r = jmhip_scratch(ctx, 0, (size_t)n * sizeof(jmhip_ip8_blk), &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * 64, &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(din, blks, (size_t)n * sizeof(jmhip_ip8_blk), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_intrapred8x8, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, (const jmhip_ip8_blk *)din, n, (uint8_t *)dout);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}
