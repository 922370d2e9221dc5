// tq16.hip -- Intra16x16 luma residual transform / quantisation / reconstruction of whole macroblocks (gfx950).
//
// Device counterpart of residual_transform_quant_luma_16x16 (reference: lencod/src/block.c:208-349) with
//   forward4x4 / inverse4x4            lcommon/src/transform.c:20-118
//   hadamard4x4 / ihadamard4x4         transform.c:121-220
//   quant_dc4x4_normal                 lencod/src/quant4x4_normal.c:200-259 (bound with and without adaptive rounding, quant4x4.c:56-63)
//   quant_ac4x4_normal / _around       quant4x4_normal.c:117-191 / quant4x4_around.c:129-213
//   sample_reconstruct, DQ_BITS 6      lencod/src/blk_prediction.c:48-62
//
// Sixteen lanes per macroblock, lane b = 4x4 block b in raster order: the block's coefficients live in registers; the sixteen DC
// coefficients are exchanged with wave shuffles and every lane runs the (small) DC path itself, keeping its own dequantised DC;
// "any AC level" is a ballot.  Algorithmic bytes per macroblock: 512 in + 1168 out.
// JM hands quant_ac4x4_around the adaptive-rounding array of the I16MB mode WITHOUT the block's row offset (block.c:247): every
// block row writes rows 0..3 of it and the last one wins.  The kernel returns exactly those rows (fadjust[4][16]: block row 3).
#include "jmhip_internal.h"
static_assert(sizeof(jmhip_tq16_out) == 1224, "jmhip_tq16_out layout");

__device__ __forceinline__ int iabs16_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ void fwd4h(int &a, int &b, int &c, int &d)
{
  const int e0 = a + d, e1 = b + c, o0 = b - c, o1 = a - d;
  a = e0 + e1; b = (o1 << 1) + o0; c = e0 - e1; d = o1 - (o0 << 1);
}
__device__ __forceinline__ void inv4h(int &a, int &b, int &c, int &d)
{
  const int e0 = a + c, e1 = a - c, o0 = (b >> 1) - d, o1 = b + (d >> 1);
  a = e0 + o1; b = e1 + o0; c = e1 - o0; d = e0 - o1;
}

__global__ __launch_bounds__(256) void k_tq_luma16x16(jmhip_tq_params prm, const uint8_t *__restrict__ orig, const uint8_t *__restrict__ pred,
                                                      int n, jmhip_tq16_out *__restrict__ out)
{
  const int lane = threadIdx.x & 63, b = lane & 15, gb = lane & ~15;
  const int item = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = item < n;
  const int it = live ? item : 0;
  const int jj = b >> 2, ii = b & 3;
  jmhip_tq16_out *o = out + it;

  // ---- residual and forward transform of the lane's block
  int m[16], pr[16];
  {
    const uint8_t *po = orig + (long)it * 256 + (4 * jj) * 16 + 4 * ii, *pp = pred + (long)it * 256 + (4 * jj) * 16 + 4 * ii;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t wo = *(const uint32_t *)(po + 16 * j), wp = *(const uint32_t *)(pp + 16 * j);
#pragma unroll
      for (int i = 0; i < 4; i++) { pr[4 * j + i] = (wp >> (8 * i)) & 255; m[4 * j + i] = (int)((wo >> (8 * i)) & 255) - pr[4 * j + i]; }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) fwd4h(m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]);
#pragma unroll
  for (int i = 0; i < 4; i++) fwd4h(m[i], m[4 + i], m[8 + i], m[12 + i]);

  // ---- DC path on the macroblock's sixteen DC coefficients (every lane, redundantly)
  int t[16];
#pragma unroll
  for (int q = 0; q < 16; q++) t[q] = __shfl(m[0], gb + q, 64);
  {                                                        // hadamard4x4: rows, then columns with >> 1 (transform.c:121-168)
    int u[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int e0 = t[4 * i] + t[4 * i + 3], e1 = t[4 * i + 1] + t[4 * i + 2], o0 = t[4 * i + 1] - t[4 * i + 2], o1 = t[4 * i] - t[4 * i + 3];
      u[4 * i] = e0 + e1; u[4 * i + 1] = o1 + o0; u[4 * i + 2] = e0 - e1; u[4 * i + 3] = o1 - o0;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int e0 = u[i] + u[12 + i], e1 = u[4 + i] + u[8 + i], o0 = u[4 + i] - u[8 + i], o1 = u[i] - u[12 + i];
      t[i] = (e0 + e1) >> 1; t[4 + i] = (o0 + o1) >> 1; t[8 + i] = (e0 - e1) >> 1; t[12 + i] = (o1 - o0) >> 1;
    }
  }
  constexpr int ZZ[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
  int dcl[17], dcr[17], ndc = 0;
  {                                                        // quant_dc4x4_normal: levels stay in t[]
    const int q_bits = 15 + prm.qp_per + 1;
    int run = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int idx = ZZ[k], c = t[idx];
      if (c != 0) {
        int l = (iabs16_(c) * prm.q[0].ScaleComp + (prm.q[0].OffsetComp << 1)) >> q_bits;
        if (l != 0) {
          if (prm.cavlc) l = min(l, 2063);
          l = c < 0 ? -l : l;
          t[idx] = l;
#pragma unroll
          for (int w = 0; w < 16; w++) if (w == ndc) { dcl[w] = l; dcr[w] = run; }
          ndc++; run = 0;
        } else { t[idx] = 0; run++; }
      } else run++;
    }
  }
  int mydc = 0;
  if (ndc) {                                               // ihadamard4x4 (transform.c:170-220), then the DC's own dequantisation (block.c:294)
    int u[16], r[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int e0 = t[4 * i] + t[4 * i + 2], e1 = t[4 * i] - t[4 * i + 2], o0 = t[4 * i + 1] - t[4 * i + 3], o1 = t[4 * i + 1] + t[4 * i + 3];
      u[4 * i] = e0 + o1; u[4 * i + 1] = e1 + o0; u[4 * i + 2] = e1 - o0; u[4 * i + 3] = e0 - o1;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int e0 = u[i] + u[8 + i], e1 = u[i] - u[8 + i], o0 = u[4 + i] - u[12 + i], o1 = u[4 + i] + u[12 + i];
      r[i] = e0 + o1; r[4 + i] = e1 + o0; r[8 + i] = e1 - o0; r[12 + i] = e0 - o1;
    }
    int v = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) v = (w == b) ? r[w] : v;
    mydc = (((v * prm.q[0].InvScaleComp) << prm.qp_per) + 32) >> 6;
  }
  m[0] = mydc;

  // ---- AC quantisation of the lane's block (coefficients 1..15 of the zig-zag)
  int nz = 0, ncoef = 0;
  int16_t alev[16]; uint8_t arun[16]; int16_t fadj[16];
#pragma unroll
  for (int c = 0; c < 16; c++) { alev[c] = 0; arun[c] = 0; fadj[c] = 0; }
  {
    const int q_bits = 15 + prm.qp_per;
    int run = 0;
#pragma unroll
    for (int c = 1; c < 16; c++) {
      const int idx = ZZ[c], v = m[idx];
      if (v != 0) {
        const int scaled = iabs16_(v) * prm.q[idx].ScaleComp;
        int l = (scaled + prm.q[idx].OffsetComp) >> q_bits;
        if (l != 0) {
          if (prm.cavlc) l = min(l, 2063);
          if (prm.adaptive_rounding) fadj[idx] = (int16_t)((prm.adapt_rnd_weight * (scaled - (l << q_bits)) + (1 << q_bits)) >> (q_bits + 1));
          l = v < 0 ? -l : l;
          m[idx] = (((l * prm.q[idx].InvScaleComp) << prm.qp_per) + 8) >> 4;
#pragma unroll
          for (int w = 0; w < 16; w++) if (w == ncoef) { alev[w] = (int16_t)l; arun[w] = (uint8_t)run; }
          ncoef++; run = 0; nz = 1;
        } else { m[idx] = 0; run++; }
      } else run++;
    }
  }
  const bool any_ac = ((__ballot(nz != 0) >> gb) & 0xffffull) != 0;
  if (m[0] != 0 || nz) {
#pragma unroll
    for (int i = 0; i < 4; i++) inv4h(m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]);
#pragma unroll
    for (int i = 0; i < 4; i++) inv4h(m[i], m[4 + i], m[8 + i], m[12 + i]);
  }
  if (!live) return;
  // ---- outputs: reconstruction, the block's AC list at JM's cofAC index b8 * 4 + b4, the rows of fadjust JM ends up with
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int v = pr[4 * j + i] + ((m[4 * j + i] + 32) >> 6);
      v = v < 0 ? 0 : (v > prm.max_pel ? prm.max_pel : v);
      w |= (uint32_t)v << (8 * i);
    }
    *(uint32_t *)(o->rec + (4 * jj + j) * 16 + 4 * ii) = w;
  }
  {
    const int cb = (2 * (jj >> 1) + (ii >> 1)) * 4 + 2 * (jj & 1) + (ii & 1);
#pragma unroll
    for (int c = 0; c < 16; c++) { o->ac_level[cb][c] = alev[c]; o->ac_run[cb][c] = arun[c]; }
    o->ac_ncoef[cb] = (uint8_t)ncoef;
  }
  if (jj == 3) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int i = 0; i < 4; i++) o->fadjust[j][4 * ii + i] = fadj[4 * j + i];
  }
  if (b == 0) {
#pragma unroll
    for (int w = 0; w < 17; w++) { o->dc_level[w] = (int16_t)(w < ndc ? dcl[w < 16 ? w : 0] : 0); o->dc_run[w] = (uint8_t)(w < ndc ? dcr[w < 16 ? w : 0] : 0); }
    o->dc_nonzero = (uint8_t)(ndc != 0);
    o->ac_coef = (uint8_t)(any_ac ? 15 : 0);
    o->reserved_[0] = o->reserved_[1] = o->reserved_[2] = 0;
  }
}

static int tq16_check(jmhip_ctx *ctx, const jmhip_tq_params *prm, const void *a, const void *b, const void *c, int n, const char *who)
{
  if (!prm || n < 0 || (n > 0 && (!a || !b || !c))) return jmhip_fail(ctx, JMHIP_EINVAL, "%s: bad argument", who);
  if (prm->qp_per < 0 || prm->qp_per > 8) return jmhip_fail(ctx, JMHIP_EINVAL, "%s: qp_per %d outside 0..8", who, prm->qp_per);
  return JMHIP_OK;
}

extern "C" int jmhip_tq_luma16x16_dev(jmhip_ctx *ctx, const jmhip_tq_params *prm, const uint8_t *d_orig, const uint8_t *d_pred, int32_t n, jmhip_tq16_out *d_out)
{
  if (!ctx) return JMHIP_EINVAL;
  int r = tq16_check(ctx, prm, d_orig, d_pred, d_out, n, "jmhip_tq_luma16x16_dev");
  if (r || n == 0) return r;
  hipLaunchKernelGGL(k_tq_luma16x16, dim3((n + 15) / 16), dim3(256), 0, ctx->stream, *prm, d_orig, d_pred, n, d_out);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_tq_luma16x16(jmhip_ctx *ctx, const jmhip_tq_params *prm, const uint8_t *orig, const uint8_t *pred, int32_t n, jmhip_tq16_out *out)
{
  if (!ctx) return JMHIP_EINVAL;
  int r = tq16_check(ctx, prm, orig, pred, out, n, "jmhip_tq_luma16x16");
  if (r || n == 0) return r;
  void *din, *dout;
  if ((r = jmhip_scratch(ctx, 0, (size_t)n * 512, &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * sizeof(jmhip_tq16_out), &dout))) return r;
  uint8_t *d_orig = (uint8_t *)din, *d_pred = d_orig + (size_t)n * 256;
  HIPCHK(ctx, hipMemcpyAsync(d_orig, orig, (size_t)n * 256, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_pred, pred, (size_t)n * 256, hipMemcpyHostToDevice, ctx->stream));
  if ((r = jmhip_tq_luma16x16_dev(ctx, prm, d_orig, d_pred, n, (jmhip_tq16_out *)dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * sizeof(jmhip_tq16_out), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}
