// tq_chroma.hip -- chroma residual transform / quantisation / reconstruction of one plane of a macroblock (gfx950).
//
// Device counterpart of residual_transform_quant_chroma_4x4 (reference: lencod/src/block.c:954-1200) with
//   forward4x4 / inverse4x4                        lcommon/src/transform.c:20-118
//   hadamard2x2 / ihadamard2x2 / hadamard4x2 / ihadamard4x2   transform.c:220-330
//   quant_dc2x2_normal / quant_dc4x2_normal        lencod/src/quantChroma_normal.c:37 / :110   (the _around variants are identical)
//   quant_ac4x4_normal / quant_ac4x4_around        lencod/src/quant4x4_normal.c:117 / quant4x4_around.c:129
//   SCAN_YUV422, cbp_blk_chroma, hor/ver_offset    block.c:88-166 (block k = raster order of the plane's 4x4 blocks)
//   _CHROMA_COEFF_COST_ 4                          lencod/inc/defines.h:115
//
// Eight lanes per item, lane k = 4x4 block k (4:2:0 uses four of them): the block's 16 coefficients live in registers; the DC
// coefficients of the plane are exchanged with wave shuffles and every lane runs the (tiny) DC path itself; the coefficient cost
// and the "which blocks are coded" bits are group-wide reductions (shuffle / ballot).  Algorithmic bytes per item: 256 in + 808 out.
#include "jmhip_internal.h"

__device__ __forceinline__ int iabsc_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ void fwd4c(int &a, int &b, int &c, int &d)
{
  const int e0 = a + d, e1 = b + c, o0 = b - c, o1 = a - d;
  a = e0 + e1; b = (o1 << 1) + o0; c = e0 - e1; d = o1 - (o0 << 1);
}
__device__ __forceinline__ void inv4c(int &a, int &b, int &c, int &d)
{
  const int e0 = a + c, e1 = a - c, o0 = (b >> 1) - d, o1 = b + (d >> 1);
  a = e0 + o1; b = e1 + o0; c = e1 - o0; d = e0 - o1;
}

__global__ __launch_bounds__(256) void k_tq_chroma(jmhip_tqc_params prm, jmhip_tqc_mb *__restrict__ mbs, const uint8_t *__restrict__ orig,
                                                   const uint8_t *__restrict__ pred, int n, jmhip_tqc_out *__restrict__ out)
{
  const int lane = threadIdx.x & 63, k = lane & 7, gb = lane & ~7;
  const int item = blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool live = item < n;
  const int it = live ? item : 0;
  const int yuv = prm.yuv_format, nblk = yuv == 2 ? 8 : 4;
  const bool blk = k < nblk;                              // this lane owns a 4x4 block
  const int n1 = 4 * (k & 1), n2 = 4 * (k >> 1);
  const jmhip_tqc_mb mb = mbs[it];
  const int uv = mb.uv;
  jmhip_tqc_out *o = out + it;

  // ---- residual and forward transform of the lane's block
  int m[16], pr[16], z = 0;
  {
    const uint8_t *po = orig + (long)it * 128 + n2 * 8 + n1, *pp = pred + (long)it * 128 + n2 * 8 + n1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t wo = blk ? *(const uint32_t *)(po + 8 * j) : 0u, wp = blk ? *(const uint32_t *)(pp + 8 * j) : 0u;
#pragma unroll
      for (int i = 0; i < 4; i++) { pr[4 * j + i] = (wp >> (8 * i)) & 255; m[4 * j + i] = (int)((wo >> (8 * i)) & 255) - pr[4 * j + i]; z |= m[4 * j + i]; }
    }
  }
  if (z) {                                                // check_zero blocks stay zero (block.c:1017-1027)
#pragma unroll
    for (int i = 0; i < 4; i++) fwd4c(m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]);
#pragma unroll
    for (int i = 0; i < 4; i++) fwd4c(m[i], m[4 + i], m[8 + i], m[12 + i]);
  }
  // ---- DC path, run by every lane on the group's eight DC coefficients
  int dc[8];
#pragma unroll
  for (int q = 0; q < 8; q++) dc[q] = __shfl(m[0], gb + q, 64);
  int t[8], lev[8], dcl[9], dcr[9], ndc = 0, dczero = 0;
  if (yuv == 1) {                                         // hadamard2x2 of {dc00, dc01, dc10, dc11} = blocks 0..3 (transform.c:301)
    const int p0 = dc[0] + dc[1], p1 = dc[0] - dc[1], p2 = dc[2] + dc[3], p3 = dc[2] - dc[3];
    t[0] = p0 + p2; t[1] = p1 + p3; t[2] = p0 - p2; t[3] = p1 - p3; t[4] = t[5] = t[6] = t[7] = 0;
  } else {                                                // hadamard4x2 of tblk[a][b] = DC of block (row group b, column group a) (transform.c:220)
    int in[8], s8[8];
#pragma unroll
    for (int b = 0; b < 4; b++) { in[b] = dc[2 * b]; in[4 + b] = dc[2 * b + 1]; }
#pragma unroll
    for (int i = 0; i < 4; i++) { s8[i] = in[i] + in[4 + i]; s8[4 + i] = in[i] - in[4 + i]; }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int t0 = s8[4 * i] + s8[4 * i + 3], t1 = s8[4 * i + 1] + s8[4 * i + 2], t2 = s8[4 * i + 1] - s8[4 * i + 2], t3 = s8[4 * i] - s8[4 * i + 3];
      t[4 * i] = t0 + t1; t[4 * i + 1] = t3 + t2; t[4 * i + 2] = t0 - t1; t[4 * i + 3] = t3 - t2;
    }
  }
  {
    const int q_bits = 15 + prm.qp_per_dc + 1;
    constexpr int S422[8] = {0, 1, 4, 2, 3, 5, 6, 7};     // SCAN_YUV422 (j, i) -> j * 4 + i (block.c:88)
    int run = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      lev[q] = 0;
      if (q < (yuv == 1 ? 4 : 8)) {
        // unrolled with a literal index per format so t[] stays in registers
        const int idx = yuv == 1 ? q : S422[q];
        int c = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) c = (w == idx) ? t[w] : c;
        int nv = c;
        if (c != 0) {
          int l = (iabsc_(c) * prm.q_dc.ScaleComp + (prm.q_dc.OffsetComp << 1)) >> q_bits;
          if (l != 0) {
            if (prm.cavlc) l = min(l, 2063);
            l = c < 0 ? -l : l;
            nv = (l * prm.q_dc.InvScaleComp) << prm.qp_per_dc;
#pragma unroll
            for (int w = 0; w < 9; w++) if (w == ndc) { dcl[w] = l; dcr[w] = run; }
            ndc++; run = 0; dczero = 1;
          } else { nv = 0; run++; }
        } else run++;
#pragma unroll
        for (int w = 0; w < 8; w++) t[w] = (w == idx) ? nv : t[w];
      }
    }
  }
  int mydc;
  if (yuv == 1) {                                         // ihadamard2x2, >> 5 (block.c:1048-1054)
    const int t0 = t[0] + t[1], t1 = t[0] - t[1], t2 = t[2] + t[3], t3 = t[2] - t[3];
    const int r[4] = {(t0 + t2) >> 5, (t1 + t3) >> 5, (t0 - t2) >> 5, (t1 - t3) >> 5};
    mydc = r[k & 3];
  } else {                                                // ihadamard4x2 (result transposed back), (x + 32) >> 6 (block.c:1083-1092)
    int s8[8], r[8];
#pragma unroll
    for (int i = 0; i < 4; i++) { s8[i] = t[i] + t[4 + i]; s8[4 + i] = t[i] - t[4 + i]; }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int t0 = s8[4 * i] + s8[4 * i + 2], t1 = s8[4 * i] - s8[4 * i + 2], t2 = s8[4 * i + 1] - s8[4 * i + 3], t3 = s8[4 * i + 1] + s8[4 * i + 3];
      r[i] = t0 + t3; r[2 + i] = t1 + t2; r[4 + i] = t1 - t2; r[6 + i] = t0 - t3;      // r[2 * rowgroup + colgroup] = block index
    }
    int v = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) v = (w == k) ? r[w] : v;
    mydc = (v + 32) >> 6;
  }
  m[0] = mydc;
  // ---- AC quantisation of the lane's block (coefficients 1..15 of the zig-zag)
  int cost = 0, nz = 0, ncoef = 0;
  int16_t alev[16]; uint8_t arun[16]; int16_t fadj[16];
#pragma unroll
  for (int c = 0; c < 16; c++) { alev[c] = 0; arun[c] = 0; fadj[c] = 0; }
  {
    const int q_bits = 15 + prm.qp_per_ac;
    constexpr int ZZ[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
    constexpr int CC[16] = {3, 2, 2, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int run = 0;
#pragma unroll
    for (int c = 1; c < 16; c++) {
      const int idx = ZZ[c], v = m[idx];
      if (v != 0) {
        const int scaled = iabsc_(v) * prm.q_ac[idx].ScaleComp;
        int l = (scaled + prm.q_ac[idx].OffsetComp) >> q_bits;
        if (l != 0) {
          if (prm.cavlc) l = min(l, 2063);
          if (prm.adaptive_rounding) fadj[idx] = (int16_t)((prm.adapt_rnd_weight * (scaled - (l << q_bits)) + (1 << q_bits)) >> (q_bits + 1));
          int cc = 0;
#pragma unroll
          for (int w = 0; w < 16; w++) cc = (w == run) ? CC[w] : cc;
          cost += (l > 1) ? 999999 : cc;
          l = v < 0 ? -l : l;
          m[idx] = (((l * prm.q_ac[idx].InvScaleComp) << prm.qp_per_ac) + 8) >> 4;
#pragma unroll
          for (int w = 0; w < 16; w++) if (w == ncoef) { alev[w] = (int16_t)l; arun[w] = (uint8_t)run; }
          ncoef++; run = 0; nz = 1;
        } else { m[idx] = 0; run++; }
      } else run++;
    }
  }
  if (!blk) { nz = 0; cost = 0; }
  // ---- group-wide: summed cost, which blocks are coded; thresholding (block.c:1139-1171)
  int tot = cost;
  tot += __shfl_xor(tot, 1, 64); tot += __shfl_xor(tot, 2, 64); tot += __shfl_xor(tot, 4, 64);
  unsigned nzmask = (unsigned)((__ballot(nz != 0) >> gb) & 0xffull);
  const bool drop = nzmask != 0 && tot < 4;
  if (drop && nz) {
#pragma unroll
    for (int c = 1; c < 16; c++) m[c] = 0;                // raster positions 1..15 are exactly the AC coefficients
#pragma unroll
    for (int c = 0; c < 16; c++) alev[c] = 0;
    nz = 0;
  }
  // ---- inverse transform of every block with a DC or surviving AC; reconstruction
  const bool doinv = blk && (m[0] != 0 || nz);
  const bool any = ((__ballot(doinv) >> gb) & 0xffull) != 0;
  if (doinv) {
#pragma unroll
    for (int i = 0; i < 4; i++) inv4c(m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]);
#pragma unroll
    for (int i = 0; i < 4; i++) inv4c(m[i], m[4 + i], m[8 + i], m[12 + i]);
  }
  if (live && blk) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t w = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        int v = pr[4 * j + i];
        if (any) { v += (m[4 * j + i] + 32) >> 6; v = v < 0 ? 0 : (v > prm.max_pel ? prm.max_pel : v); }
        w |= (uint32_t)v << (8 * i);
      }
      *(uint32_t *)(o->rec + (n2 + j) * 8 + n1) = w;
#pragma unroll
      for (int i = 0; i < 4; i++) o->fadjust[(n2 + j) * 8 + n1 + i] = fadj[4 * j + i];
    }
#pragma unroll
    for (int c = 0; c < 16; c++) { o->ac_level[k][c] = alev[c]; o->ac_run[k][c] = arun[c]; }
    o->ac_ncoef[k] = (uint8_t)(drop ? 0 : ncoef);
  }
  if (live && !blk) {                                     // 4:2:0: the unused half of the record is zero
#pragma unroll
    for (int c = 0; c < 16; c++) { o->ac_level[k][c] = 0; o->ac_run[k][c] = 0; }
    o->ac_ncoef[k] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { *(uint32_t *)(o->rec + (n2 + j) * 8 + n1) = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) o->fadjust[(n2 + j) * 8 + n1 + i] = 0; }
  }
  if (live && k == 0) {
#pragma unroll
    for (int w = 0; w < 9; w++) { o->dc_level[w] = (int16_t)(w < ndc ? dcl[w] : 0); o->dc_run[w] = (uint8_t)(w < ndc ? dcr[w] : 0); }
    o->dc_nonzero = (uint8_t)dczero;
    o->reserved_[0] = o->reserved_[1] = o->reserved_[2] = o->reserved_[3] = 0;
    // cbp_blk / cr_cbp exactly as JM updates them
    long long cbp = mb.cbp_blk;
    int cr_cbp = mb.cr_cbp;
    const int uv_scale = uv * (yuv == 2 ? 2 : 1);
    if (dczero) {                                          // 32-bit int mask, sign-extended (block.c:1044 / :1080)
      cbp |= (long long)(int)(yuv == 1 ? 0xf0000u << (uv << 2) : 0xff0000u << (uv << 3));
      cr_cbp = max(1, cr_cbp);
    }
    cbp |= (long long)nzmask << (16 + 4 * uv_scale);
    if (drop && !dczero) cbp &= ~((long long)(yuv == 1 ? 0xf0000 : 0xff0000) << (uv << (1 + yuv)));
    if (nzmask != 0 && !drop) cr_cbp = 2;
    jmhip_tqc_mb r; r.cbp_blk = cbp; r.cr_cbp = cr_cbp; r.uv = uv;
    mbs[it] = r;
  }
}

extern "C" int jmhip_tq_chroma_dev(jmhip_ctx *ctx, const jmhip_tqc_params *prm, jmhip_tqc_mb *d_mbs, const uint8_t *d_orig, const uint8_t *d_pred,
                                   int32_t n, jmhip_tqc_out *d_out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!prm || n < 0 || (n > 0 && (!d_mbs || !d_orig || !d_pred || !d_out))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_chroma_dev: bad argument");
  if (prm->yuv_format != 1 && prm->yuv_format != 2) return jmhip_fail(ctx, JMHIP_EUNSUPPORTED, "jmhip_tq_chroma_dev: yuv_format %d (4:2:0 and 4:2:2 only)", prm->yuv_format);
  if (prm->qp_per_ac < 0 || prm->qp_per_ac > 8 || prm->qp_per_dc < 0 || prm->qp_per_dc > 9) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_chroma_dev: qp_per outside range");
  if (n == 0) return JMHIP_OK;
  hipLaunchKernelGGL(k_tq_chroma, dim3((n + 31) / 32), dim3(256), 0, ctx->stream, *prm, d_mbs, d_orig, d_pred, n, d_out);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_tq_chroma(jmhip_ctx *ctx, const jmhip_tqc_params *prm, jmhip_tqc_mb *mbs, const uint8_t *orig, const uint8_t *pred,
                               int32_t n, jmhip_tqc_out *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!prm || !mbs || !orig || !pred || !out || n < 0) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_chroma: bad argument");
  if (prm->yuv_format != 1 && prm->yuv_format != 2) return jmhip_fail(ctx, JMHIP_EUNSUPPORTED, "jmhip_tq_chroma: yuv_format %d (4:2:0 and 4:2:2 only)", prm->yuv_format);
  if (prm->qp_per_ac < 0 || prm->qp_per_ac > 8 || prm->qp_per_dc < 0 || prm->qp_per_dc > 9) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_chroma: qp_per outside range");
  if (n == 0) return JMHIP_OK;
  for (int i = 0; i < n; i++) if (mbs[i].uv != 0 && mbs[i].uv != 1) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_chroma: item %d: uv %d", i, mbs[i].uv);
  int r; void *din, *dout;
  const size_t in_bytes = (size_t)n * (256 + sizeof(jmhip_tqc_mb));
  if ((r = jmhip_scratch(ctx, 0, in_bytes, &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * sizeof(jmhip_tqc_out), &dout))) return r;
  uint8_t *d_orig = (uint8_t *)din, *d_pred = d_orig + (size_t)n * 128;
  jmhip_tqc_mb *d_mbs = (jmhip_tqc_mb *)(d_pred + (size_t)n * 128);
  HIPCHK(ctx, hipMemcpyAsync(d_orig, orig, (size_t)n * 128, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_pred, pred, (size_t)n * 128, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_mbs, mbs, (size_t)n * sizeof(jmhip_tqc_mb), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_tq_chroma, dim3((n + 31) / 32), dim3(256), 0, ctx->stream, *prm, d_mbs, d_orig, d_pred, n, (jmhip_tqc_out *)dout);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(mbs, d_mbs, (size_t)n * sizeof(jmhip_tqc_mb), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * sizeof(jmhip_tqc_out), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}
