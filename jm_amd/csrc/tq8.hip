// tq8.hip -- K7+K8 for the 8x8 transform, and the DC transforms / DC quantiser (gfx950).
//
// Device counterpart of (reference):
//   residual_transform_quant_luma_8x8 / _cavlc   lencod/src/transform8x8.c:522-586 / :604-672
//   forward8x8 / inverse8x8                      lcommon/src/transform.c:353-448 / :450-547
//   quant_8x8_normal / quant_8x8cavlc_normal     lencod/src/quant8x8_normal.c:43-107 / :123-203
//   quant_8x8_around / quant_8x8cavlc_around     lencod/src/quant8x8_around.c:43-123 / :136-220
//   SNGL_SCAN8x8, SNGL_SCAN8x8_CAVLC, COEFF_COST8x8   lencod/src/transform8x8.c
//   hadamard4x4 / ihadamard4x4 / hadamard4x2 / ihadamard4x2 / hadamard2x2 / ihadamard2x2   transform.c:121-330
//   quant_dc4x4_normal                           lencod/src/quant4x4_normal.c:200-259
//
// One lane = one block, as in tq.hip: the 64 residuals of an 8x8 block live in registers through both butterfly passes, the
// quantiser walks the zig-zag fully unrolled (every coefficient / parameter index is a literal), the level/run lists are
// appended in a per-lane LDS record, and the 408-byte records leave through a coalesced dword copy.
// Algorithmic bytes per 8x8 block: 128 in + 408 out.
#include "jmhip_internal.h"

__device__ __forceinline__ int iabs8_(int v) { return v < 0 ? -v : v; }

__device__ __forceinline__ void fwd8r(int &p0, int &p1, int &p2, int &p3, int &p4, int &p5, int &p6, int &p7)
{
  int a0 = p0 + p7, a1 = p1 + p6, a2 = p2 + p5, a3 = p3 + p4;
  const int b0 = a0 + a3, b1 = a1 + a2, b2 = a0 - a3, b3 = a1 - a2;
  a0 = p0 - p7; a1 = p1 - p6; a2 = p2 - p5; a3 = p3 - p4;
  const int b4 = a1 + a2 + ((a0 >> 1) + a0), b5 = a0 - a3 - ((a2 >> 1) + a2);
  const int b6 = a0 + a3 - ((a1 >> 1) + a1), b7 = a1 - a2 + ((a3 >> 1) + a3);
  p0 = b0 + b1; p1 = b4 + (b7 >> 2); p2 = b2 + (b3 >> 1); p3 = b5 + (b6 >> 2);
  p4 = b0 - b1; p5 = b6 - (b5 >> 2); p6 = (b2 >> 1) - b3; p7 = (b4 >> 2) - b7;
}
__device__ __forceinline__ void inv8r(int &p0, int &p1, int &p2, int &p3, int &p4, int &p5, int &p6, int &p7)
{
  int a0 = p0 + p4, a1 = p0 - p4, a2 = p6 - (p2 >> 1), a3 = p2 + (p6 >> 1);
  const int b0 = a0 + a3, b2 = a1 - a2, b4 = a1 + a2, b6 = a0 - a3;
  a0 = -p3 + p5 - p7 - (p7 >> 1); a1 = p1 + p7 - p3 - (p3 >> 1);
  a2 = -p1 + p7 + p5 + (p5 >> 1); a3 = p3 + p5 + p1 + (p1 >> 1);
  const int b1 = a0 + (a3 >> 2), b3 = a1 + (a2 >> 2), b5 = a2 - (a1 >> 2), b7 = a3 - (a0 >> 2);
  p0 = b0 + b7; p1 = b2 - b5; p2 = b4 + b3; p3 = b6 + b1;
  p4 = b6 - b1; p5 = b4 - b3; p6 = b2 + b5; p7 = b0 - b7;
}

#define TQ8_THREADS 64
__global__ __launch_bounds__(TQ8_THREADS) void k_tq_luma8x8(jmhip_tq8_params prm, const uint8_t *__restrict__ orig, const uint8_t *__restrict__ pred,
                                                            int n, jmhip_tq8_out *__restrict__ out)
{
  __shared__ __attribute__((aligned(16))) jmhip_tq8_out s_out[TQ8_THREADS];
  const int b = blockIdx.x * TQ8_THREADS + threadIdx.x;
  jmhip_tq8_out &o = s_out[threadIdx.x];
  if (b < n) {
    int m[64], any = 0;
    uint32_t pw[16];                                       // prediction, packed
    {
      const uint4 *po = (const uint4 *)(orig + (long)b * 64), *pp = (const uint4 *)(pred + (long)b * 64);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint4 vo = po[k], vp = pp[k];
        const uint32_t wo[4] = {vo.x, vo.y, vo.z, vo.w}, wp[4] = {vp.x, vp.y, vp.z, vp.w};
#pragma unroll
        for (int t = 0; t < 16; t++) {
          m[16 * k + t] = (int)((wo[t >> 2] >> (8 * (t & 3))) & 255) - (int)((wp[t >> 2] >> (8 * (t & 3))) & 255);
          any |= m[16 * k + t];
        }
#pragma unroll
        for (int t = 0; t < 4; t++) pw[4 * k + t] = wp[t];
      }
    }
#pragma unroll
    for (int k = 0; k < 68; k++) { o.level[k] = 0; o.run[k] = 0; }
#pragma unroll
    for (int k = 0; k < 64; k++) o.fadjust[k] = 0;
    int nonzero = 0, cost = 0;
    int ncoef[4] = {0, 0, 0, 0};
    if (prm.cavlc || any) {                                // check_zero only guards the non-CAVLC function
#pragma unroll
      for (int i = 0; i < 8; i++) fwd8r(m[8 * i], m[8 * i + 1], m[8 * i + 2], m[8 * i + 3], m[8 * i + 4], m[8 * i + 5], m[8 * i + 6], m[8 * i + 7]);
#pragma unroll
      for (int i = 0; i < 8; i++) fwd8r(m[i], m[8 + i], m[16 + i], m[24 + i], m[32 + i], m[40 + i], m[48 + i], m[56 + i]);
      const int q_bits = 16 + prm.qp_per;
      // zig-zag position k -> raster index (SNGL_SCAN8x8); the CAVLC functions are handed the de-interleaved table
      // SNGL_SCAN8x8_CAVLC: list l = k & 3 takes zig-zag positions l, l+4, l+8, ... in that order, so walking the plain
      // zig-zag and appending position k to list k & 3 produces the same four lists
      constexpr int ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                              35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
      int run[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 64; k++) {
        const int idx = ZZ[k], l = prm.cavlc ? (k & 3) : 0;
        const int c = m[idx];
        int fadj = 0;
        if (c != 0) {
          const int scaled = iabs8_(c) * prm.q[idx].ScaleComp;
          int lev = (scaled + prm.q[idx].OffsetComp) >> q_bits;
          if (lev != 0) {
            if (prm.cavlc) lev = min(lev, 2063);
            if (prm.adaptive_rounding) fadj = (prm.adapt_rnd_weight * (scaled - (lev << q_bits)) + (1 << q_bits)) >> (q_bits + 1);
            const int r = run[l];
            cost += (lev > 1) ? 999999 : (r < 4 ? 3 : (r < 12 ? 2 : (r < 24 ? 1 : 0)));     // COEFF_COST8x8[0]
            lev = c < 0 ? -lev : lev;
            m[idx] = (((lev * prm.q[idx].InvScaleComp) << prm.qp_per) + 32) >> 6;
            const int slot = (prm.cavlc ? 17 * l : 0) + ncoef[l];
            o.level[slot] = (int16_t)lev; o.run[slot] = (uint8_t)r; ncoef[l]++;
            run[l] = 0; nonzero = 1;
          } else { m[idx] = 0; run[l]++; }
        } else run[l]++;
        o.fadjust[idx] = (int16_t)fadj;
      }
    }
    if (nonzero) {
#pragma unroll
      for (int i = 0; i < 8; i++) inv8r(m[8 * i], m[8 * i + 1], m[8 * i + 2], m[8 * i + 3], m[8 * i + 4], m[8 * i + 5], m[8 * i + 6], m[8 * i + 7]);
#pragma unroll
      for (int i = 0; i < 8; i++) inv8r(m[i], m[8 + i], m[16 + i], m[24 + i], m[32 + i], m[40 + i], m[48 + i], m[56 + i]);
#pragma unroll
      for (int k = 0; k < 64; k++) {
        const int v = ((m[k] + 32) >> 6) + (int)((pw[k >> 2] >> (8 * (k & 3))) & 255);
        o.rec[k] = (uint8_t)(v < 0 ? 0 : (v > prm.max_pel ? prm.max_pel : v));
      }
    } else {
#pragma unroll
      for (int k = 0; k < 64; k++) o.rec[k] = (uint8_t)((pw[k >> 2] >> (8 * (k & 3))) & 255);
    }
    // CAVLC: a list walks the zig-zag positions l, l+4, ... -- but JM's run counts zeros WITHIN the list (the scan quarter), which
    // is what run[l] above counts: position k only advances run[k & 3]
    o.coeff_cost = cost; o.nonzero = (uint8_t)nonzero; o.any_residual = any ? 1 : 0;
#pragma unroll
    for (int l = 0; l < 4; l++) o.ncoef[l] = (uint8_t)ncoef[l];
    o.reserved_[0] = o.reserved_[1] = 0;
  }
  __syncthreads();
  const int first = blockIdx.x * TQ8_THREADS, cnt = min(TQ8_THREADS, n - first);
  const uint32_t *src = (const uint32_t *)s_out;
  uint32_t *dst = (uint32_t *)(out + first);
  const int ndw = cnt * (int)(sizeof(jmhip_tq8_out) / 4);
  for (int k = threadIdx.x; k < ndw; k += TQ8_THREADS) dst[k] = src[k];
}

extern "C" int jmhip_tq_luma8x8_dev(jmhip_ctx *ctx, const jmhip_tq8_params *prm, const uint8_t *d_orig, const uint8_t *d_pred, int32_t n, jmhip_tq8_out *d_out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!prm || !d_orig || !d_pred || !d_out || n < 0) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_luma8x8_dev: bad argument");
  if (prm->qp_per < 0 || prm->qp_per > 8) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_luma8x8: qp_per %d outside 0..8", prm->qp_per);
  if (n == 0) return JMHIP_OK;
  jmhip_time_begin(ctx, 3);
  hipLaunchKernelGGL(k_tq_luma8x8, dim3((n + TQ8_THREADS - 1) / TQ8_THREADS), dim3(TQ8_THREADS), 0, ctx->stream, *prm, d_orig, d_pred, n, d_out);
  jmhip_time_end(ctx, 3);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_tq_luma8x8(jmhip_ctx *ctx, const jmhip_tq8_params *prm, const uint8_t *orig, const uint8_t *pred, int32_t n, jmhip_tq8_out *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!prm || !orig || !pred || !out || n < 0) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_luma8x8: bad argument");
  if (n == 0) return JMHIP_OK;
  int r; void *din, *dout;
  if ((r = jmhip_scratch(ctx, 0, (size_t)n * 128, &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * sizeof(jmhip_tq8_out), &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(din, orig, (size_t)n * 64, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync((uint8_t *)din + (size_t)n * 64, pred, (size_t)n * 64, hipMemcpyHostToDevice, ctx->stream));
  if ((r = jmhip_tq_luma8x8_dev(ctx, prm, (const uint8_t *)din, (const uint8_t *)din + (size_t)n * 64, n, (jmhip_tq8_out *)dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * sizeof(jmhip_tq8_out), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ DC transforms
__global__ __launch_bounds__(256) void k_dc_transform(int kind, const int32_t *__restrict__ in, int n, int32_t *__restrict__ out)
{
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  if (kind == JMHIP_DC_HADAMARD4x4 || kind == JMHIP_DC_IHADAMARD4x4) {
    int s[16], t[16];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int4 v = ((const int4 *)(in + (long)b * 16))[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
    if (kind == JMHIP_DC_HADAMARD4x4) {                    // transform.c:121-168: rows, then columns with >> 1
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int e0 = s[4 * i] + s[4 * i + 3], e1 = s[4 * i + 1] + s[4 * i + 2], o0 = s[4 * i + 1] - s[4 * i + 2], o1 = s[4 * i] - s[4 * i + 3];
        t[4 * i] = e0 + e1; t[4 * i + 1] = o1 + o0; t[4 * i + 2] = e0 - e1; t[4 * i + 3] = o1 - o0;
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int e0 = t[i] + t[12 + i], e1 = t[4 + i] + t[8 + i], o0 = t[4 + i] - t[8 + i], o1 = t[i] - t[12 + i];
        s[i] = (e0 + e1) >> 1; s[4 + i] = (o0 + o1) >> 1; s[8 + i] = (e0 - e1) >> 1; s[12 + i] = (o1 - o0) >> 1;
      }
    } else {                                               // transform.c:170-220
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int e0 = s[4 * i] + s[4 * i + 2], e1 = s[4 * i] - s[4 * i + 2], o0 = s[4 * i + 1] - s[4 * i + 3], o1 = s[4 * i + 1] + s[4 * i + 3];
        t[4 * i] = e0 + o1; t[4 * i + 1] = e1 + o0; t[4 * i + 2] = e1 - o0; t[4 * i + 3] = e0 - o1;
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int e0 = t[i] + t[8 + i], e1 = t[i] - t[8 + i], o0 = t[4 + i] - t[12 + i], o1 = t[4 + i] + t[12 + i];
        s[i] = e0 + o1; s[4 + i] = e1 + o0; s[8 + i] = e1 - o0; s[12 + i] = e0 - o1;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) ((int4 *)(out + (long)b * 16))[k] = make_int4(s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
  } else if (kind == JMHIP_DC_HADAMARD4x2 || kind == JMHIP_DC_IHADAMARD4x2) {
    int s[8], t[8], r[8];
#pragma unroll
    for (int k = 0; k < 2; k++) { const int4 v = ((const int4 *)(in + (long)b * 8))[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
#pragma unroll
    for (int i = 0; i < 4; i++) { t[i] = s[i] + s[4 + i]; t[4 + i] = s[i] - s[4 + i]; }
    if (kind == JMHIP_DC_HADAMARD4x2) {                    // transform.c:220-256
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int t0 = t[4 * i] + t[4 * i + 3], t1 = t[4 * i + 1] + t[4 * i + 2], t2 = t[4 * i + 1] - t[4 * i + 2], t3 = t[4 * i] - t[4 * i + 3];
        r[4 * i] = t0 + t1; r[4 * i + 1] = t3 + t2; r[4 * i + 2] = t0 - t1; r[4 * i + 3] = t3 - t2;
      }
    } else {                                               // transform.c:258-298, result transposed [4][2]
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int t0 = t[4 * i] + t[4 * i + 2], t1 = t[4 * i] - t[4 * i + 2], t2 = t[4 * i + 1] - t[4 * i + 3], t3 = t[4 * i + 1] + t[4 * i + 3];
        r[i] = t0 + t3; r[2 + i] = t1 + t2; r[4 + i] = t1 - t2; r[6 + i] = t0 - t3;
      }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) ((int4 *)(out + (long)b * 8))[k] = make_int4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
  } else {                                                 // 2x2, forward and inverse are the same butterfly (transform.c:301-330)
    const int4 v = ((const int4 *)in)[b];
    const int p0 = v.x + v.y, p1 = v.x - v.y, p2 = v.z + v.w, p3 = v.z - v.w;
    ((int4 *)out)[b] = make_int4(p0 + p2, p1 + p3, p0 - p2, p1 - p3);
  }
}

extern "C" int jmhip_dc_transform(jmhip_ctx *ctx, int32_t kind, const int32_t *in, int32_t n, int32_t *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!in || !out || n < 0 || kind < 0 || kind > JMHIP_DC_IHADAMARD2x2) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_dc_transform: bad argument");
  if (n == 0) return JMHIP_OK;
  const size_t per = kind <= JMHIP_DC_IHADAMARD4x4 ? 16 : (kind <= JMHIP_DC_IHADAMARD4x2 ? 8 : 4), bytes = (size_t)n * per * 4;
  int r; void *din, *dout;
  if ((r = jmhip_scratch(ctx, 0, bytes, &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, bytes, &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_dc_transform, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, kind, (const int32_t *)din, n, (int32_t *)dout);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ DC quantiser
__global__ __launch_bounds__(256) void k_quant_dc4x4(jmhip_qparam q, int qp_per, int cavlc, int32_t *__restrict__ blocks, int n, jmhip_dc_out *__restrict__ out)
{
  __shared__ jmhip_dc_out s_out[256];
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  jmhip_dc_out &o = s_out[threadIdx.x];
  int m[16];
#pragma unroll
  for (int k = 0; k < 4; k++) { const int4 v = ((const int4 *)(blocks + (long)b * 16))[k]; m[4 * k] = v.x; m[4 * k + 1] = v.y; m[4 * k + 2] = v.z; m[4 * k + 3] = v.w; }
#pragma unroll
  for (int k = 0; k < 17; k++) { o.level[k] = 0; o.run[k] = 0; }
  const int q_bits = 15 + qp_per + 1;
  constexpr int ZZ[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
  int run = 0, ncoef = 0, nonzero = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int idx = ZZ[k], c = m[idx];
    if (c != 0) {
      int lev = (iabs8_(c) * q.ScaleComp + (q.OffsetComp << 1)) >> q_bits;
      if (lev != 0) {
        if (cavlc) lev = min(lev, 2063);
        lev = c < 0 ? -lev : lev;
        m[idx] = lev; o.level[ncoef] = (int16_t)lev; o.run[ncoef] = (uint8_t)run; ncoef++; run = 0; nonzero = 1;
      } else { m[idx] = 0; run++; }
    } else run++;
  }
  o.nonzero = (uint8_t)nonzero;
#pragma unroll
  for (int k = 0; k < 4; k++) ((int4 *)(blocks + (long)b * 16))[k] = make_int4(m[4 * k], m[4 * k + 1], m[4 * k + 2], m[4 * k + 3]);
  out[b] = o;
}

extern "C" int jmhip_quant_dc4x4(jmhip_ctx *ctx, const jmhip_qparam *q, int32_t qp_per, int32_t cavlc, int32_t *blocks, int32_t n, jmhip_dc_out *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!q || !blocks || !out || n < 0 || qp_per < 0 || qp_per > 8) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_quant_dc4x4: bad argument");
  if (n == 0) return JMHIP_OK;
  int r; void *dblk, *dout;
  if ((r = jmhip_scratch(ctx, 0, (size_t)n * 64, &dblk))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * sizeof(jmhip_dc_out), &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(dblk, blocks, (size_t)n * 64, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_quant_dc4x4, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, *q, qp_per, cavlc, (int32_t *)dblk, n, (jmhip_dc_out *)dout);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(blocks, dblk, (size_t)n * 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * sizeof(jmhip_dc_out), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}
