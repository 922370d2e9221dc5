// tq.hip -- K7+K8: batched 4x4 luma residual transform, quantisation, dequantisation, inverse
// transform and reconstruction; plus the bare 4x4 / 8x8 transforms (gfx950).
//
// Device counterpart of (reference):
//   residual_transform_quant_luma_4x4   lencod/src/block.c:661-725   ("dct_4x4" of the north star)
//   forward4x4 / inverse4x4             lcommon/src/transform.c:20-68 / :70-118
//   forward8x8 / inverse8x8             lcommon/src/transform.c:353-448 / :450-547
//   quant_4x4_normal                    lencod/src/quant4x4_normal.c:39-115
//   quant_4x4_around                    lencod/src/quant4x4_around.c:40-127
//   sample_reconstruct                  lcommon/src/blk_prediction.c:48-62   (DQ_BITS 6)
//   frame zig-zag SNGL_SCAN block.c:170-176, COEFF_COST4x4[0] block.c:72-77, MAX_VALUE 999999 defines.h:124,
//   CAVLC_LEVEL_LIMIT 2063 defines.h:100, Q_BITS 15 defines.h:312
//
// One lane = one 4x4 block: the 16 residuals live in registers and both butterfly passes are
// register-to-register, so no cross-lane traffic is needed at all.  Input is two coalesced 16-byte
// loads per lane; the 104-byte result records are transposed through LDS so that the stores to HBM
// are 16-byte-per-lane coalesced as well.  Algorithmic bytes per block: 32 in + 104 out.
#include "jmhip_internal.h"

__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }

__device__ __forceinline__ void fwd4(int &a, int &b, int &c, int &d)
{
  int e0 = a + d, e1 = b + c, o0 = b - c, o1 = a - d;
  a = e0 + e1; b = (o1 << 1) + o0; c = e0 - e1; d = o1 - (o0 << 1);
}
__device__ __forceinline__ void inv4(int &a, int &b, int &c, int &d)
{
  int e0 = a + c, e1 = a - c, o0 = (b >> 1) - d, o1 = b + (d >> 1);
  a = e0 + o1; b = e1 + o0; c = e1 - o0; d = e0 - o1;
}
__device__ __forceinline__ void forward4x4_regs(int m[16])
{
#pragma unroll
  for (int i = 0; i < 4; i++) fwd4(m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]);
#pragma unroll
  for (int i = 0; i < 4; i++) fwd4(m[i], m[4 + i], m[8 + i], m[12 + i]);
}
__device__ __forceinline__ void inverse4x4_regs(int m[16])
{
#pragma unroll
  for (int i = 0; i < 4; i++) inv4(m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]);
#pragma unroll
  for (int i = 0; i < 4; i++) inv4(m[i], m[4 + i], m[8 + i], m[12 + i]);
}

__device__ __forceinline__ void fwd8(int p[8])
{
  int a0 = p[0] + p[7], a1 = p[1] + p[6], a2 = p[2] + p[5], a3 = p[3] + p[4];
  int b0 = a0 + a3, b1 = a1 + a2, b2 = a0 - a3, b3 = a1 - a2;
  a0 = p[0] - p[7]; a1 = p[1] - p[6]; a2 = p[2] - p[5]; a3 = p[3] - p[4];
  int b4 = a1 + a2 + ((a0 >> 1) + a0), b5 = a0 - a3 - ((a2 >> 1) + a2);
  int b6 = a0 + a3 - ((a1 >> 1) + a1), b7 = a1 - a2 + ((a3 >> 1) + a3);
  p[0] = b0 + b1; p[1] = b4 + (b7 >> 2); p[2] = b2 + (b3 >> 1); p[3] = b5 + (b6 >> 2);
  p[4] = b0 - b1; p[5] = b6 - (b5 >> 2); p[6] = (b2 >> 1) - b3; p[7] = (b4 >> 2) - b7;
}
__device__ __forceinline__ void inv8(int p[8])
{
  int a0 = p[0] + p[4], a1 = p[0] - p[4], a2 = p[6] - (p[2] >> 1), a3 = p[2] + (p[6] >> 1);
  int b0 = a0 + a3, b2 = a1 - a2, b4 = a1 + a2, b6 = a0 - a3;
  a0 = -p[3] + p[5] - p[7] - (p[7] >> 1); a1 = p[1] + p[7] - p[3] - (p[3] >> 1);
  a2 = -p[1] + p[7] + p[5] + (p[5] >> 1); a3 = p[3] + p[5] + p[1] + (p[1] >> 1);
  int b1 = a0 + (a3 >> 2), b3 = a1 + (a2 >> 2), b5 = a2 - (a1 >> 2), b7 = a3 - (a0 >> 2);
  p[0] = b0 + b7; p[1] = b2 - b5; p[2] = b4 + b3; p[3] = b6 + b1;
  p[4] = b6 - b1; p[5] = b4 - b3; p[6] = b2 + b5; p[7] = b0 - b7;
}

template <int INV>
__global__ __launch_bounds__(256) void k_xform4(const int32_t *__restrict__ in, int n, int32_t *__restrict__ out)
{
  int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  int m[16];
  const int4 *pi = (const int4 *)(in + (long)b * 16);
#pragma unroll
  for (int k = 0; k < 4; k++) { int4 v = pi[k]; m[4 * k] = v.x; m[4 * k + 1] = v.y; m[4 * k + 2] = v.z; m[4 * k + 3] = v.w; }
  if (INV) inverse4x4_regs(m); else forward4x4_regs(m);
  int4 *po = (int4 *)(out + (long)b * 16);
#pragma unroll
  for (int k = 0; k < 4; k++) po[k] = make_int4(m[4 * k], m[4 * k + 1], m[4 * k + 2], m[4 * k + 3]);
}

template <int INV>
__global__ __launch_bounds__(64) void k_xform8(const int32_t *__restrict__ in, int n, int32_t *__restrict__ out)
{
  // 8 lanes per block: lane r holds row r, the column pass goes through LDS
  __shared__ int s[8][8][9];
  const int g = threadIdx.x >> 3, r = threadIdx.x & 7, b = blockIdx.x * 8 + g;
  int p[8];
  if (b < n) {
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = in[(long)b * 64 + r * 8 + i];
    if (INV) inv8(p); else fwd8(p);
#pragma unroll
    for (int i = 0; i < 8; i++) s[g][r][i] = p[i];
  }
  __syncthreads();
  if (b < n) {
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = s[g][i][r];       // column r
    if (INV) inv8(p); else fwd8(p);
#pragma unroll
    for (int i = 0; i < 8; i++) out[(long)b * 64 + i * 8 + r] = p[i];
  }
}


// residual_transform_quant_luma_4x4 of one block held in registers (block.c:661-725): orig / pred as four row dwords, the record into LDS
__device__ __forceinline__ void tq_luma4x4_block(const jmhip_tq_params &prm, const uint32_t (&wo)[4], const uint32_t (&wp)[4], jmhip_tq_out &o)
{
  int m[16], pr[16], any = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    pr[k] = (wp[k >> 2] >> (8 * (k & 3))) & 255;
    m[k] = (int)((wo[k >> 2] >> (8 * (k & 3))) & 255) - pr[k];
    any |= m[k];
  }
  int nonzero = 0, ncoef = 0, cost = 0;
  int16_t *s_lev = o.level; uint8_t *s_run = o.run;       // level/run lists are appended in LDS (dynamic index)
#pragma unroll
  for (int k = 0; k < 16; k++) { o.level[k] = 0; o.run[k] = 0; o.fadjust[k] = 0; }
  if (any) {                                                                 // check_zero, block.c:627-640
    forward4x4_regs(m);
    const int q_bits = 15 + prm.qp_per;
    int run = 0;
    // zig-zag scan position k -> raster index j*4+i (SNGL_SCAN, block.c:170-176); the loop is fully
    // unrolled so every m[] / q[] index is a literal and the block stays in registers
    constexpr int ZZ[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
    constexpr int CC[16] = {3, 2, 2, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // COEFF_COST4x4[0], block.c:72-77
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int idx = ZZ[k];
      const int c = m[idx];
      int fadj = 0;
      if (c != 0) {
        const int scaled = iabs_(c) * prm.q[idx].ScaleComp;
        int lev = (scaled + prm.q[idx].OffsetComp) >> q_bits;
        if (lev != 0) {
          if (prm.cavlc) lev = min(lev, 2063);
          if (prm.adaptive_rounding) fadj = (prm.adapt_rnd_weight * (scaled - (lev << q_bits)) + (1 << q_bits)) >> (q_bits + 1);
          int cc = 0;
#pragma unroll
          for (int t = 0; t < 16; t++) cc = (t == run) ? CC[t] : cc;
          cost += (lev > 1) ? 999999 : cc;
          lev = c < 0 ? -lev : lev;
          m[idx] = (((lev * prm.q[idx].InvScaleComp) << prm.qp_per) + 8) >> 4;
          s_lev[ncoef] = (int16_t)lev; s_run[ncoef] = (uint8_t)run; ncoef++;
          run = 0; nonzero = 1;
        } else { m[idx] = 0; run++; }
      } else run++;
      o.fadjust[idx] = (int16_t)fadj;
    }
  }
  if (nonzero) {
    inverse4x4_regs(m);
#pragma unroll
    for (int k = 0; k < 16; k++) { int v = ((m[k] + 32) >> 6) + pr[k]; o.rec[k] = (uint8_t)(v < 0 ? 0 : (v > prm.max_pel ? prm.max_pel : v)); }
  } else {
#pragma unroll
    for (int k = 0; k < 16; k++) o.rec[k] = (uint8_t)pr[k];
  }
  o.coeff_cost = cost; o.nonzero = (uint8_t)nonzero; o.any_residual = any ? 1 : 0; o.ncoef = (uint8_t)ncoef; o.reserved_ = 0;
}

__global__ __launch_bounds__(256) void k_tq_luma4x4(jmhip_tq_params prm, const uint8_t *__restrict__ orig, const uint8_t *__restrict__ pred,
                                                    int n, jmhip_tq_out *__restrict__ out)
{
  __shared__ __attribute__((aligned(16))) jmhip_tq_out s_out[256];
  const int b = blockIdx.x * 256 + threadIdx.x;
  jmhip_tq_out &o = s_out[threadIdx.x];
  if (b < n) {
    const uint4 vo = *(const uint4 *)(orig + (long)b * 16), vp = *(const uint4 *)(pred + (long)b * 16);
    const uint32_t wo[4] = {vo.x, vo.y, vo.z, vo.w}, wp[4] = {vp.x, vp.y, vp.z, vp.w};
    tq_luma4x4_block(prm, wo, wp, o);
  }
  __syncthreads();
  // coalesced copy-out of this workgroup's records (26 dwords each)
  const int first = blockIdx.x * 256, cnt = min(256, n - first);
  const uint32_t *src = (const uint32_t *)s_out;
  uint32_t *dst = (uint32_t *)(out + first);
  const int ndw = cnt * (int)(sizeof(jmhip_tq_out) / 4);
  for (int k = threadIdx.x; k < ndw; k += 256) dst[k] = src[k];
}

// ---- P pictures coded as 16x16 macroblocks, one launch per picture: luma_prediction of each window job's 16x16 partition with the vector
//      the refinement left (as k_mc_mb16), residual_transform_quant_luma_4x4 of its sixteen blocks (as k_tq_luma4x4) and the reconstructed
//      samples written into the picture (as k_tq_rec_to_plane) -- the prediction never leaves the registers.  16 lanes per macroblock
//      (lane = 4x4 block), 16 macroblocks per workgroup; records leave through LDS, the four blocks of a block row as one 416-byte run.
__device__ __forceinline__ uint32_t ld4u_tq(const uint8_t *p) { struct __attribute__((packed)) U { uint32_t v; }; return ((const U *)p)->v; }
__global__ __launch_bounds__(256) void k_mb16_recon_luma(jmhip_tq_params prm, const jmhip_me_job *__restrict__ jobs, const jmhip_me_result *__restrict__ results, int n,
                                                         const uint8_t *__restrict__ planes, int pitch, long plane_stride, int W, int H,
                                                         int y_offset, int blocks_per_row, const uint8_t *__restrict__ orig,
                                                         jmhip_tq_out *__restrict__ out, uint8_t *__restrict__ pred, uint8_t *__restrict__ plane, int plane_pitch)
{
  __shared__ __attribute__((aligned(16))) jmhip_tq_out s_out[256];
  __shared__ int s_blk[256];
  const int tid = threadIdx.x, j = blockIdx.x * 16 + (tid >> 4), l = tid & 15, bx = l & 3, by = l >> 2;
  jmhip_tq_out &o = s_out[tid];
  s_blk[tid] = -1;
  if (j < n) {
    const int mb_x = jobs[j].mb_x, mb_y = jobs[j].mb_y;
    const jmhip_me_best mv = results[j].best[0];
    const int qx = (mb_x << 2) + mv.mv_x, qy = (mb_y << 2) + mv.mv_y;
    const int yy = min(max(qy >> 2, -JMHIP_PAD_Y), H + 3), xx = min(max(qx >> 2, -JMHIP_PAD_X), W + 15);     // one clamped origin per block (UMVLine4X)
    const uint8_t *src = planes + ((qy & 3) * 4 + (qx & 3)) * plane_stride + (long)(yy + JMHIP_PAD_Y + 4 * by) * pitch + xx + JMHIP_PAD_X + 4 * bx;
    const long blk = (long)(((mb_y - y_offset) >> 2) + by) * blocks_per_row + (mb_x >> 2) + bx;
    const uint32_t wp[4] = {ld4u_tq(src), ld4u_tq(src + pitch), ld4u_tq(src + 2 * pitch), ld4u_tq(src + 3 * pitch)};
    const uint4 vo = *(const uint4 *)(orig + blk * 16);
    const uint32_t wo[4] = {vo.x, vo.y, vo.z, vo.w};
    if (pred) *(uint4 *)(pred + blk * 16) = make_uint4(wp[0], wp[1], wp[2], wp[3]);
    tq_luma4x4_block(prm, wo, wp, o);
    s_blk[tid] = (int)blk;
    uint8_t *d = plane + (long)(mb_y - y_offset + 4 * by) * plane_pitch + mb_x + 4 * bx;
#pragma unroll
    for (int r = 0; r < 4; r++) *(uint32_t *)(d + (long)r * plane_pitch) = *(const uint32_t *)(o.rec + 4 * r);
  }
  __syncthreads();
  constexpr int RD = (int)(sizeof(jmhip_tq_out) / 4);                          // 26 dwords per record
  const uint32_t *src = (const uint32_t *)s_out;
  uint32_t *dst = (uint32_t *)out;
  for (int k = tid; k < 256 * RD; k += 256) {
    const int r = k / RD, w = k - r * RD, blk = s_blk[r];
    if (blk >= 0) dst[(long)blk * RD + w] = src[k];
  }
}

extern "C" int jmhip_mb16_recon_luma_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_tq_params *prm, const jmhip_me_job *d_jobs, const jmhip_me_result *d_results,
                                         int32_t njobs, int32_t y_offset, int32_t blocks_per_row, const uint8_t *d_orig, jmhip_tq_out *d_out,
                                         uint8_t *d_pred, uint8_t *d_plane, int32_t pitch_bytes)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!prm || njobs < 0 || slot < 0 || slot >= ctx->cfg.num_ref_slots || blocks_per_row < 4 || pitch_bytes < 4 * blocks_per_row || (pitch_bytes & 3) ||
      (njobs > 0 && (!d_jobs || !d_results || !d_orig || !d_out || !d_plane)))
    return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mb16_recon_luma_dev: bad argument");
  if (prm->qp_per < 0 || prm->qp_per > 8) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_mb16_recon_luma_dev: qp_per %d outside 0..8", prm->qp_per);
  if (njobs == 0) return JMHIP_OK;
  hipLaunchKernelGGL(k_mb16_recon_luma, dim3((njobs + 15) / 16), dim3(256), 0, ctx->stream, *prm, d_jobs, d_results, njobs, (const uint8_t *)ctx->d_sub[slot], ctx->pitch,
                     (long)ctx->plane_stride, ctx->W, ctx->H, y_offset, blocks_per_row, d_orig, d_out, d_pred, d_plane, pitch_bytes);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_tq_luma4x4_dev(jmhip_ctx *ctx, const jmhip_tq_params *prm, const uint8_t *d_orig, const uint8_t *d_pred, int32_t n, jmhip_tq_out *d_out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!prm || !d_orig || !d_pred || !d_out || n < 0) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_luma4x4_dev: bad argument");
  if (prm->qp_per < 0 || prm->qp_per > 8) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_luma4x4: qp_per %d outside 0..8", prm->qp_per);
  if (n == 0) return JMHIP_OK;
  jmhip_time_begin(ctx, 3);
  hipLaunchKernelGGL(k_tq_luma4x4, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, *prm, d_orig, d_pred, n, d_out);
  jmhip_time_end(ctx, 3);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_tq_luma4x4(jmhip_ctx *ctx, const jmhip_tq_params *prm, const uint8_t *orig, const uint8_t *pred, int32_t n, jmhip_tq_out *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!prm || !orig || !pred || !out || n < 0) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_tq_luma4x4: bad argument");
  if (n == 0) return JMHIP_OK;
  int r; void *din, *dout;
  if ((r = jmhip_scratch(ctx, 0, (size_t)n * 32, &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * sizeof(jmhip_tq_out), &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(din, orig, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync((uint8_t *)din + (size_t)n * 16, pred, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
  if ((r = jmhip_tq_luma4x4_dev(ctx, prm, (const uint8_t *)din, (const uint8_t *)din + (size_t)n * 16, n, (jmhip_tq_out *)dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * sizeof(jmhip_tq_out), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

template <int INV, int SZ>
static int xform_host(jmhip_ctx *ctx, const int32_t *in, int32_t n, int32_t *out, const char *name)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!in || !out || n < 0) return jmhip_fail(ctx, JMHIP_EINVAL, "%s: bad argument", name);
  if (n == 0) return JMHIP_OK;
  int r; void *din, *dout;
  const size_t bytes = (size_t)n * SZ * SZ * 4;
  if ((r = jmhip_scratch(ctx, 0, bytes, &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, bytes, &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, ctx->stream));
  if (SZ == 4) hipLaunchKernelGGL((k_xform4<INV>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const int32_t *)din, n, (int32_t *)dout);
  else         hipLaunchKernelGGL((k_xform8<INV>), dim3((n + 7) / 8), dim3(64), 0, ctx->stream, (const int32_t *)din, n, (int32_t *)dout);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}
extern "C" int jmhip_forward4x4(jmhip_ctx *ctx, const int32_t *in, int32_t n, int32_t *out) { return xform_host<0, 4>(ctx, in, n, out, "jmhip_forward4x4"); }
extern "C" int jmhip_inverse4x4(jmhip_ctx *ctx, const int32_t *in, int32_t n, int32_t *out) { return xform_host<1, 4>(ctx, in, n, out, "jmhip_inverse4x4"); }
extern "C" int jmhip_forward8x8(jmhip_ctx *ctx, const int32_t *in, int32_t n, int32_t *out) { return xform_host<0, 8>(ctx, in, n, out, "jmhip_forward8x8"); }
extern "C" int jmhip_inverse8x8(jmhip_ctx *ctx, const int32_t *in, int32_t n, int32_t *out) { return xform_host<1, 8>(ctx, in, n, out, "jmhip_inverse8x8"); }
