// distortion.hip -- the block distortions JM's mode decision calls through VideoParameters.distortion4x4 / distortion8x8
// (lencod/inc/global.h:1470-1471, bound in lencod/src/me_distortion.c:148-166), batched over difference blocks (gfx950):
//   distortion4x4SAD :38   distortion4x4SSE :52   distortion4x4SATD :66 (HadamardSAD4x4 :175-258)
//   distortion8x8SAD :107  distortion8x8SSE :121  distortion8x8SATD :135 (HadamardSAD8x8 :266-341)
// each returning dist_scale(value) = value << 5 (LAMBDA_ACCURACY_BITS).  One lane per block; 32 + 8 (4x4) or 128 + 8 (8x8)
// algorithmic bytes per block: a stream.
#include "jmhip_internal.h"

__device__ __forceinline__ int iabsd_(int v) { return v < 0 ? -v : v; }

template <int N>                      // N = 4 or 8: one-dimensional Hadamard butterflies in JM's order of additions
__device__ __forceinline__ void had1d(int *v)
{
  if (N == 4) {
    const int s0 = v[0] + v[3], s1 = v[1] + v[2], s2 = v[1] - v[2], s3 = v[0] - v[3];
    v[0] = s0 + s1; v[1] = s0 - s1; v[2] = s2 + s3; v[3] = s3 - s2;
  } else {
    int a[8];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = v[i] + v[i + 4]; a[i + 4] = v[i] - v[i + 4]; }
    const int b[8] = {a[0] + a[2], a[1] + a[3], a[0] - a[2], a[1] - a[3], a[4] + a[6], a[5] + a[7], a[4] - a[6], a[5] - a[7]};
#pragma unroll
    for (int i = 0; i < 4; i++) { v[2 * i] = b[2 * i] + b[2 * i + 1]; v[2 * i + 1] = b[2 * i] - b[2 * i + 1]; }
  }
}

template <int N>
__global__ __launch_bounds__(256) void k_distortion(int metric, const int16_t *__restrict__ diff, int n, int64_t *__restrict__ out)
{
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  int m[N * N];
  {
    const uint4 *p = (const uint4 *)(diff + (long)b * N * N);
#pragma unroll
    for (int k = 0; k < N * N / 8; k++) {
      const uint4 v = p[k];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; i++) { m[8 * k + 2 * i] = (int)(int16_t)(w[i] & 0xffff); m[8 * k + 2 * i + 1] = (int)(int16_t)(w[i] >> 16); }
    }
  }
  long long d = 0;
  if (metric == JMHIP_METRIC_SAD) {
#pragma unroll
    for (int k = 0; k < N * N; k++) d += iabsd_(m[k]);
  } else if (metric == JMHIP_METRIC_SSE) {
#pragma unroll
    for (int k = 0; k < N * N; k++) d += (long long)m[k] * m[k];
  } else {
    int s = 0;
#pragma unroll
    for (int j = 0; j < N; j++) had1d<N>(m + N * j);
#pragma unroll
    for (int i = 0; i < N; i++) {
      int v[N];
#pragma unroll
      for (int j = 0; j < N; j++) v[j] = m[N * j + i];
      had1d<N>(v);
#pragma unroll
      for (int j = 0; j < N; j++) s += iabsd_(v[j]);
    }
    d = N == 4 ? (s + 1) >> 1 : (s + 2) >> 2;
  }
  out[b] = d << 5;
}

extern "C" int jmhip_distortion(jmhip_ctx *ctx, int32_t metric, int32_t size, const int16_t *diff, int32_t n, int64_t *out)
{
  if (!ctx) return JMHIP_EINVAL;
  if (n < 0 || (n > 0 && (!diff || !out)) || (size != 4 && size != 8) || metric < JMHIP_METRIC_SAD || metric > JMHIP_METRIC_SATD)
    return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_distortion: bad argument (metric %d, block size %d)", metric, size);
  if (n == 0) return JMHIP_OK;
  int r; void *din, *dout;
  const size_t in_bytes = (size_t)n * size * size * sizeof(int16_t);
  if ((r = jmhip_scratch(ctx, 0, in_bytes, &din))) return r;
  if ((r = jmhip_scratch(ctx, 1, (size_t)n * sizeof(int64_t), &dout))) return r;
  HIPCHK(ctx, hipMemcpyAsync(din, diff, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (size == 4) hipLaunchKernelGGL(k_distortion<4>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, metric, (const int16_t *)din, n, (int64_t *)dout);
  else hipLaunchKernelGGL(k_distortion<8>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, metric, (const int16_t *)din, n, (int64_t *)dout);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}
