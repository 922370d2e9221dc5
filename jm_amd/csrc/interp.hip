// interp.hip -- K5: the 16 quarter-pel luma planes of a reference picture (gfx950).
//
// Device counterpart of getSubImagesLuma (reference: lencod/src/img_luma.c:611-679 and its
// helpers :40-596).  All of JM's border special cases are index clamps into the padded
// plane, and the padded integer plane is itself an edge-replicated copy of the picture, so
// the whole function is a stencil over an infinitely edge-replicated source:
//   P00 = src                                   P02 = clip((h+16)>>5),  h = 6-tap(1,-5,20,20,-5,1) along x
//   P20 = clip((v+16)>>5), v = 6-tap along y    P22 = clip((vv+512)>>10), vv = 6-tap along y of the UNCLIPPED h
//   quarter planes = (a+b+1)>>1 of the pairs listed at img_luma.c:653-678.
//
// Layout: 16 planes of Hp x pitch bytes, plane (j,i) at base + (j*4+i)*plane_stride, row 0 / col 0
// = padded origin (picture sample (0,0) sits at row 20, col 32).
//
// One workgroup = one 64x16 output tile of all 16 planes.  The source tile (+halo) and the
// unclipped horizontal sums are staged in LDS, so HBM sees one read of the source and one
// write of each plane: algorithmic bytes = W*H + 16*Wp*Hp per reference picture.
#include "jmhip_internal.h"

#define TW 64
#define TH 16
#define SROWS (TH + 5)       // source rows y0-2 .. y0+TH+2
#define SCOLS (TW + 6)       // source cols x0-2 .. x0+TW+3
#define SPITCH 72

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int clip255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int tap6(int c, int b, int a, int d, int e, int f) { return 20 * (a + d) - 5 * (b + e) + (c + f); }

__global__ __launch_bounds__(256) void k_subplanes(const uint8_t *__restrict__ src, int src_pitch, int W, int H,
                                                   uint8_t *__restrict__ dst, int pitch, long plane_stride, int Wp, int Hp)
{
  __shared__ uint8_t s_src[SROWS][SPITCH];
  __shared__ int16_t s_h[SROWS][TW];
  __shared__ uint8_t s_v[TH][TW + 8];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;      // padded coordinates of the tile origin

  // source tile with halo, coordinates clamped into the picture (= infinite edge replication)
  for (int k = tid; k < SROWS * SCOLS; k += 256) {
    int r = k / SCOLS, c = k - r * SCOLS;
    int sy = clampi(y0 - 2 + r - JMHIP_PAD_Y, 0, H - 1), sx = clampi(x0 - 2 + c - JMHIP_PAD_X, 0, W - 1);
    s_src[r][c] = src[(long)sy * src_pitch + sx];
  }
  __syncthreads();
  // unclipped horizontal six-tap sums for all SROWS rows (imgY_sub_tmp, img_luma.c:151-236)
  for (int k = tid; k < SROWS * TW; k += 256) {
    int r = k / TW, c = k - r * TW;                            // h at padded col x0+c needs src cols c .. c+5 of the tile
    const uint8_t *p = &s_src[r][c];
    s_h[r][c] = (int16_t)tap6(p[0], p[1], p[2], p[3], p[4], p[5]);
  }
  // vertical six-tap on the integer samples for cols x0 .. x0+TW (one extra for the x+1 neighbours)
  for (int k = tid; k < TH * (TW + 1); k += 256) {
    int r = k / (TW + 1), c = k - r * (TW + 1);                // P20 at tile row r, col c -> src tile col c+2, rows r .. r+5
    int v = tap6(s_src[r][c + 2], s_src[r + 1][c + 2], s_src[r + 2][c + 2], s_src[r + 3][c + 2], s_src[r + 4][c + 2], s_src[r + 5][c + 2]);
    s_v[r][c] = (uint8_t)clip255((v + 16) >> 5);
  }
  __syncthreads();

  // each thread produces four horizontally adjacent samples of one row and stores them as one dword per plane: a wave
  // writes 4 rows x 64 contiguous bytes per plane (the store path is what bounds this kernel)
  const int tx = (tid & 15) * 4, r = tid >> 4;
  const int x = x0 + tx, y = y0 + r;
  if (x >= Wp || y >= Hp) return;
  uint32_t o[16];
#pragma unroll
  for (int k = 0; k < 16; k++) o[k] = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int c = tx + q;
    const int p00 = s_src[r + 2][c + 2], p00r = s_src[r + 2][c + 3], p00d = s_src[r + 3][c + 2];
    const int p02 = clip255((s_h[r + 2][c] + 16) >> 5), p02d = clip255((s_h[r + 3][c] + 16) >> 5);
    const int p20 = s_v[r][c], p20r = s_v[r][c + 1];
    const int vv = tap6(s_h[r][c], s_h[r + 1][c], s_h[r + 2][c], s_h[r + 3][c], s_h[r + 4][c], s_h[r + 5][c]);
    const int p22 = clip255((vv + 512) >> 10);
#define ST(j, i, val) o[(j) * 4 + (i)] |= (uint32_t)(val) << (8 * q)
#define AV(a, b) (((a) + (b) + 1) >> 1)
    ST(0, 0, p00);            ST(0, 1, AV(p00, p02));   ST(0, 2, p02);            ST(0, 3, AV(p02, p00r));
    ST(1, 0, AV(p00, p20));   ST(1, 1, AV(p02, p20));   ST(1, 2, AV(p02, p22));   ST(1, 3, AV(p02, p20r));
    ST(2, 0, p20);            ST(2, 1, AV(p20, p22));   ST(2, 2, p22);            ST(2, 3, AV(p22, p20r));
    ST(3, 0, AV(p20, p00d));  ST(3, 1, AV(p20, p02d));  ST(3, 2, AV(p22, p02d));  ST(3, 3, AV(p02d, p20r));
#undef ST
#undef AV
  }
  uint8_t *op = dst + (long)y * pitch + x;
#pragma unroll
  for (int k = 0; k < 16; k++) *(uint32_t *)(op + k * plane_stride) = o[k];
}

int jmhip_launch_subplanes(jmhip_ctx *ctx, const uint8_t *d_luma, int pitch, uint8_t *d_planes)
{
  dim3 grid((ctx->Wp + TW - 1) / TW, (ctx->Hp + TH - 1) / TH);
  jmhip_time_begin(ctx, 0);
  hipLaunchKernelGGL(k_subplanes, grid, dim3(256), 0, ctx->stream, d_luma, pitch, ctx->W, ctx->H, d_planes, ctx->pitch,
                     (long)ctx->plane_stride, ctx->Wp, ctx->Hp);
  jmhip_time_end(ctx, 0);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}
