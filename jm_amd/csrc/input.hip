// input.hip -- the source picture: a frame as it lies in the file -> the coded-size planes on the device (gfx950).
//
// Device counterpart of (reference):
//   read_one_frame   lcommon/src/input.c:792-868   hands the Y, U, V parts of the file buffer to buf2img
//   buf2img_basic    lcommon/src/input.c:552-600   one byte per sample, source size == output size: a straight copy
//   pad_borders      lcommon/src/input.c:880-925   coded size > picture size (not a multiple of 16): every sample right of the picture
//                                                  repeats its left neighbour, every row below it repeats the row above
// which together are out[y][x] = src[min(y, src_h - 1)][min(x, src_w - 1)] per plane.  A copy: HBM-bound, 1.5 bytes read and written per
// luma sample position (4.7 MB + 4.7 MB at 1080p 4:2:0); a thread writes four samples of one row.
#include "jmhip_internal.h"

__global__ __launch_bounds__(256) void k_load_frame(const uint8_t *__restrict__ raw, int sw, int sh, int scw, int sch,
                                                    uint8_t *__restrict__ y, int pitch_y, int W, int H,
                                                    uint8_t *__restrict__ u, uint8_t *__restrict__ v, int pitch_c, int cw, int ch)
{
  int row = blockIdx.y;                                  // rows of Y, then U, then V
  const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const uint8_t *src; uint8_t *dst; int w, srcw, srch, dpitch;
  if (row < H) { src = raw; dst = y; w = W; srcw = sw; srch = sh; dpitch = pitch_y; }
  else {
    row -= H;
    const int pl = row >= ch;
    if (pl) row -= ch;
    src = raw + (long)sw * sh + (long)pl * scw * sch; dst = pl ? v : u; w = cw; srcw = scw; srch = sch; dpitch = pitch_c;
  }
  if (x4 >= w) return;
  const uint8_t *s = src + (long)min(row, srch - 1) * srcw;
  uint32_t o = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) o |= (uint32_t)s[min(x4 + k, srcw - 1)] << (8 * k);
  *(uint32_t *)(dst + (long)row * dpitch + x4) = o;
}

static int check_source(jmhip_ctx *ctx, const char *who, const void *raw, int src_w, int src_h)
{
  const int fmt = ctx->cfg.yuv_format;
  if (!raw || src_w <= 0 || src_h <= 0 || src_w > ctx->W || src_h > ctx->H || ctx->W - src_w >= 16 || ctx->H - src_h >= 16 ||
      (fmt && (src_w & 1)) || (fmt == 1 && (src_h & 1)))
    return jmhip_fail(ctx, JMHIP_EINVAL, "%s: source %dx%d does not pad to the coded %dx%d (less than one macroblock of padding, even size for sub-sampled chroma)",
                      who, src_w, src_h, ctx->W, ctx->H);
  return JMHIP_OK;
}

extern "C" int jmhip_set_current_frame_dev(jmhip_ctx *ctx, const uint8_t *d_raw, int32_t src_w, int32_t src_h)
{
  if (!ctx) return JMHIP_EINVAL;
  int r = check_source(ctx, "jmhip_set_current_frame_dev", d_raw, src_w, src_h);
  if (r) return r;
  const int fmt = ctx->cfg.yuv_format;
  if (fmt && !ctx->d_cur_c) HIPCHK(ctx, hipMalloc((void **)&ctx->d_cur_c, (size_t)2 * ctx->cw * ctx->ch));
  const int scw = fmt ? src_w / 2 : 0, sch = fmt == 1 ? src_h / 2 : (fmt == 2 ? src_h : 0);
  hipLaunchKernelGGL(k_load_frame, dim3((ctx->W / 4 + 255) / 256, ctx->H + 2 * ctx->ch), dim3(256), 0, ctx->stream, d_raw, src_w, src_h, scw, sch,
                     ctx->d_cur, ctx->cur_pitch, ctx->W, ctx->H, ctx->d_cur_c, ctx->d_cur_c ? ctx->d_cur_c + (size_t)ctx->cw * ctx->ch : nullptr, ctx->cw, ctx->cw, ctx->ch);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_set_current_frame(jmhip_ctx *ctx, const uint8_t *raw, int32_t src_w, int32_t src_h)
{
  if (!ctx) return JMHIP_EINVAL;
  int r = check_source(ctx, "jmhip_set_current_frame", raw, src_w, src_h);
  if (r) return r;
  const int fmt = ctx->cfg.yuv_format;
  const size_t bytes = (size_t)src_w * src_h + (fmt ? (size_t)2 * (src_w / 2) * (fmt == 1 ? src_h / 2 : src_h) : 0);
  HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage, raw, bytes, hipMemcpyHostToDevice, ctx->stream));      // d_stage holds 3 x W x H bytes
  if ((r = jmhip_set_current_frame_dev(ctx, ctx->d_stage, src_w, src_h))) return r;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

// the current picture from imgpel planes that already have the coded size (p_Vid->pCurImg = pImgOrg[0], pImgOrg[1], pImgOrg[2] after pad_borders)
extern "C" int jmhip_set_current_planes(jmhip_ctx *ctx, const uint16_t *y, int32_t pitch_y, const uint16_t *u, const uint16_t *v, int32_t pitch_c)
{
  if (!ctx) return JMHIP_EINVAL;
  const int fmt = ctx->cfg.yuv_format;
  if (!y || pitch_y < ctx->W || (fmt && (!u || !v || pitch_c < ctx->cw))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_set_current_planes: bad argument");
  JMHIP_DEVICE(ctx);
  const size_t ny = (size_t)ctx->W * ctx->H, nc = (size_t)ctx->cw * ctx->ch;
  if (ny + 2 * nc > ctx->h_stage_bytes) return jmhip_fail(ctx, JMHIP_ENOMEM, "jmhip_set_current_planes: staging area too small");
  if (fmt && !ctx->d_cur_c) HIPCHK(ctx, hipMalloc((void **)&ctx->d_cur_c, 2 * nc));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));            // the staging area may still be in flight
  uint8_t *st = ctx->h_stage;
  for (int j = 0; j < ctx->H; j++) { const uint16_t *s = y + (size_t)j * pitch_y; uint8_t *d = st + (size_t)j * ctx->W; for (int i = 0; i < ctx->W; i++) d[i] = (uint8_t)s[i]; }
  for (int k = 0; k < (fmt ? 2 : 0); k++) {
    const uint16_t *src = k ? v : u;
    for (int j = 0; j < ctx->ch; j++) { const uint16_t *s = src + (size_t)j * pitch_c; uint8_t *d = st + ny + k * nc + (size_t)j * ctx->cw; for (int i = 0; i < ctx->cw; i++) d[i] = (uint8_t)s[i]; }
  }
  HIPCHK(ctx, hipMemcpy2DAsync(ctx->d_cur, ctx->cur_pitch, st, ctx->W, ctx->W, ctx->H, hipMemcpyHostToDevice, ctx->stream));
  if (fmt) HIPCHK(ctx, hipMemcpyAsync(ctx->d_cur_c, st + ny, 2 * nc, hipMemcpyHostToDevice, ctx->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_current_planes_dev(jmhip_ctx *ctx, const uint8_t **d_y, int32_t *pitch_y, const uint8_t **d_u, const uint8_t **d_v, int32_t *pitch_c)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!d_y || !pitch_y) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_current_planes_dev: bad argument");
  *d_y = ctx->d_cur; *pitch_y = ctx->cur_pitch;
  if (d_u) *d_u = ctx->d_cur_c;
  if (d_v) *d_v = ctx->d_cur_c ? ctx->d_cur_c + (size_t)ctx->cw * ctx->ch : nullptr;
  if (pitch_c) *pitch_c = ctx->cw;
  return JMHIP_OK;
}

// host copies as imgpel, tight pitches (W, W/2): what p_Vid->pImgOrg[0..2] hold
extern "C" int jmhip_get_current_planes(jmhip_ctx *ctx, uint16_t *y, uint16_t *u, uint16_t *v)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!y) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_get_current_planes: bad argument");
  if ((u || v) && !ctx->d_cur_c) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_get_current_planes: no chroma planes (jmhip_set_current_frame has not run)");
  uint8_t *st = ctx->h_stage;
  const size_t ny = (size_t)ctx->W * ctx->H, nc = (size_t)ctx->cw * ctx->ch;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy2DAsync(st, ctx->W, ctx->d_cur, ctx->cur_pitch, ctx->W, ctx->H, hipMemcpyDeviceToHost, ctx->stream));
  if (u || v) HIPCHK(ctx, hipMemcpyAsync(st + ny, ctx->d_cur_c, 2 * nc, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < ny; i++) y[i] = st[i];
  if (u) for (size_t i = 0; i < nc; i++) u[i] = st[ny + i];
  if (v) for (size_t i = 0; i < nc; i++) v[i] = st[ny + nc + i];
  return JMHIP_OK;
}

// ------------------------------------------------------------------ the general reader (jmhip_load_frame): every planar case of read_one_frame
// One thread per sample of the coded planes (a copy with a little arithmetic: HBM-bound, symbol_bytes read + 2 written per sample).
struct LoadPlane { long src_off, dst_off; int w, h, ow, oh, cw, ch, shift; };
struct LoadArgs { LoadPlane pl[3]; int nplanes, sb, bitshift_fn; long total; };

__global__ __launch_bounds__(256) void k_load_frame_ex(const uint8_t *__restrict__ raw, LoadArgs a, uint16_t *__restrict__ y, uint16_t *__restrict__ u, uint16_t *__restrict__ v)
{
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.total) return;
  int k = 0;
  if (a.nplanes > 1 && i >= a.pl[1].dst_off) k = i >= a.pl[2].dst_off ? 2 : 1;
  const LoadPlane p = a.pl[k];
  const long e = i - p.dst_off;
  const int r = (int)(e / p.cw), c = (int)(e - (long)r * p.cw);
  const int rr = min(r, p.oh - 1), cc = min(c, p.ow - 1);          // pad_borders: right of / below the picture repeats the last column / row
  const bool same = p.w == p.ow && p.h == p.oh;
  long si = -1;                                                   // index of the file sample that lands at (rr, cc), or none
  long sbyte = -1;                                                // ... or its byte offset where JM itself counts in bytes
  if (!a.bitshift_fn && a.sb == 2 && same) {                      // buf2img_basic's single memcpy: w * h samples back to back in rows that are cw apart
    const long f = (long)rr * p.cw + cc;
    if (f < (long)p.w * p.h) si = f;
  } else {
    const int iw = min(p.w, p.ow), ih = min(p.h, p.oh);
    const int dx = (!same && p.ow >= p.w) ? (p.ow - p.w) >> 1 : 0, dy = (!same && p.oh >= p.h) ? (p.oh - p.h) >> 1 : 0;
    if (rr >= dy && rr < dy + ih && cc >= dx && cc < dx + iw) {
      si = (long)(rr - dy) * p.w + (cc - dx);
      // buf2img_basic :586-588 (imgpel-sized samples, sizes differ): the row is taken at temp_buf[row * size_x], an unsigned char pointer -- row * size_x BYTES into the plane
      if (!a.bitshift_fn && a.sb == 2) sbyte = (long)(rr - dy) * p.w + 2 * (cc - dx);
    }
  }
  int val = 0;
  if (si >= 0) {
    const uint8_t *s = raw + p.src_off + (sbyte >= 0 ? sbyte : si * a.sb);
    val = a.sb == 1 ? s[0] : (int)s[0] | ((int)s[1] << 8);
    if (a.bitshift_fn) val = p.shift > 0 ? (val + (1 << (p.shift - 1))) >> p.shift : val << (-p.shift);       // rshift_rnd
  }
  (k == 0 ? y : (k == 1 ? u : v))[e] = (uint16_t)val;
}

static int load_args(jmhip_ctx *ctx, const char *who, const jmhip_frame_format *f, LoadArgs &a, size_t &raw_bytes, size_t (&n)[3])
{
  if (!f) return jmhip_fail(ctx, JMHIP_EINVAL, "%s: null format", who);
  if (f->yuv_format < 0 || f->yuv_format > 3 || (f->symbol_bytes != 1 && f->symbol_bytes != 2) || f->src_w < 1 || f->src_h < 1 || f->out_w < 1 || f->out_h < 1 ||
      f->coded_w < f->out_w || f->coded_h < f->out_h || (f->coded_w & 15) || (f->coded_h & 15) || f->coded_w - f->out_w >= 16 || f->coded_h - f->out_h >= 16)
    return jmhip_fail(ctx, JMHIP_EINVAL, "%s: format %d, %dx%d -> %dx%d in %dx%d, %d byte(s) per sample", who, f->yuv_format, f->src_w, f->src_h, f->out_w, f->out_h, f->coded_w, f->coded_h, f->symbol_bytes);
  const int sx = (f->yuv_format == 1 || f->yuv_format == 2) ? 1 : 0, sy = f->yuv_format == 1 ? 1 : 0;
  if (sx && ((f->src_w | f->out_w) & 1)) return jmhip_fail(ctx, JMHIP_EINVAL, "%s: odd width with sub-sampled chroma", who);
  if (sy && ((f->src_h | f->out_h) & 1)) return jmhip_fail(ctx, JMHIP_EINVAL, "%s: odd height with 4:2:0", who);
  a.nplanes = f->yuv_format ? 3 : 1; a.sb = f->symbol_bytes;
  a.bitshift_fn = !(f->src_depth[0] == f->out_depth[0] && f->src_depth[1] == f->out_depth[1]);     // initInput lcommon/src/input.c:41-53
  long so = 0, dof = 0;
  for (int k = 0; k < 3; k++) {
    const int c = k ? 1 : 0;
    LoadPlane &p = a.pl[k];
    p.w = c ? f->src_w >> sx : f->src_w; p.h = c ? f->src_h >> sy : f->src_h;
    p.ow = c ? f->out_w >> sx : f->out_w; p.oh = c ? f->out_h >> sy : f->out_h;
    p.cw = c ? f->coded_w >> sx : f->coded_w; p.ch = c ? f->coded_h >> sy : f->coded_h;
    p.shift = f->src_depth[k] - f->out_depth[k];
    if (k < a.nplanes && (f->src_depth[k] < 1 || f->src_depth[k] > 8 * f->symbol_bytes || f->out_depth[k] < 1 || f->out_depth[k] > 16 || (a.bitshift_fn && 8 * f->symbol_bytes - p.shift > 16)))
      return jmhip_fail(ctx, JMHIP_EUNSUPPORTED, "%s: %d-bit samples in %d byte(s) to %d bits (input.c:440-443: would not fit imgpel)", who, f->src_depth[k], f->symbol_bytes, f->out_depth[k]);
    if (k >= a.nplanes) { p.src_off = so; p.dst_off = dof; n[k] = 0; continue; }
    p.src_off = so; p.dst_off = dof;
    so += (long)p.w * p.h * f->symbol_bytes;
    n[k] = (size_t)p.cw * p.ch;
    dof += (long)n[k];
  }
  a.total = dof; raw_bytes = (size_t)so;
  return JMHIP_OK;
}

extern "C" int jmhip_load_frame_dev(jmhip_ctx *ctx, const jmhip_frame_format *f, const uint8_t *d_raw, uint16_t *d_y, uint16_t *d_u, uint16_t *d_v)
{
  if (!ctx) return JMHIP_EINVAL;
  LoadArgs a; size_t rb, n[3];
  int r = load_args(ctx, "jmhip_load_frame_dev", f, a, rb, n);
  if (r) return r;
  if (!d_raw || !d_y || (a.nplanes == 3 && (!d_u || !d_v))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_load_frame_dev: null plane");
  // the kernel indexes each plane from its own base
  LoadArgs b = a;
  hipLaunchKernelGGL(k_load_frame_ex, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, ctx->stream, d_raw, b, d_y, d_u, d_v);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_load_frame(jmhip_ctx *ctx, const jmhip_frame_format *f, const uint8_t *raw, uint16_t *y, uint16_t *u, uint16_t *v)
{
  if (!ctx) return JMHIP_EINVAL;
  LoadArgs a; size_t rb, n[3];
  int r = load_args(ctx, "jmhip_load_frame", f, a, rb, n);
  if (r) return r;
  if (!raw || !y || (a.nplanes == 3 && (!u || !v))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_load_frame: null plane");
  uint8_t *d_raw = nullptr; uint16_t *d_out = nullptr;
  const size_t tot = n[0] + n[1] + n[2];
  HIPCHK(ctx, hipMalloc((void **)&d_raw, rb));
  if (hipMalloc((void **)&d_out, tot * 2) != hipSuccess) { (void)hipFree(d_raw); return jmhip_fail(ctx, JMHIP_ENOMEM, "jmhip_load_frame: %zu bytes of device memory", tot * 2); }
  hipError_t e = hipMemcpyAsync(d_raw, raw, rb, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) { r = jmhip_load_frame_dev(ctx, f, d_raw, d_out, d_out + n[0], d_out + n[0] + n[1]); if (r) { (void)hipFree(d_raw); (void)hipFree(d_out); return r; } }
  if (e == hipSuccess) e = hipMemcpyAsync(y, d_out, n[0] * 2, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && a.nplanes == 3) e = hipMemcpyAsync(u, d_out + n[0], n[1] * 2, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && a.nplanes == 3) e = hipMemcpyAsync(v, d_out + n[0] + n[1], n[2] * 2, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_raw); (void)hipFree(d_out);
  HIPCHK(ctx, e);
  return JMHIP_OK;
}
