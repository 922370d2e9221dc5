// ctx.hip -- context, device-resident frames, host<->device staging for libjmhip.
#include <stdarg.h>
#include <stdlib.h>
#include <time.h>
#include "jmhip_internal.h"

char g_jmhip_create_err[512] = "";
int g_jmhip_multi_device = 0;      // a context on a device other than 0 has been created in this process

int jmhip_fail(jmhip_ctx *ctx, int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(ctx ? ctx->err : g_jmhip_create_err, 512, fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char *jmhip_last_error(const jmhip_ctx *ctx) { return ctx ? ctx->err : g_jmhip_create_err; }

int jmhip_scratch(jmhip_ctx *ctx, int which, size_t bytes, void **out)
{
  void **p = which ? &ctx->d_scratch2 : &ctx->d_scratch;
  size_t *sz = which ? &ctx->scratch2_bytes : &ctx->scratch_bytes;
  if (*sz < bytes) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (*p) HIPCHK(ctx, hipFree(*p));
    *p = NULL; *sz = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(ctx, hipMalloc(p, want));
    *sz = want;
  }
  *out = *p;
  return JMHIP_OK;
}

void jmhip_time_begin(jmhip_ctx *ctx, int kind)
{
  if (ctx->timing) (void)hipEventRecord(ctx->ev0[kind], ctx->stream);
}
void jmhip_time_end(jmhip_ctx *ctx, int kind)
{
  if (ctx->timing) { (void)hipEventRecord(ctx->ev1[kind], ctx->stream); ctx->ev_valid[kind] = 1; }
}

static void spiral_fill(int R, int16_t *sp)   // JM spiral order, lencod/src/mv_search.c:405-442
{
  int k = 1;
  sp[0] = sp[1] = 0;
  for (int l = 1; l <= (R > 1 ? R : 1); l++) {
    for (int i = -l + 1; i < l; i++) {
      sp[2 * k] = (int16_t)i;  sp[2 * k + 1] = (int16_t)-l; k++;
      sp[2 * k] = (int16_t)i;  sp[2 * k + 1] = (int16_t)l;  k++;
    }
    for (int i = -l; i <= l; i++) {
      sp[2 * k] = (int16_t)-l; sp[2 * k + 1] = (int16_t)i; k++;
      sp[2 * k] = (int16_t)l;  sp[2 * k + 1] = (int16_t)i; k++;
    }
  }
}

extern "C" int jmhip_create(jmhip_ctx **out, const jmhip_config *cfg)
{
  if (!out || !cfg) return jmhip_fail(NULL, JMHIP_EINVAL, "jmhip_create: null argument");
  *out = NULL;
  if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 15) || (cfg->height & 15))
    return jmhip_fail(NULL, JMHIP_EINVAL, "width/height must be positive multiples of 16 (got %dx%d)", cfg->width, cfg->height);
  if (cfg->bit_depth != 8) return jmhip_fail(NULL, JMHIP_EUNSUPPORTED, "bit_depth %d: only 8-bit video is implemented", cfg->bit_depth);
  if (cfg->yuv_format < 0 || cfg->yuv_format > 2) return jmhip_fail(NULL, JMHIP_EUNSUPPORTED, "yuv_format %d unsupported", cfg->yuv_format);
  if (cfg->search_range < 1 || cfg->search_range > JMHIP_MAX_SEARCH_RANGE)
    return jmhip_fail(NULL, JMHIP_EINVAL, "search_range %d outside 1..%d", cfg->search_range, JMHIP_MAX_SEARCH_RANGE);
  if (cfg->num_ref_slots < 1 || cfg->num_ref_slots > 32) return jmhip_fail(NULL, JMHIP_EINVAL, "num_ref_slots %d outside 1..32", cfg->num_ref_slots);
  static const bool iprof = getenv("JMHIP_INIT_PROF") != nullptr;      // measurement aid (profiles/r05_init_prof.sh): where jmhip_create's time goes
  struct timespec ip0; clock_gettime(CLOCK_MONOTONIC, &ip0);
#define IPROF(what) do { if (iprof) { struct timespec t_; clock_gettime(CLOCK_MONOTONIC, &t_); fprintf(stderr, "jmhip_create: %7.1f ms  %s\n", 1e3 * (t_.tv_sec - ip0.tv_sec) + 1e-6 * (t_.tv_nsec - ip0.tv_nsec), what); } } while (0)
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  IPROF("hipGetDeviceCount (the runtime starts)");
  if (e != hipSuccess || ndev <= 0)
    return jmhip_fail(NULL, JMHIP_ENODEV, "no HIP device (%s); libjmhip has no CPU fallback", e == hipSuccess ? "count 0" : hipGetErrorString(e));
  if (cfg->device < 0 || cfg->device >= ndev) return jmhip_fail(NULL, JMHIP_ENODEV, "device %d not in 0..%d", cfg->device, ndev - 1);
  if ((e = hipSetDevice(cfg->device)) != hipSuccess) return jmhip_fail(NULL, JMHIP_EHIP, "hipSetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, cfg->device)) != hipSuccess) return jmhip_fail(NULL, JMHIP_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return jmhip_fail(NULL, JMHIP_ENODEV, "device %d is %s; libjmhip is built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);

  IPROF("hipSetDevice, hipGetDeviceProperties");
  jmhip_ctx *c = (jmhip_ctx *)calloc(1, sizeof(jmhip_ctx));
  if (!c) return jmhip_fail(NULL, JMHIP_ENOMEM, "out of host memory");
  c->cfg = *cfg;
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (cfg->device != 0) g_jmhip_multi_device = 1;
  c->stream = (hipStream_t)cfg->stream;
  { const char *fg = getenv("JMHIP_FORCE_GENERIC"); c->force_generic = fg && fg[0] == '1'; }
  { const char *fg = getenv("JMHIP_DEBLOCK_DIAG"); c->force_db_diag = fg && fg[0] == '1'; }
  c->db_no_prefill = getenv("JMHIP_DEBLOCK_NO_PREFILL") != nullptr;
  { const char *e = getenv("JMHIP_DEBLOCK_SPARSE_PCT"); c->db_sparse_pct = e ? atoi(e) : 40; }
  c->refine_per_block = getenv("JMHIP_REFINE_PER_BLOCK") != nullptr;
  { const char *e = getenv("JMHIP_MB_PROF"); c->mb_prof_mode = e ? atoi(e) : 0; }
  c->W = cfg->width; c->H = cfg->height;
  c->Wp = c->W + 2 * JMHIP_PAD_X; c->Hp = c->H + 2 * JMHIP_PAD_Y;
  c->pitch = (c->Wp + 63) & ~63;
  c->plane_stride = (((int64_t)c->pitch * c->Hp) + 255) & ~(int64_t)255;
  c->cw = cfg->yuv_format ? c->W / 2 : 0;
  c->ch = cfg->yuv_format == 2 ? c->H : (cfg->yuv_format == 1 ? c->H / 2 : 0);
  c->cur_pitch = (c->W + 63) & ~63;
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { jmhip_fail(NULL, JMHIP_EHIP, "%s: %s", #call, hipGetErrorString(e_)); jmhip_destroy(c); return JMHIP_EHIP; } } while (0)
  CK(hipMalloc((void **)&c->d_cur, (size_t)c->cur_pitch * c->H));
  c->d_sub = (uint8_t **)calloc(cfg->num_ref_slots, sizeof(uint8_t *));
  c->d_refc = (uint8_t **)calloc(cfg->num_ref_slots, sizeof(uint8_t *));
  for (int s = 0; s < cfg->num_ref_slots; s++) {
    CK(hipMalloc((void **)&c->d_sub[s], (size_t)c->plane_stride * 16));
    CK(hipMemsetAsync(c->d_sub[s], 0, (size_t)c->plane_stride * 16, c->stream));
  }
  IPROF("reference slots: hipMalloc + hipMemsetAsync (the stream's queue, the fill kernel)");
  // pinned staging: one padded plane or one picture's three planes as bytes at a time (every user checks its size against h_stage_bytes).  It was sixteen padded planes of
  // uint16 (75 MB at 1080p: 10 - 15 ms of page pinning in front of a sequence's first picture, profiles/r05_init_prof.txt) although nothing ever staged more than one plane
  {
    const size_t a = (size_t)2 * c->Wp * c->Hp, b = (size_t)3 * c->cur_pitch * c->H;
    c->h_stage_bytes = (a > b ? a : b) + 4096;
    if (c->h_stage_bytes < ((size_t)1 << 18)) c->h_stage_bytes = (size_t)1 << 18;      // ... and the spiral table below (search range 64: 66 KB)
  }
  CK(hipHostMalloc((void **)&c->h_stage, c->h_stage_bytes, hipHostMallocDefault));
  IPROF("hipHostMalloc of the staging area");
  CK(hipMalloc((void **)&c->d_stage, (size_t)c->cur_pitch * c->H * 3));
  {
    // (through the pinned staging area on the context's stream: a synchronous copy from pageable memory sets up the runtime's own staging and the null stream's queue,
    // 11 ms before a sequence's first picture)
    int R = cfg->search_range, n = (2 * R + 1) * (2 * R + 1);
    if ((size_t)(n > 9 ? n : 9) * 4 > c->h_stage_bytes) { jmhip_fail(NULL, JMHIP_EINVAL, "spiral table larger than staging"); jmhip_destroy(c); return JMHIP_EINVAL; }
    spiral_fill(R, (int16_t *)c->h_stage);
    CK(hipMalloc((void **)&c->d_spiral, (size_t)n * 4));
    CK(hipMemcpyAsync(c->d_spiral, c->h_stage, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));      // (the stream is synchronised below, before anybody reuses the staging area)
  }
  IPROF("spiral table");
  CK(hipMalloc((void **)&c->d_me_declined, 64));
  CK(hipMemsetAsync(c->d_me_declined, 0, 64, c->stream));
  CK(hipMemsetAsync(c->d_me_declined + 5, 0xff, 4, c->stream));       // index of the first bad job: none
  CK(hipMalloc(&c->d_db_prep, (size_t)(c->W / 16) * (c->H / 16) * 192));
  CK(hipMalloc((void **)&c->d_db_sync, 64 + (size_t)(c->H / 16) * 2 * 6 * 8));
  CK(hipMalloc(&c->d_db_hand, (size_t)(c->W / 16) * (c->H / 16) * 192));
  CK(hipMalloc((void **)&c->d_db_flags, (size_t)(c->W / 16) * (c->H / 16) * 2 + 16));
  CK(hipMalloc(&c->d_db_tasks, 1024 * 8 + 256 * 8 * 8));      // task list + the task builder's row masks
  for (int k = 0; k < JMHIP_NKINDS; k++) { CK(hipEventCreate(&c->ev0[k])); CK(hipEventCreate(&c->ev1[k])); }
  IPROF("small buffers, events");
  CK(hipStreamSynchronize(c->stream));
  IPROF("hipStreamSynchronize");
#undef CK
#undef IPROF
  *out = c;
  return JMHIP_OK;
}

extern "C" void jmhip_destroy(jmhip_ctx *c)
{
  if (!c) return;
  (void)hipStreamSynchronize(c->stream);
  // pictures in flight run on streams of their own (jmhip_seq_open) and read / write the slots freed below: nothing of theirs may still be running (the adapter's full
  // teardown no longer has jmhip_synchronize in front of it: round 6)
  if (c->seq) (void)jmhip_seq_sync_all(c);
  if (c->d_cur) (void)hipFree(c->d_cur);
  if (c->d_cur_c) (void)hipFree(c->d_cur_c);
  if (c->d_sub) { for (int s = 0; s < c->cfg.num_ref_slots; s++) if (c->d_sub[s]) (void)hipFree(c->d_sub[s]); free(c->d_sub); }
  if (c->d_refc) { for (int s = 0; s < c->cfg.num_ref_slots; s++) if (c->d_refc[s]) (void)hipFree(c->d_refc[s]); free(c->d_refc); }
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->d_scratch2) (void)hipFree(c->d_scratch2);
  if (c->d_spiral) (void)hipFree(c->d_spiral);
  if (c->d_db_prep) (void)hipFree(c->d_db_prep);
  if (c->d_me_declined) (void)hipFree(c->d_me_declined);
  if (c->d_db_sync) (void)hipFree(c->d_db_sync);
  if (c->d_db_hand) (void)hipFree(c->d_db_hand);
  if (c->d_db_flags) (void)hipFree(c->d_db_flags);
  if (c->d_db_tasks) (void)hipFree(c->d_db_tasks);
  jmhip_mb_free(c);
  for (int k = 0; k < JMHIP_NKINDS; k++) { if (c->ev0[k]) (void)hipEventDestroy(c->ev0[k]); if (c->ev1[k]) (void)hipEventDestroy(c->ev1[k]); }
  free(c);
}

// not part of the ABI: copies the deblocking pipeline's sync / profile words to the host (profiles/prof_deblock.py)
extern "C" int jmhip_debug_read_db_sync(jmhip_ctx *ctx, void *out, size_t bytes)
{
  if (!ctx || !out || bytes > 64 + (size_t)(ctx->H / 16) * 2 * 6 * 8) return JMHIP_EINVAL;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy(out, ctx->d_db_sync, bytes, hipMemcpyDeviceToHost));
  return JMHIP_OK;
}

// The deblocking row pipeline raises a device-side error word when a bounded wait runs out (it never hangs); it is read
// back at the next synchronisation point of a context that has launched the pipeline.
int jmhip_check_deblock_error(jmhip_ctx *ctx)
{
  if (!ctx->db_launched) return JMHIP_OK;
  unsigned words[2] = {0, 0};
  HIPCHK(ctx, hipMemcpyAsync(words, ctx->d_db_sync, sizeof words, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->db_launched = 0;
  if (words[1] != 0) return jmhip_fail(ctx, JMHIP_EHIP, "deblocking row pipeline: a hand-over wait timed out (error word %u); the frame is incomplete", words[1]);
  return JMHIP_OK;
}

extern "C" int jmhip_set_stream(jmhip_ctx *ctx, void *hip_stream)
{
  if (!ctx) return JMHIP_EINVAL;
  ctx->stream = (hipStream_t)hip_stream;
  return JMHIP_OK;
}

extern "C" int jmhip_synchronize(jmhip_ctx *ctx)
{
  if (!ctx) return JMHIP_EINVAL;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int r = jmhip_seq_sync_all(ctx);
  if (r) return r;
  // pictures in flight: what jmhip_seq_wait would have told about each of them (the first error is reported, every entry is taken out of flight: a sticky error word
  // would otherwise make the entry's next launch a silent no-op)
  // (an entry that streams its records to the host keeps doing so: jmhip_seq_record still answers for its picture afterwards -- every flag is set by then)
  for (int k = 0; k < ctx->seq_depth; k++)
    if (ctx->seq[k].in_flight) { const int was = ctx->seq[k].streaming; const int q = jmhip_seq_wait(ctx, k); ctx->seq[k].streaming = was; if (q && !r) r = q; }
  if (r) return r;
  r = jmhip_check_job_error(ctx);
  if (r) return r;
  if ((r = jmhip_check_mb_error(ctx))) return r;
  return jmhip_check_deblock_error(ctx);
}

extern "C" int jmhip_plane_geometry(const jmhip_ctx *ctx, int32_t *pitch, int32_t *rows, int64_t *plane_stride)
{
  if (!ctx) return JMHIP_EINVAL;
  if (pitch) *pitch = ctx->pitch;
  if (rows) *rows = ctx->Hp;
  if (plane_stride) *plane_stride = ctx->plane_stride;
  return JMHIP_OK;
}

extern "C" int jmhip_enable_timing(jmhip_ctx *ctx, int32_t on)
{
  if (!ctx) return JMHIP_EINVAL;
  ctx->timing = on;
  return JMHIP_OK;
}

extern "C" int jmhip_last_kernel_ms(jmhip_ctx *ctx, int32_t kind, float *ms)
{
  if (!ctx || !ms || kind < 0 || kind >= JMHIP_NKINDS) return JMHIP_EINVAL;
  if (!ctx->ev_valid[kind]) return jmhip_fail(ctx, JMHIP_EINVAL, "no timed launch of kind %d yet (jmhip_enable_timing first)", kind);
  HIPCHK(ctx, hipEventSynchronize(ctx->ev1[kind]));
  HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev0[kind], ctx->ev1[kind]));
  return JMHIP_OK;
}

// narrow a host imgpel (uint16) plane into pinned u8 staging and copy it to `dst` (pitch dst_pitch)
static int upload_u16_as_u8(jmhip_ctx *ctx, const uint16_t *src, int pitch_samples, int w, int h, uint8_t *dst, int dst_pitch)
{
  if ((size_t)w * h > ctx->h_stage_bytes) return jmhip_fail(ctx, JMHIP_EINVAL, "plane larger than staging");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // staging buffer reuse
  uint8_t *st = ctx->h_stage;
  for (int y = 0; y < h; y++) {
    const uint16_t *s = src + (size_t)y * pitch_samples;
    uint8_t *d = st + (size_t)y * w;
    for (int x = 0; x < w; x++) d[x] = (uint8_t)s[x];
  }
  HIPCHK(ctx, hipMemcpy2DAsync(dst, dst_pitch, st, w, w, h, hipMemcpyHostToDevice, ctx->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_set_current(jmhip_ctx *ctx, const uint16_t *luma, int32_t pitch_samples)
{
  if (!ctx || !luma || pitch_samples < ctx->W) return ctx ? jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_set_current: bad argument") : JMHIP_EINVAL;
  int r = upload_u16_as_u8(ctx, luma, pitch_samples, ctx->W, ctx->H, ctx->d_cur, ctx->cur_pitch);
  if (r) return r;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_set_current_dev(jmhip_ctx *ctx, const uint8_t *d_luma, int32_t pitch_bytes)
{
  if (!ctx || !d_luma || pitch_bytes < ctx->W) return ctx ? jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_set_current_dev: bad argument") : JMHIP_EINVAL;
  HIPCHK(ctx, hipMemcpy2DAsync(ctx->d_cur, ctx->cur_pitch, d_luma, pitch_bytes, ctx->W, ctx->H, hipMemcpyDeviceToDevice, ctx->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_set_reference_dev(jmhip_ctx *ctx, int32_t slot, const uint8_t *d_luma, int32_t pitch_bytes)
{
  if (!ctx || !d_luma || slot < 0 || slot >= ctx->cfg.num_ref_slots || pitch_bytes < ctx->W)
    return ctx ? jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_set_reference_dev: bad argument") : JMHIP_EINVAL;
  if (ctx->seq) { (void)jmhip_seq_sync_all(ctx); ctx->slot_entry[slot] = -1; }      // no picture in flight reads or writes the slot any more
  jmhip_mb_slot_motion_reset(ctx, slot);
  return jmhip_launch_subplanes(ctx, d_luma, pitch_bytes, ctx->d_sub[slot]);
}

extern "C" int jmhip_set_reference(jmhip_ctx *ctx, int32_t slot, const uint16_t *luma, int32_t pitch_samples)
{
  if (!ctx || !luma || slot < 0 || slot >= ctx->cfg.num_ref_slots || pitch_samples < ctx->W)
    return ctx ? jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_set_reference: bad argument") : JMHIP_EINVAL;
  int r = upload_u16_as_u8(ctx, luma, pitch_samples, ctx->W, ctx->H, ctx->d_stage, ctx->cur_pitch);
  if (r) return r;
  if (ctx->seq) { (void)jmhip_seq_sync_all(ctx); ctx->slot_entry[slot] = -1; }
  jmhip_mb_slot_motion_reset(ctx, slot);
  r = jmhip_launch_subplanes(ctx, ctx->d_stage, ctx->cur_pitch, ctx->d_sub[slot]);
  if (r) return r;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

static int refc_slot(jmhip_ctx *ctx, int slot)
{
  if (!ctx->d_refc[slot]) HIPCHK(ctx, hipMalloc((void **)&ctx->d_refc[slot], (size_t)2 * ctx->cw * ctx->ch));
  return JMHIP_OK;
}
// both chroma planes of a reference into the context's slot (U then V, pitch = width); 8 bytes per thread when everything is 8-byte aligned
__global__ __launch_bounds__(256) void k_copy_chroma_planes(const uint8_t *__restrict__ u, const uint8_t *__restrict__ v, int pitch,
                                                            uint8_t *__restrict__ dst, int cw, int ch, int units, int wide)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= units * ch) return;
  const int y = i / units, x = i - y * units;
  const uint8_t *src = (blockIdx.y ? v : u) + (size_t)y * pitch;
  uint8_t *d = dst + (size_t)blockIdx.y * cw * ch + (size_t)y * cw;
  if (wide) ((uint2 *)d)[x] = ((const uint2 *)src)[x];
  else d[x] = src[x];
}

extern "C" int jmhip_set_reference_chroma_dev(jmhip_ctx *ctx, int32_t slot, const uint8_t *d_u, const uint8_t *d_v, int32_t pitch_bytes)
{
  if (!ctx || !d_u || !d_v || slot < 0 || slot >= ctx->cfg.num_ref_slots || ctx->cfg.yuv_format == 0 || pitch_bytes < ctx->cw)
    return ctx ? jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_set_reference_chroma_dev: bad argument") : JMHIP_EINVAL;
  int r = refc_slot(ctx, slot);
  if (r) return r;
  if (ctx->seq && ctx->slot_entry && ctx->slot_entry[slot] >= 0) {      // a picture in flight reads or writes the slot's chroma: it comes first (as in jmhip_set_reference)
    if ((r = jmhip_seq_sync_all(ctx))) return r;
    ctx->slot_entry[slot] = -1;
  }
  // one launch for both planes (two 2-D copies of the runtime were 5 us each at 1080p)
  const bool wide = !((ctx->cw | pitch_bytes) & 7) && !(((uintptr_t)d_u | (uintptr_t)d_v) & 7);
  const int units = wide ? ctx->cw / 8 : ctx->cw;
  hipLaunchKernelGGL(k_copy_chroma_planes, dim3((units * ctx->ch + 255) / 256, 2), dim3(256), 0, ctx->stream,
                     d_u, d_v, pitch_bytes, ctx->d_refc[slot], ctx->cw, ctx->ch, units, wide ? 1 : 0);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}
extern "C" int jmhip_set_reference_chroma(jmhip_ctx *ctx, int32_t slot, const uint16_t *u, const uint16_t *v, int32_t pitch_samples)
{
  if (!ctx || !u || !v || slot < 0 || slot >= ctx->cfg.num_ref_slots || ctx->cfg.yuv_format == 0 || pitch_samples < ctx->cw)
    return ctx ? jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_set_reference_chroma: bad argument") : JMHIP_EINVAL;
  int r = refc_slot(ctx, slot);
  if (r) return r;
  for (int k = 0; k < 2; k++) {
    r = upload_u16_as_u8(ctx, k ? v : u, pitch_samples, ctx->cw, ctx->ch, ctx->d_refc[slot] + (size_t)k * ctx->cw * ctx->ch, ctx->cw);
    if (r) return r;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

extern "C" const uint8_t *jmhip_subplanes_dev(jmhip_ctx *ctx, int32_t slot)
{
  if (!ctx || slot < 0 || slot >= ctx->cfg.num_ref_slots) return NULL;
  return ctx->d_sub[slot];
}

extern "C" int jmhip_get_subplanes(jmhip_ctx *ctx, int32_t slot, uint16_t *out)
{
  if (!ctx || !out || slot < 0 || slot >= ctx->cfg.num_ref_slots) return ctx ? jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_get_subplanes: bad argument") : JMHIP_EINVAL;
  // planes come back through the pinned staging area as u8 and are widened to imgpel on the host
  size_t plane_px = (size_t)ctx->Wp * ctx->Hp;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 16; k++) {
    HIPCHK(ctx, hipMemcpy2DAsync(ctx->h_stage, ctx->Wp, ctx->d_sub[slot] + (size_t)k * ctx->plane_stride, ctx->pitch,
                                 ctx->Wp, ctx->Hp, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    uint16_t *o = out + (size_t)k * plane_px;
    for (size_t i = 0; i < plane_px; i++) o[i] = ctx->h_stage[i];
  }
  return JMHIP_OK;
}
