// me_fullsearch.hip -- K1+K2+K3: full-search SAD surface, 41-partition aggregation, cost + argmin (gfx950).
//
// Device counterpart of JM's integer-pel search (reference, all under lencod/src):
//   setup_fast_full_search            me_fullfast.c:269-608   sixteen 4x4 SADs at every window position
//   update_full_search_large_blocks   me_fullfast.c:195-260   4x4 -> 4x8/8x4/8x8/8x16/16x8/16x16
//   fast_full_search_motion_estimation me_fullfast.c:618-689  } argmin of (SAD<<5) + lambda*mvbits, first
//   full_search_motion_estimation     me_fullsearch.c:39-103  } spiral index wins ties (strict '<')
//   mv_cost lencod/inc/mv_search.h:100-112, mvbits LUT mv_search.c:366-374 (closed form 2*floor(log2|d|)+3)
//   spiral order mv_search.c:405-442 (closed form in spiral_index() below)
//   UMVLine4X origin clamp lencod/inc/refbuf.h:22-26: on the integer plane the padded picture is an edge
//   replica, so clamping the block origin equals clamping every sample coordinate into the padded plane.
//
// One workgroup (256 threads = 4 waves) = one search window job (jmhip_me_job): the (2R+16)^2 window of
// the integer plane and the 16x16 current macroblock are staged in LDS once (coalesced row reads from
// HBM), every thread owns window positions, keeps the running per-partition minimum as one packed key
// (cost << 16 | spiral index) in registers, and the workgroup min-reduces the 41 keys at the end.
// Algorithmic HBM bytes per job: 256 + (2R+16)^2 in, 328 out; nothing else leaves the CU.
#include "jmhip_internal.h"
#include "me_common.h"

struct JobLds {
  int16_t pred[NP][2];
  uint64_t mask;
  int lambda, max_mvd, cx, cy, R;
};

__device__ __forceinline__ void stage_job(const jmhip_me_job *__restrict__ job, const uint8_t *__restrict__ cur, int cur_pitch,
                                          const uint8_t *__restrict__ ref00, int pitch, long plane_stride, int W, int H,
                                          uint8_t *s_win, int wpitch, uint32_t *s_cur, JobLds *s_job)
{
  const int tid = threadIdx.x;
  const int R = job->search_range;
  if (tid < NP) { s_job->pred[tid][0] = job->pred[tid][0]; s_job->pred[tid][1] = job->pred[tid][1]; }
  if (tid == 0) {
    s_job->mask = job->part_mask; s_job->lambda = job->lambda; s_job->max_mvd = job->max_mvd;
    s_job->cx = job->center_x; s_job->cy = job->center_y; s_job->R = R;
  }
  // current macroblock: 16 rows x 16 bytes, one dword per thread
  if (tid < 64) {
    int r = tid >> 2, c = tid & 3;
    s_cur[tid] = *(const uint32_t *)(cur + (long)(job->mb_y + r) * cur_pitch + job->mb_x + c * 4);
  }
  // search window: rows/cols clamped into the padded plane.  The plane is the integer one unless the
  // level's MV clip (clip_mv_range, conformance.c:640: limits 8191 / 2047 are not multiples of 4) left
  // the centre on a fractional phase, in which case JM's UMVLine4X reads that phase's plane.
  const int wsz = 2 * R + 16;
  const int x0 = job->mb_x + (job->center_x >> 2) - R, y0 = job->mb_y + (job->center_y >> 2) - R;     // picture coordinates
  const uint8_t *plane = ref00 + ((job->center_y & 3) * 4 + (job->center_x & 3)) * plane_stride;
  for (int k = tid; k < wsz * wpitch; k += 256) {
    int r = k / wpitch, c = k - r * wpitch;
    int yy = min(max(y0 + r, -JMHIP_PAD_Y), H + JMHIP_PAD_Y - 1) + JMHIP_PAD_Y;
    int xx = min(max(x0 + c, -JMHIP_PAD_X), W + JMHIP_PAD_X - 1) + JMHIP_PAD_X;
    s_win[k] = plane[(long)yy * pitch + xx];
  }
}

__global__ __launch_bounds__(256) void k_me_fullsearch(const jmhip_me_job *__restrict__ jobs, jmhip_me_result *__restrict__ results,
                                                       const uint8_t *__restrict__ cur, int cur_pitch,
                                                       const uint8_t *__restrict__ ref00, int pitch, long plane_stride, int W, int H,
                                                       const int16_t *__restrict__ spiral, int skip_fast,
                                                       const unsigned *__restrict__ declined, unsigned *__restrict__ declined_next, int njobs, const unsigned *__restrict__ jerr)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (*jerr) return;                                       // a job record failed k_check_me_jobs
  if (skip_fast) {                                         // k_me_fs_fast ran before this launch and counted the jobs it left for us
    if (blockIdx.x == 0 && threadIdx.x == 0) *declined_next = 0;          // the counter of the NEXT launch pair (ping-pong)
    if (*declined == 0) return;
  }
  // after k_me_fs_fast the grid is a few workgroups per compute unit that walk the job list (an empty launch of one workgroup per job
  // took 6.6 us at 1080p); on its own the kernel is launched with one workgroup per job
  for (int jb = blockIdx.x; jb < njobs; jb += gridDim.x) {
  const jmhip_me_job *job = jobs + jb;
  if (skip_fast && job_is_fast(job)) continue;
  __syncthreads();                                         // the previous job's LDS is no longer read
  const int R = job->search_range;
  const int wpitch = (2 * R + 16 + 4 + 3) & ~3;
  uint8_t *s_win = smem;                                                   // (2R+16) x wpitch
  uint32_t *s_cur = (uint32_t *)(smem + (((2 * R + 16) * wpitch + 15) & ~15));   // 64 dwords
  JobLds *s_job = (JobLds *)(s_cur + 64);
  unsigned long long *s_red = (unsigned long long *)(s_job + 1);            // [4][NP]
  stage_job(job, cur, cur_pitch, ref00, pitch, plane_stride, W, H, s_win, wpitch, s_cur, s_job);
  __syncthreads();

  const int tid = threadIdx.x, n1 = 2 * R + 1, npos = n1 * n1;
  const int lambda = s_job->lambda, cx = s_job->cx, cy = s_job->cy, guard = s_job->max_mvd ? s_job->max_mvd - 1 : 0x7fffffff;
  const uint64_t mask = s_job->mask;
  unsigned long long best[NP];
#pragma unroll
  for (int p = 0; p < NP; p++) best[p] = ~0ull;

  for (int pos = tid; pos < npos; pos += 256) {
    const int wy = pos / n1, wx = pos - wy * n1, dx = wx - R, dy = wy - R;
    uint32_t s7[16], sp[NP];
    sad16(s_win, wpitch, s_cur, wx, wy, s7);
    aggregate41(s7, sp);
    const unsigned idx = (unsigned)spiral_index(dx, dy);
    const int candx = cx + 4 * dx, candy = cy + 4 * dy;
#pragma unroll
    for (int p = 0; p < NP; p++) {
      if ((mask >> p) & 1) {                                              // wave-uniform
        int mx = candx - s_job->pred[p][0], my = candy - s_job->pred[p][1];
        if (imax_(iabs_(mx), iabs_(my)) < guard) {
          unsigned cost = (sp[p] << 5) + (unsigned)(lambda * (mvbits(mx) + mvbits(my)));
          unsigned long long key = ((unsigned long long)cost << 16) | idx;
          best[p] = key < best[p] ? key : best[p];
        }
      }
    }
  }
  // workgroup min-reduction of the 41 keys
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int p = 0; p < NP; p++) {
    unsigned long long k = best[p];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      unsigned long long o = __shfl_xor(k, off, 64);
      k = o < k ? o : k;
    }
    if (lane == 0) s_red[wave * NP + p] = k;
  }
  __syncthreads();
  if (tid < NP && ((mask >> tid) & 1)) {
    unsigned long long k = s_red[tid];
    for (int w = 1; w < 4; w++) { unsigned long long o = s_red[w * NP + tid]; k = o < k ? o : k; }
    jmhip_me_best b;
    if (k == ~0ull) { b.mv_x = (int16_t)cx; b.mv_y = (int16_t)cy; b.cost = 0x7fffffff; }       // every candidate guarded out
    else {
      unsigned idx = (unsigned)(k & 0xffff);
      b.mv_x = (int16_t)(cx + 4 * spiral[2 * idx]); b.mv_y = (int16_t)(cy + 4 * spiral[2 * idx + 1]);
      b.cost = (int32_t)(k >> 16);
    }
    results[jb].best[tid] = b;
  }
  }
}

// BlockSAD tables for a host-side argmin: table[7][16][max_pos] uint16, JM's order
__global__ __launch_bounds__(256) void k_me_sad_tables(const jmhip_me_job *__restrict__ jobs, uint16_t *__restrict__ tables, long table_stride,
                                                       const uint8_t *__restrict__ cur, int cur_pitch,
                                                       const uint8_t *__restrict__ ref00, int pitch, long plane_stride, int W, int H)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const jmhip_me_job *job = jobs + blockIdx.x;
  const int R = job->search_range;
  const int wpitch = (2 * R + 16 + 4 + 3) & ~3;
  uint8_t *s_win = smem;
  uint32_t *s_cur = (uint32_t *)(smem + (((2 * R + 16) * wpitch + 15) & ~15));
  JobLds *s_job = (JobLds *)(s_cur + 64);
  stage_job(job, cur, cur_pitch, ref00, pitch, plane_stride, W, H, s_win, wpitch, s_cur, s_job);
  __syncthreads();
  const int n1 = 2 * R + 1, npos = n1 * n1;
  uint16_t *tab = tables + (long)blockIdx.x * table_stride;
  // partition p of the ABI order -> (blocktype, 4x4 raster index) of JM's BlockSAD[list][ref][type][index]
  for (int pos = threadIdx.x; pos < npos; pos += 256) {
    const int wy = pos / n1, wx = pos - wy * n1;
    uint32_t s7[16], sp[NP];
    sad16(s_win, wpitch, s_cur, wx, wy, s7);
    aggregate41(s7, sp);
    const int idx = spiral_index(wx - R, wy - R);
#define PUT(type, k, v) tab[((long)((type) - 1) * 16 + (k)) * npos + idx] = (uint16_t)(v)
    PUT(1, 0, sp[0]);
    PUT(2, 0, sp[1]); PUT(2, 8, sp[2]);
    PUT(3, 0, sp[3]); PUT(3, 2, sp[4]);
    PUT(4, 0, sp[5]); PUT(4, 2, sp[6]); PUT(4, 8, sp[7]); PUT(4, 10, sp[8]);
#pragma unroll
    for (int by = 0; by < 4; by++) { PUT(5, by * 4, sp[9 + by * 2]); PUT(5, by * 4 + 2, sp[9 + by * 2 + 1]); }
#pragma unroll
    for (int bx = 0; bx < 4; bx++) { PUT(6, bx, sp[17 + bx]); PUT(6, 8 + bx, sp[21 + bx]); }
#pragma unroll
    for (int k = 0; k < 16; k++) PUT(7, k, s7[k]);
#undef PUT
  }
}

static size_t me_lds_bytes(int R)
{
  int wpitch = (2 * R + 16 + 4 + 3) & ~3;
  size_t win = (((size_t)(2 * R + 16) * wpitch + 15) & ~(size_t)15);
  return win + 64 * 4 + sizeof(JobLds) + 4 * NP * 8 + 64;
}

// ---- device-resident job records are checked on the device (the `_dev` entry points never see them on the host): a bad record raises the
// context's job error word -- 1 + the index of the first bad job -- and every ME kernel returns at once while the word is set, so a bad job list
// costs an error at the next jmhip_synchronize instead of an out-of-range LDS / plane access.  The word stays set until it has been reported.
__global__ __launch_bounds__(256) void k_check_me_jobs(const jmhip_me_job *__restrict__ jobs, int n, int R_cfg, int W, int H, unsigned *__restrict__ err)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const jmhip_me_job *j = jobs + i;
  const bool bad = j->search_range < 1 || j->search_range > R_cfg || j->mb_x < 0 || j->mb_y < 0 || j->mb_x + 16 > W || j->mb_y + 16 > H || (j->mb_x & 3);
  if (bad) atomicMin(err + 1, (unsigned)i), atomicOr(err, 1u);
}
__global__ __launch_bounds__(256) void k_check_subpel_jobs(const jmhip_subpel_job *__restrict__ jobs, int n, int W, int H, unsigned *__restrict__ err)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const jmhip_subpel_job *j = jobs + i;
  const bool bad = j->pos_x < 0 || j->pos_y < 0 || j->bsx < 4 || j->bsy < 4 || j->bsx > 16 || j->bsy > 16 || (j->bsx & 3) || (j->bsy & 3) || j->pos_x + j->bsx > W || j->pos_y + j->bsy > H;
  if (bad) atomicMin(err + 1, (unsigned)i), atomicOr(err, 1u);
}
void jmhip_launch_check_me_jobs(jmhip_ctx *ctx, const jmhip_me_job *d_jobs, int n)
{
  hipLaunchKernelGGL(k_check_me_jobs, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_jobs, n, ctx->cfg.search_range, ctx->W, ctx->H, ctx->d_me_declined + 4);
  ctx->me_jobs_checked = 1;
}
void jmhip_launch_check_subpel_jobs(jmhip_ctx *ctx, const jmhip_subpel_job *d_jobs, int n)
{
  hipLaunchKernelGGL(k_check_subpel_jobs, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_jobs, n, ctx->W, ctx->H, ctx->d_me_declined + 4);
  ctx->me_jobs_checked = 1;
}
int jmhip_check_job_error(jmhip_ctx *ctx)
{
  if (!ctx->me_jobs_checked) return JMHIP_OK;
  unsigned w[2] = {0, 0};
  HIPCHK(ctx, hipMemcpyAsync(w, ctx->d_me_declined + 4, sizeof w, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->me_jobs_checked = 0;
  if (w[0]) {
    const unsigned clear[2] = {0, 0xffffffffu};
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_me_declined + 4, clear, sizeof clear, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return jmhip_fail(ctx, JMHIP_EINVAL, "device-resident job %u is invalid (search range beyond the context's, or a block outside the picture): the motion estimation launches since then did nothing", w[1]);
  }
  return JMHIP_OK;
}

static int check_jobs_host(jmhip_ctx *ctx, const jmhip_me_job *jobs, int n)
{
  for (int i = 0; i < n; i++) {
    const jmhip_me_job *j = &jobs[i];
    if (j->search_range < 1 || j->search_range > ctx->cfg.search_range)
      return jmhip_fail(ctx, JMHIP_EINVAL, "job %d: search_range %d outside 1..%d", i, j->search_range, ctx->cfg.search_range);
    if (j->mb_x < 0 || j->mb_y < 0 || j->mb_x + 16 > ctx->W || j->mb_y + 16 > ctx->H || (j->mb_x & 3))
      return jmhip_fail(ctx, JMHIP_EINVAL, "job %d: macroblock (%d,%d) outside the %dx%d picture", i, j->mb_x, j->mb_y, ctx->W, ctx->H);
  }
  return JMHIP_OK;
}

extern "C" int jmhip_me_fullsearch_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *d_jobs, int32_t njobs, jmhip_me_result *d_results)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!d_jobs || !d_results || njobs < 0 || slot < 0 || slot >= ctx->cfg.num_ref_slots) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_fullsearch_dev: bad argument");
  if (njobs == 0) return JMHIP_OK;
  const int use_fast = !ctx->force_generic;
  jmhip_launch_check_me_jobs(ctx, d_jobs, njobs);
  jmhip_time_begin(ctx, 1);
  unsigned *dec = ctx->d_me_declined + (ctx->me_launches & 1), *dec_next = ctx->d_me_declined + ((ctx->me_launches + 1) & 1);
  ctx->me_launches++;
  if (use_fast) jmhip_launch_me_fast(ctx, slot, d_jobs, njobs, d_results, dec);
  hipLaunchKernelGGL(k_me_fullsearch, dim3(use_fast ? (njobs < 1024 ? njobs : 1024) : njobs), dim3(256), me_lds_bytes(ctx->cfg.search_range), ctx->stream,
                     d_jobs, d_results, ctx->d_cur, ctx->cur_pitch, ctx->d_sub[slot], ctx->pitch, (long)ctx->plane_stride, ctx->W, ctx->H, ctx->d_spiral, use_fast,
                     dec, dec_next, njobs, ctx->d_me_declined + 4);
  jmhip_time_end(ctx, 1);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_me_fullsearch(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *jobs, int32_t njobs, jmhip_me_result *results)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!jobs || !results || njobs < 0) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_fullsearch: bad argument");
  if (njobs == 0) return JMHIP_OK;
  int r = check_jobs_host(ctx, jobs, njobs);
  if (r) return r;
  void *dj, *dr;
  if ((r = jmhip_scratch(ctx, 0, sizeof(jmhip_me_job) * (size_t)njobs, &dj))) return r;
  if ((r = jmhip_scratch(ctx, 1, sizeof(jmhip_me_result) * (size_t)njobs, &dr))) return r;
  HIPCHK(ctx, hipMemcpyAsync(dj, jobs, sizeof(jmhip_me_job) * (size_t)njobs, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dr, results, sizeof(jmhip_me_result) * (size_t)njobs, hipMemcpyHostToDevice, ctx->stream));   // keep unmasked entries
  if ((r = jmhip_me_fullsearch_dev(ctx, slot, (const jmhip_me_job *)dj, njobs, (jmhip_me_result *)dr))) return r;
  HIPCHK(ctx, hipMemcpyAsync(results, dr, sizeof(jmhip_me_result) * (size_t)njobs, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_me_sad_tables(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *jobs, int32_t njobs, uint16_t *tables)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!jobs || !tables || njobs < 0 || slot < 0 || slot >= ctx->cfg.num_ref_slots) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_sad_tables: bad argument");
  if (njobs == 0) return JMHIP_OK;
  int r = check_jobs_host(ctx, jobs, njobs);
  if (r) return r;
  // tables are laid out back to back, each sized for its own job's search range; the launch uses a common
  // stride so all jobs of one call must share a search range
  const int R = jobs[0].search_range;
  for (int i = 1; i < njobs; i++) if (jobs[i].search_range != R) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_me_sad_tables: mixed search ranges in one call");
  const long stride = 7L * 16 * (2 * R + 1) * (2 * R + 1);
  void *dj, *dt;
  if ((r = jmhip_scratch(ctx, 0, sizeof(jmhip_me_job) * (size_t)njobs, &dj))) return r;
  if ((r = jmhip_scratch(ctx, 1, sizeof(uint16_t) * (size_t)stride * njobs, &dt))) return r;
  HIPCHK(ctx, hipMemcpyAsync(dj, jobs, sizeof(jmhip_me_job) * (size_t)njobs, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(dt, 0, sizeof(uint16_t) * (size_t)stride * njobs, ctx->stream));   // entries JM never fills stay 0
  hipLaunchKernelGGL(k_me_sad_tables, dim3(njobs), dim3(256), me_lds_bytes(ctx->cfg.search_range), ctx->stream,
                     (const jmhip_me_job *)dj, (uint16_t *)dt, stride, ctx->d_cur, ctx->cur_pitch, ctx->d_sub[slot], ctx->pitch, (long)ctx->plane_stride, ctx->W, ctx->H);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(tables, dt, sizeof(uint16_t) * (size_t)stride * njobs, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return JMHIP_OK;
}
