// deblock.hip -- K9+K10: in-loop deblocking of a frame (gfx950).
//
// Device counterpart of (reference, lencod/src):
//   DeblockFrame / DeblockMb            loopFilter.c:63-71 / :120-297   edge order, skip rules
//   GetStrengthVer / GetStrengthHor     loop_filter_normal.c:52-168 / :177-292
//   EdgeLoopLumaVer / EdgeLoopLumaHor   loop_filter_normal.c:301-440 / :444-581
//   EdgeLoopChromaVer / ChromaHor       loop_filter_normal.c:590-672 / :677-757
//   ALPHA_TABLE / BETA_TABLE / CLIP_TAB / chroma_edge / pelnum_cr   lencod/inc/loop_filter.h:32-59
//
// JM filters macroblocks in raster order, in place: a macroblock's left/top edge reads samples the
// left/top neighbours have already finished filtering.  Exactly that order is kept by running the
// macroblocks of one 2:1 anti-diagonal (x + 2y = const) together -- they touch disjoint samples and
// all their predecessors lie on earlier diagonals (it is the schedule of JM's own JM_PARALLEL_DEBLOCK
// variant, loopFilter.c:92-110) -- one launch per diagonal, kernel boundaries providing the ordering.
//
// One workgroup = one wave = one macroblock.  The macroblock plus the 4 (luma) / 2 (chroma) sample
// columns and rows of its left / top neighbours are staged in LDS, the 8 edges are filtered there in
// JM's order (lane = sample row for vertical edges, sample column for horizontal ones), and only
// the samples an edge can modify are written back.  This path is latency-bound by construction
// (W/16 + 2(H/16 - 1) dependent steps per frame), not HBM-bound: DESIGN.md, section "deblock".
#include "deblock_common.h"

struct DbArgs {
  uint8_t *Y, *U, *V; int pitchY, pitchC;
  const jmhip_db_mb *mbs; const jmhip_db_motion *motion;
  int mb_w, mb_h, yuv_format, direct8x8;
  int wave_index;       // x + 2y of this launch
  int y_first;          // first macroblock row on the diagonal
};

__global__ __launch_bounds__(64) void k_deblock_diag(DbArgs A)
{
  __shared__ uint8_t s_y[20 * LP];
  __shared__ uint8_t s_c[2][18 * CP];
  __shared__ uint8_t s_str[2][4][4];
  const int lane = threadIdx.x;
  const int mby = A.y_first + blockIdx.x, mbx = A.wave_index - 2 * mby;
  const int addr = mby * A.mb_w + mbx;
  const jmhip_db_mb *q = &A.mbs[addr];
  if (q->df_disable_idc == 1) return;
  const int fmt = A.yuv_format, cw = fmt ? 8 : 0, ch = fmt == 2 ? 16 : (fmt == 1 ? 8 : 0);

  // ---- stage: luma rows -4..15 x cols -4..15 (neighbour samples only where the neighbour exists)
  for (int k = lane; k < 20 * 20; k += 64) {
    int r = k / 20 - 4, c = k % 20 - 4;
    int yy = mby * 16 + r, xx = mbx * 16 + c;
    s_y[(r + 4) * LP + c + 4] = (yy >= 0 && xx >= 0) ? A.Y[(long)yy * A.pitchY + xx] : 0;
  }
  if (fmt) {
    for (int k = lane; k < 2 * (ch + 2) * (cw + 2); k += 64) {
      int uv = k / ((ch + 2) * (cw + 2)), kk = k % ((ch + 2) * (cw + 2));
      int r = kk / (cw + 2) - 2, c = kk % (cw + 2) - 2;
      int yy = mby * ch + r, xx = mbx * cw + c;
      const uint8_t *img = uv ? A.V : A.U;
      s_c[uv][(r + 2) * CP + c + 2] = (yy >= 0 && xx >= 0) ? img[(long)yy * A.pitchC + xx] : 0;
    }
  }
  // ---- edge flags (DeblockMb :150-165) and the 32 segment strengths
  int left_ok = mbx != 0, top_ok = mby != 0;
  if (q->df_disable_idc == 2) {
    left_ok = mbx != 0 && A.mbs[addr - 1].slice_nr == q->slice_nr;
    top_ok  = mby != 0 && A.mbs[addr - A.mb_w].slice_nr == q->slice_nr;
  }
  if (lane < 32) {
    int dir = lane >> 4, edge = (lane >> 2) & 3, idx = lane & 3;
    int ok = edge || (dir == 0 ? left_ok : top_ok);
    s_str[dir][edge][idx] = ok ? (uint8_t)strength_of(dir, edge, idx, addr, A.mb_w, A.mbs, A.motion) : 0;
  }
  __syncthreads();

  const int t8 = q->transform8x8, cbp = q->cbp, mbt = q->mb_type, st = q->slice_type;
  for (int dir = 0; dir < 2; dir++) {
    for (int edge = 0; edge < 4; edge++) {
      const int non8x8 = (edge & 1) ? !t8 : 1;
      if (cbp == 0) {                                                        // loopFilter.c:173-184 / :222-233
        const int skip8 = dir == 0 ? (fmt != 3) : (fmt == 1);
        if (!non8x8 && skip8) continue;
        if (edge > 0 && (st == 0 || st == 1)) {
          if ((mbt == 0 && st == 0) || mbt == 1 || mbt == (dir == 0 ? 2 : 3)) continue;
          if ((edge & 1) && (mbt == (dir == 0 ? 3 : 2) || (mbt == 0 && st == 1 && A.direct8x8))) continue;
        }
      }
      if (!(edge || (dir == 0 ? left_ok : top_ok))) continue;
      const uint8_t *S = s_str[dir][edge];
      if (!(S[0] | S[1] | S[2] | S[3])) continue;
      const jmhip_db_mb *p = edge ? q : (dir == 0 ? &A.mbs[addr - 1] : &A.mbs[addr - A.mb_w]);
      if (non8x8 && lane < 16) {
        const int QP = (p->qp + q->qp + 1) >> 1;
        const int iA = clip3(0, 51, QP + q->df_alpha_c0), iB = clip3(0, 51, QP + q->df_beta);
        const int alpha = c_alpha[iA], beta = c_beta[iB];
        if (alpha | beta) {
          const int bS = S[lane >> 2];
          uint8_t *s = dir == 0 ? &s_y[(lane + 4) * LP + edge * 4 + 4] : &s_y[(edge * 4 + 4) * LP + lane + 4];
          filter_luma_line(s, dir == 0 ? 1 : LP, bS, alpha, beta, c_tc0[iA][bS > 3 ? 3 : bS]);
        }
      }
      if (fmt == 1 || fmt == 2) {
        const int ecr = c_chroma_edge[dir][edge][fmt];
        const int pelnum = dir == 0 ? ch : cw;                                // pelnum_cr[dir][fmt]
        if (ecr >= 0 && lane >= 16 && lane < 16 + 2 * pelnum) {
          const int uv = (lane - 16) / pelnum, k = (lane - 16) % pelnum;
          const int QP = (p->qpc[uv] + q->qpc[uv] + 1) >> 1;
          const int iA = clip3(0, 51, QP + q->df_alpha_c0), iB = clip3(0, 51, QP + q->df_beta);
          const int alpha = c_alpha[iA], beta = c_beta[iB];
          if (alpha | beta) {
            const int bS = S[pelnum == 8 ? (k >> 1) : (k >> 2)];
            uint8_t *s = dir == 0 ? &s_c[uv][(k + 2) * CP + ecr + 2] : &s_c[uv][(ecr + 2) * CP + k + 2];
            filter_chroma_line(s, dir == 0 ? 1 : CP, bS, alpha, beta, c_tc0[iA][bS > 3 ? 3 : bS]);
          }
        }
      }
      __syncthreads();       // single wave: orders the LDS writes of this edge before the next edge's reads
    }
  }
  __syncthreads();
  // ---- write back: own macroblock, 3 columns of the left neighbour, 3 rows of the top neighbour
  for (int k = lane; k < 19 * 19; k += 64) {
    int r = k / 19 - 3, c = k % 19 - 3;
    if (r < 0 && c < 0) continue;
    int yy = mby * 16 + r, xx = mbx * 16 + c;
    if (yy < 0 || xx < 0) continue;
    A.Y[(long)yy * A.pitchY + xx] = s_y[(r + 4) * LP + c + 4];
  }
  if (fmt) {
    for (int k = lane; k < 2 * (ch + 1) * (cw + 1); k += 64) {
      int uv = k / ((ch + 1) * (cw + 1)), kk = k % ((ch + 1) * (cw + 1));
      int r = kk / (cw + 1) - 1, c = kk % (cw + 1) - 1;
      if (r < 0 && c < 0) continue;
      int yy = mby * ch + r, xx = mbx * cw + c;
      if (yy < 0 || xx < 0) continue;
      uint8_t *img = uv ? A.V : A.U;
      img[(long)yy * A.pitchC + xx] = s_c[uv][(r + 2) * CP + c + 2];
    }
  }
}

extern "C" int jmhip_deblock_frame_dev(jmhip_ctx *ctx, uint8_t *d_Y, int32_t pitchY, uint8_t *d_U, uint8_t *d_V, int32_t pitchC,
                                       const jmhip_db_mb *d_mbs, const jmhip_db_motion *d_motion, int32_t direct8x8)
{
  if (!ctx) return JMHIP_EINVAL;
  if (!d_Y || !d_mbs || !d_motion || (ctx->cfg.yuv_format && (!d_U || !d_V))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_deblock_frame_dev: bad argument");
  if (ctx->bdb_used && ctx->stream != ctx->bdb_stream) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->bdb_ev, 0));      // the workspace: behind the B pictures in flight that were filtered last
  DbArgs A;
  A.Y = d_Y; A.U = d_U; A.V = d_V; A.pitchY = pitchY; A.pitchC = pitchC; A.mbs = d_mbs; A.motion = d_motion;
  A.mb_w = ctx->W / 16; A.mb_h = ctx->H / 16; A.yuv_format = ctx->cfg.yuv_format; A.direct8x8 = direct8x8;
  // row pipeline (deblock_rows.hip) whenever the caller's planes allow its aligned 16/8-byte accesses
  const bool aligned = ((uintptr_t)d_Y & 15) == 0 && (pitchY & 15) == 0 &&
                       (!ctx->cfg.yuv_format || ((((uintptr_t)d_U | (uintptr_t)d_V) & 7) == 0 && (pitchC & 7) == 0));
  jmhip_time_begin(ctx, 4);
  if (aligned && !ctx->force_db_diag) {
    jmhip_launch_deblock_rows(ctx, d_Y, pitchY, d_U, d_V, pitchC, d_mbs, d_motion, direct8x8);
    ctx->db_launched = 1;
  } else {
    const int nwaves = A.mb_w + 2 * (A.mb_h - 1);
    for (int w = 0; w < nwaves; w++) {
      // rows y with 0 <= w - 2y < mb_w
      int y_lo = w - (A.mb_w - 1); y_lo = y_lo <= 0 ? 0 : (y_lo + 1) / 2;
      int y_hi = w / 2; if (y_hi > A.mb_h - 1) y_hi = A.mb_h - 1;
      if (y_hi < y_lo) continue;
      A.wave_index = w; A.y_first = y_lo;
      hipLaunchKernelGGL(k_deblock_diag, dim3(y_hi - y_lo + 1), dim3(64), 0, ctx->stream, A);
    }
  }
  jmhip_time_end(ctx, 4);
  HIPCHK(ctx, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_deblock_frame(jmhip_ctx *ctx, uint16_t *imgY, int32_t pitchY, uint16_t *imgU, uint16_t *imgV, int32_t pitchC,
                                   const jmhip_db_mb *mbs, const jmhip_db_motion *motion, int32_t direct8x8)
{
  if (!ctx) return JMHIP_EINVAL;
  const int fmt = ctx->cfg.yuv_format;
  if (!imgY || !mbs || !motion || (fmt && (!imgU || !imgV))) return jmhip_fail(ctx, JMHIP_EINVAL, "jmhip_deblock_frame: bad argument");
  const int W = ctx->W, H = ctx->H, cw = ctx->cw, ch = ctx->ch, nmb = (W / 16) * (H / 16);
  const size_t ybytes = (size_t)W * H, cbytes = (size_t)cw * ch;
  const size_t side = sizeof(jmhip_db_mb) * (size_t)nmb + sizeof(jmhip_db_motion) * (size_t)(W / 4) * (H / 4);
  int r; void *dpix, *dside;
  if ((r = jmhip_scratch(ctx, 0, ybytes + 2 * cbytes + 256, &dpix))) return r;
  if ((r = jmhip_scratch(ctx, 1, side + 256, &dside))) return r;
  if (ybytes + 2 * cbytes > ctx->h_stage_bytes) return jmhip_fail(ctx, JMHIP_EINVAL, "frame larger than staging");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  uint8_t *st = ctx->h_stage;
  for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) st[(size_t)y * W + x] = (uint8_t)imgY[(size_t)y * pitchY + x];
  for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) {
    st[ybytes + (size_t)y * cw + x] = (uint8_t)imgU[(size_t)y * pitchC + x];
    st[ybytes + cbytes + (size_t)y * cw + x] = (uint8_t)imgV[(size_t)y * pitchC + x];
  }
  uint8_t *dY = (uint8_t *)dpix, *dU = dY + ybytes, *dV = dU + cbytes;
  jmhip_db_mb *dm = (jmhip_db_mb *)dside;
  jmhip_db_motion *dmo = (jmhip_db_motion *)((uint8_t *)dside + ((sizeof(jmhip_db_mb) * (size_t)nmb + 15) & ~(size_t)15));
  HIPCHK(ctx, hipMemcpyAsync(dpix, st, ybytes + 2 * cbytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dm, mbs, sizeof(jmhip_db_mb) * (size_t)nmb, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dmo, motion, sizeof(jmhip_db_motion) * (size_t)(W / 4) * (H / 4), hipMemcpyHostToDevice, ctx->stream));
  if ((r = jmhip_deblock_frame_dev(ctx, dY, W, fmt ? dU : NULL, fmt ? dV : NULL, cw, dm, dmo, direct8x8))) return r;
  HIPCHK(ctx, hipMemcpyAsync(st, dpix, ybytes + 2 * cbytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if ((r = jmhip_check_deblock_error(ctx))) return r;
  for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) imgY[(size_t)y * pitchY + x] = st[(size_t)y * W + x];
  for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) {
    imgU[(size_t)y * pitchC + x] = st[ybytes + (size_t)y * cw + x];
    imgV[(size_t)y * pitchC + x] = st[ybytes + cbytes + (size_t)y * cw + x];
  }
  return JMHIP_OK;
}
