/*
 * oracle/jmo_deblock.c -- TEST INFRASTRUCTURE (parity oracle, see jmo.h).
 * CPU restatement of JM 19.0's in-loop deblocking filter for frame pictures
 * without MBAFF: lencod/src/loopFilter.c:63-297 (DeblockFrame / DeblockMb),
 * lencod/src/loop_filter_normal.c (GetStrengthVer/Hor :52-292, EdgeLoopLuma*
 * :301-581, EdgeLoopChroma* :590-757), tables lencod/inc/loop_filter.h:32-57.
 * Macroblocks are filtered in raster order, in place; within a macroblock the
 * four vertical edges (luma, then chroma per edge) precede the four horizontal ones.
 */
#include <string.h>
#include "jmo.h"

static inline int iabs_(int x) { return x < 0 ? -x : x; }
static inline int iclip3(int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); }
static inline int clip1(int hi, int x) { return x < 0 ? 0 : (x > hi ? hi : x); }

/* loop_filter.h:32-45 -- the H.264 alpha / beta / tc0 tables (spec Table 8-16/8-17) */
static const uint8_t ALPHA_T[52] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,4,4,5,6,7,8,9,10,12,13,15,17,20,22,25,28,
                                    32,36,40,45,50,56,63,71,80,90,101,113,127,144,162,182,203,226,255,255};
static const uint8_t BETA_T[52]  = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,2,2,3,3,3,3,4,4,4,6,6,7,7,8,8,
                                    9,9,10,10,11,11,12,12,13,13,14,14,15,15,16,16,17,17,18,18};
static const uint8_t TC0_T[52][3] = {      /* CLIP_TAB columns bS = 1,2,3 (column 0 is 0, column 4 = column 3) */
  {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},
  {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,1},{0,0,1},{0,0,1},{0,0,1},{0,1,1},{0,1,1},{1,1,1},{1,1,1},{1,1,1},
  {1,1,1},{1,1,2},{1,1,2},{1,1,2},{1,1,2},{1,2,3},{1,2,3},{2,2,3},{2,2,4},{2,3,4},{2,3,4},{3,3,5},{3,4,6},
  {3,4,6},{4,5,7},{4,5,8},{4,6,9},{5,7,10},{6,8,11},{6,8,13},{7,10,14},{8,11,16},{9,12,18},{10,13,20},
  {11,15,23},{13,17,25}
};
static inline int tc0_of(int indexA, int bS) { return bS == 0 ? 0 : TC0_T[indexA][(bS > 3 ? 3 : bS) - 1]; }

static inline int is_intra_type(int t) { return t == 9 || t == 13 || t == 10 || t == 14; }   /* I4MB I8MB I16MB IPCM */

static inline int cmp_mv(const int16_t a[2], const int16_t b[2], int mvlimit)               /* loop_filter.h:62-65 */
{
  return (iabs_(a[0] - b[0]) >= 4) | (iabs_(a[1] - b[1]) >= mvlimit);
}

/* GetStrengthVer :52-168 / GetStrengthHor :177-292 */
void jmo_deblock_strength(uint8_t str[16], int dir, int edge, int mb_addr, int mb_w,
                          const jmo_db_mb *mbs, const jmo_db_motion *motion)
{
  const jmo_db_mb *q = &mbs[mb_addr], *p;
  int mbx = mb_addr % mb_w, mby = mb_addr / mb_w, bw = mb_w * 4, idx, sv;
  const int mvlimit = 4;                                  /* frame picture, no MBAFF: loopFilter.c:133 */
  if (q->slice_type == 3 || q->slice_type == 4) { memset(str, edge == 0 ? 4 : 3, 16); return; }
  p = edge ? q : (dir == 0 ? &mbs[mb_addr - 1] : &mbs[mb_addr - mb_w]);
  if (is_intra_type(q->mb_type) || is_intra_type(p->mb_type)) { memset(str, edge == 0 ? 4 : 3, 16); return; }
  for (idx = 0; idx < 4; idx++) {
    /* block coordinates (in 4x4 units inside the MB) of Q and P along this edge */
    int qbx, qby, pbx, pby, blkQ, blkP;
    if (dir == 0) { qbx = edge; qby = idx; pbx = (edge + 3) & 3; pby = idx; }
    else          { qbx = idx; qby = edge; pbx = idx; pby = (edge + 3) & 3; }
    blkQ = qby * 4 + qbx; blkP = pby * 4 + pbx;
    if (((q->cbp_blk >> blkQ) & 1) || ((p->cbp_blk >> blkP) & 1)) sv = 2;
    else if (edge && (q->mb_type == 1 || q->mb_type == (dir == 0 ? 2 : 3))) sv = 0;
    else {
      int qx = mbx * 4 + qbx, qy = mby * 4 + qby;
      int px = dir == 0 ? qx - 1 : qx, py = dir == 0 ? qy : qy - 1;
      const jmo_db_motion *a = &motion[qy * bw + qx], *b = &motion[py * bw + px];
      int a0 = a->ref_id[0], a1 = a->ref_id[1], b0 = b->ref_id[0], b1 = b->ref_id[1];
      if ((a0 == b0 && a1 == b1) || (a0 == b1 && a1 == b0)) {
        if (a0 != a1) {
          if (a0 == b0) sv = cmp_mv(a->mv[0], b->mv[0], mvlimit) | cmp_mv(a->mv[1], b->mv[1], mvlimit);
          else          sv = cmp_mv(a->mv[0], b->mv[1], mvlimit) | cmp_mv(a->mv[1], b->mv[0], mvlimit);
        } else {
          sv = (cmp_mv(a->mv[0], b->mv[0], mvlimit) | cmp_mv(a->mv[1], b->mv[1], mvlimit)) &&
               (cmp_mv(a->mv[0], b->mv[1], mvlimit) | cmp_mv(a->mv[1], b->mv[0], mvlimit));
        }
      } else sv = 1;
    }
    memset(str + 4 * idx, sv, 4);
  }
}

/* one line of samples across an edge: p[-k*st] ... | q[+k*st]; luma rules :333-432 */
static void luma_line(jmo_pel *q0p, int st, int bS, int alpha, int beta, int c0, int maxv)
{
  int L0 = q0p[-st], R0 = q0p[0];
  if (bS == 4) {
    if (iabs_(R0 - L0) < alpha) {
      int R1 = q0p[st], L1 = q0p[-2 * st];
      if (iabs_(R0 - R1) < beta && iabs_(L0 - L1) < beta) {
        int R2 = q0p[2 * st], L2 = q0p[-3 * st], RL0 = L0 + R0;
        int small_gap = iabs_(R0 - L0) < ((alpha >> 2) + 2);
        int aq = (iabs_(R0 - R2) < beta) & small_gap, ap = (iabs_(L0 - L2) < beta) & small_gap;
        if (ap) {
          int L3 = q0p[-4 * st];
          q0p[-st]     = (jmo_pel)((R1 + ((L1 + RL0) << 1) + L2 + 4) >> 3);
          q0p[-2 * st] = (jmo_pel)((L2 + L1 + RL0 + 2) >> 2);
          q0p[-3 * st] = (jmo_pel)((((L3 + L2) << 1) + L2 + L1 + RL0 + 4) >> 3);
        } else q0p[-st] = (jmo_pel)(((L1 << 1) + L0 + R1 + 2) >> 2);
        if (aq) {
          int R3 = q0p[3 * st];
          q0p[0]      = (jmo_pel)((L1 + ((R1 + RL0) << 1) + R2 + 4) >> 3);
          q0p[st]     = (jmo_pel)((R2 + R0 + L0 + R1 + 2) >> 2);
          q0p[2 * st] = (jmo_pel)((((R3 + R2) << 1) + R2 + R1 + RL0 + 4) >> 3);
        } else q0p[0] = (jmo_pel)(((R1 << 1) + R0 + L1 + 2) >> 2);
      }
    }
  } else if (bS != 0) {
    int diff = R0 - L0;
    if (iabs_(diff) < alpha) {
      int R1 = q0p[st], L1 = q0p[-2 * st];
      if (iabs_(R0 - R1) < beta && iabs_(L0 - L1) < beta) {
        int RL0 = (L0 + R0 + 1) >> 1, R2 = q0p[2 * st], L2 = q0p[-3 * st];
        int aq = iabs_(R0 - R2) < beta, ap = iabs_(L0 - L2) < beta;
        int tc = c0 + ap + aq;
        int dif = iclip3(-tc, tc, ((diff << 2) + (L1 - R1) + 4) >> 3);
        if (ap) q0p[-2 * st] = (jmo_pel)(L1 + iclip3(-c0, c0, (L2 + RL0 - (L1 << 1)) >> 1));
        if (dif != 0) { q0p[-st] = (jmo_pel)clip1(maxv, L0 + dif); q0p[0] = (jmo_pel)clip1(maxv, R0 - dif); }
        if (aq) q0p[st] = (jmo_pel)(R1 + iclip3(-c0, c0, (R2 + RL0 - (R1 << 1)) >> 1));
      }
    }
  }
}

/* chroma rules :632-662 */
static void chroma_line(jmo_pel *q0p, int st, int bS, int alpha, int beta, int c0, int maxv)
{
  int L0 = q0p[-st], R0 = q0p[0], diff = R0 - L0;
  if (bS == 0 || iabs_(diff) >= alpha) return;
  {
    int R1 = q0p[st], L1 = q0p[-2 * st];
    if (iabs_(R0 - R1) >= beta || iabs_(L0 - L1) >= beta) return;
    if (bS == 4) {
      q0p[-st] = (jmo_pel)(((L1 << 1) + L0 + R1 + 2) >> 2);
      q0p[0]   = (jmo_pel)(((R1 << 1) + R0 + L1 + 2) >> 2);
    } else {
      int tc = c0 + 1, dif = iclip3(-tc, tc, ((diff << 2) + (L1 - R1) + 4) >> 3);
      if (dif != 0) { q0p[-st] = (jmo_pel)clip1(maxv, L0 + dif); q0p[0] = (jmo_pel)clip1(maxv, R0 - dif); }
    }
  }
}

/* chroma_edge[dir][edge][yuv_format] and pelnum_cr, loop_filter.h:47-59 */
static const int8_t CHROMA_EDGE[2][4][4] = {
  {{-4, 0, 0, 0}, {-4, -4, -4, 4}, {-4, 4, 4, 8}, {-4, -4, -4, 12}},
  {{-4, 0, 0, 0}, {-4, -4, 4, 4},  {-4, 4, 8, 8}, {-4, -4, 12, 12}}
};
static const int PELNUM_CR[2][4] = {{0, 8, 16, 16}, {0, 8, 8, 16}};

void jmo_deblock_frame(jmo_pel *imgY, int pitchY, jmo_pel *imgU, jmo_pel *imgV, int pitchC,
                       int width, int height, int yuv_format, const jmo_db_mb *mbs,
                       const jmo_db_motion *motion, int max_pel_y, int max_pel_c, int direct8x8)
{
  int mb_w = width / 16, mb_h = height / 16, addr;
  int cw = yuv_format ? 8 : 0, ch = yuv_format == 2 ? 16 : (yuv_format == 1 ? 8 : 0);    /* mb_cr_size */
  for (addr = 0; addr < mb_w * mb_h; addr++) {
    const jmo_db_mb *q = &mbs[addr];
    int mbx = addr % mb_w, mby = addr / mb_w, dir, edge;
    int left_ok, top_ok, non8x8[4] = {1, 1, 1, 1};
    uint8_t str[16];
    if (q->df_disable_idc == 1) continue;                                     /* loopFilter.c:138-142 */
    non8x8[1] = non8x8[3] = !q->transform8x8;                                  /* :150-151 */
    left_ok = mbx != 0; top_ok = mby != 0;
    if (q->df_disable_idc == 2) {                                              /* :159-165 : not across slice edges */
      left_ok = mbx != 0 && mbs[addr - 1].slice_nr == q->slice_nr;
      top_ok  = mby != 0 && mbs[addr - mb_w].slice_nr == q->slice_nr;
    }
    for (dir = 0; dir < 2; dir++) {
      for (edge = 0; edge < 4; edge++) {
        if (q->cbp == 0) {                                                     /* :173-184 / :222-233 */
          int skip8 = dir == 0 ? (yuv_format != 3) : (yuv_format == 1);
          if (non8x8[edge] == 0 && skip8) continue;
          else if (edge > 0 && (q->slice_type == 0 || q->slice_type == 1)) {
            if ((q->mb_type == 0 && q->slice_type == 0) || q->mb_type == 1 || q->mb_type == (dir == 0 ? 2 : 3)) continue;
            else if ((edge & 1) && (q->mb_type == (dir == 0 ? 3 : 2) ||
                     (q->mb_type == 0 && q->slice_type == 1 && direct8x8))) continue;
          }
        }
        if (!(edge || (dir == 0 ? left_ok : top_ok))) continue;
        jmo_deblock_strength(str, dir, edge, addr, mb_w, mbs, motion);
        { int k, any = 0; for (k = 0; k < 16; k++) any |= str[k]; if (!any) continue; }
        {
          const jmo_db_mb *p = edge ? q : (dir == 0 ? &mbs[addr - 1] : &mbs[addr - mb_w]);
          int k;
          if (non8x8[edge]) {                                                  /* EdgeLoopLumaVer/Hor */
            int QP = (p->qp + q->qp + 1) >> 1;
            int iA = iclip3(0, 51, QP + q->df_alpha_c0), iB = iclip3(0, 51, QP + q->df_beta);
            int alpha = ALPHA_T[iA], beta = BETA_T[iB];
            if ((alpha | beta) != 0) {
              for (k = 0; k < 16; k++) {
                jmo_pel *s = dir == 0 ? imgY + (long)(mby * 16 + k) * pitchY + mbx * 16 + edge * 4
                                      : imgY + (long)(mby * 16 + edge * 4) * pitchY + mbx * 16 + k;
                luma_line(s, dir == 0 ? 1 : pitchY, str[k], alpha, beta, tc0_of(iA, str[k]), max_pel_y);
              }
            }
          }
          if (yuv_format == 1 || yuv_format == 2) {                            /* EdgeLoopChromaVer/Hor */
            int ecr = CHROMA_EDGE[dir][edge][yuv_format];
            if (imgU != 0 && ecr >= 0) {
              int uv, pelnum = PELNUM_CR[dir][yuv_format];
              for (uv = 0; uv < 2; uv++) {
                jmo_pel *img = uv ? imgV : imgU;
                int QP = (p->qpc[uv] + q->qpc[uv] + 1) >> 1;
                int iA = iclip3(0, 51, QP + q->df_alpha_c0), iB = iclip3(0, 51, QP + q->df_beta);
                int alpha = ALPHA_T[iA], beta = BETA_T[iB];
                if ((alpha | beta) == 0) continue;
                for (k = 0; k < pelnum; k++) {
                  int bS = str[pelnum == 8 ? (((k >> 1) << 2) + (k & 1)) : k];
                  jmo_pel *s = dir == 0 ? img + (long)(mby * ch + k) * pitchC + mbx * cw + ecr
                                        : img + (long)(mby * ch + ecr) * pitchC + mbx * cw + k;
                  chroma_line(s, dir == 0 ? 1 : pitchC, bS, alpha, beta, tc0_of(iA, bS), max_pel_c);
                }
              }
            }
          }
        }
      }
    }
  }
}
