/*
 * oracle/jmo_tq.c -- TEST INFRASTRUCTURE (parity oracle, see jmo.h).
 * CPU restatement of JM 19.0 integer transforms and scalar quantisation:
 *   lcommon/src/transform.c, lencod/src/quant4x4_normal.c, quant4x4_around.c,
 *   quant8x8_normal.c, q_matrix.c (flat matrices), q_offsets.c (default offsets),
 *   lencod/src/block.c:661-725, lcommon/src/blk_prediction.c:48-62.
 */
#include <string.h>
#include "jmo.h"

static inline int iabs_(int x) { return x < 0 ? -x : x; }
static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int rshift_rnd_sf(int x, int a) { return (x + (1 << (a - 1))) >> a; }   /* ifunctions.h:176 */
static inline int clip1(int hi, int x) { return x < 0 ? 0 : (x > hi ? hi : x); }       /* ifunctions.h:191 */

/* ---------------- 4-point stages, applied along rows then columns ---------------- */
static inline void fwd4(const int *s, int ss, int *d, int ds)         /* transform.c:27-46 / :49-66 */
{
  int e0 = s[0] + s[3 * ss], e1 = s[ss] + s[2 * ss], o0 = s[ss] - s[2 * ss], o1 = s[0] - s[3 * ss];
  d[0]      = e0 + e1;
  d[ds]     = (o1 << 1) + o0;
  d[2 * ds] = e0 - e1;
  d[3 * ds] = o1 - (o0 << 1);
}
static inline void inv4(const int *s, int ss, int *d, int ds)         /* transform.c:79-96 / :99-117 */
{
  int e0 = s[0] + s[2 * ss], e1 = s[0] - s[2 * ss];
  int o0 = (s[ss] >> 1) - s[3 * ss], o1 = s[ss] + (s[3 * ss] >> 1);
  d[0] = e0 + o1; d[ds] = e1 + o0; d[2 * ds] = e1 - o0; d[3 * ds] = e0 - o1;
}
void jmo_forward4x4(const int in[16], int out[16])
{
  int t[16], i;
  for (i = 0; i < 4; i++) fwd4(in + 4 * i, 1, t + 4 * i, 1);
  for (i = 0; i < 4; i++) fwd4(t + i, 4, out + i, 4);
}
void jmo_inverse4x4(const int in[16], int out[16])
{
  int t[16], i;
  for (i = 0; i < 4; i++) inv4(in + 4 * i, 1, t + 4 * i, 1);
  for (i = 0; i < 4; i++) inv4(t + i, 4, out + i, 4);
}
/* hadamard4x4 transform.c:121-168 : rows (t0+t1, t3+t2, t0-t1, t3-t2), columns the same then >>1 */
void jmo_hadamard4x4(const int in[16], int out[16])
{
  int t[16], i;
  for (i = 0; i < 4; i++) {
    const int *s = in + 4 * i;
    int e0 = s[0] + s[3], e1 = s[1] + s[2], o0 = s[1] - s[2], o1 = s[0] - s[3];
    t[4*i] = e0 + e1; t[4*i+1] = o1 + o0; t[4*i+2] = e0 - e1; t[4*i+3] = o1 - o0;
  }
  for (i = 0; i < 4; i++) {
    int e0 = t[i] + t[12+i], e1 = t[4+i] + t[8+i], o0 = t[4+i] - t[8+i], o1 = t[i] - t[12+i];
    out[i] = (e0 + e1) >> 1; out[4+i] = (o0 + o1) >> 1; out[8+i] = (e0 - e1) >> 1; out[12+i] = (o1 - o0) >> 1;
  }
}
/* ihadamard4x4 transform.c:170-220 */
void jmo_ihadamard4x4(const int in[16], int out[16])
{
  int t[16], i;
  for (i = 0; i < 4; i++) {
    const int *s = in + 4 * i;
    int e0 = s[0] + s[2], e1 = s[0] - s[2], o0 = s[1] - s[3], o1 = s[1] + s[3];
    t[4*i] = e0 + o1; t[4*i+1] = e1 + o0; t[4*i+2] = e1 - o0; t[4*i+3] = e0 - o1;
  }
  for (i = 0; i < 4; i++) {
    int e0 = t[i] + t[8+i], e1 = t[i] - t[8+i], o0 = t[4+i] - t[12+i], o1 = t[4+i] + t[12+i];
    out[i] = e0 + o1; out[4+i] = e1 + o0; out[8+i] = e1 - o0; out[12+i] = e0 - o1;
  }
}
/* hadamard2x2 transform.c:284-297: in = {dc00, dc01(x+4), dc10(y+4), dc11} */
void jmo_hadamard2x2(const int in[4], int out[4])
{
  int p0 = in[0] + in[1], p1 = in[0] - in[1], p2 = in[2] + in[3], p3 = in[2] - in[3];
  out[0] = p0 + p2; out[1] = p1 + p3; out[2] = p0 - p2; out[3] = p1 - p3;
}
void jmo_ihadamard2x2(const int in[4], int out[4])                   /* transform.c:299-312 */
{
  int t0 = in[0] + in[1], t1 = in[0] - in[1], t2 = in[2] + in[3], t3 = in[2] - in[3];
  out[0] = t0 + t2; out[1] = t1 + t3; out[2] = t0 - t2; out[3] = t1 - t3;
}

/* hadamard4x2 transform.c:220-256: in = tblock rows [2][4] -> out [2][4] (4:2:2 chroma DC) */
void jmo_hadamard4x2(const int in[8], int out[8])
{
  int t[8], i;
  for (i = 0; i < 4; i++) { t[i] = in[i] + in[4 + i]; t[4 + i] = in[i] - in[4 + i]; }
  for (i = 0; i < 2; i++) {
    const int *p = t + 4 * i;
    int t0 = p[0] + p[3], t1 = p[1] + p[2], t2 = p[1] - p[2], t3 = p[0] - p[3];
    out[4*i] = t0 + t1; out[4*i+1] = t3 + t2; out[4*i+2] = t0 - t1; out[4*i+3] = t3 - t2;
  }
}
/* ihadamard4x2 transform.c:258-298: in [2][4] -> out TRANSPOSED [4][2] (block[0..3][i], i = 0,1) */
void jmo_ihadamard4x2(const int in[8], int out[8])
{
  int t[8], i;
  for (i = 0; i < 4; i++) { t[i] = in[i] + in[4 + i]; t[4 + i] = in[i] - in[4 + i]; }
  for (i = 0; i < 2; i++) {
    const int *p = t + 4 * i;
    int t0 = p[0] + p[2], t1 = p[0] - p[2], t2 = p[1] - p[3], t3 = p[1] + p[3];
    out[0*2+i] = t0 + t3; out[1*2+i] = t1 + t2; out[2*2+i] = t1 - t2; out[3*2+i] = t0 - t3;
  }
}

/* ---------------- 8-point stages ---------------- */
static inline void fwd8(const int *s, int ss, int *d, int ds)         /* transform.c:364-401 */
{
  int p[8], a0, a1, a2, a3, b0, b1, b2, b3, b4, b5, b6, b7, k;
  for (k = 0; k < 8; k++) p[k] = s[k * ss];
  a0 = p[0] + p[7]; a1 = p[1] + p[6]; a2 = p[2] + p[5]; a3 = p[3] + p[4];
  b0 = a0 + a3; b1 = a1 + a2; b2 = a0 - a3; b3 = a1 - a2;
  a0 = p[0] - p[7]; a1 = p[1] - p[6]; a2 = p[2] - p[5]; a3 = p[3] - p[4];
  b4 = a1 + a2 + ((a0 >> 1) + a0);
  b5 = a0 - a3 - ((a2 >> 1) + a2);
  b6 = a0 + a3 - ((a1 >> 1) + a1);
  b7 = a1 - a2 + ((a3 >> 1) + a3);
  d[0]      = b0 + b1;
  d[ds]     = b4 + (b7 >> 2);
  d[2 * ds] = b2 + (b3 >> 1);
  d[3 * ds] = b5 + (b6 >> 2);
  d[4 * ds] = b0 - b1;
  d[5 * ds] = b6 - (b5 >> 2);
  d[6 * ds] = (b2 >> 1) - b3;
  d[7 * ds] = (b4 >> 2) - b7;
}
static inline void inv8(const int *s, int ss, int *d, int ds)         /* transform.c:461-498 */
{
  int p[8], a0, a1, a2, a3, b0, b1, b2, b3, b4, b5, b6, b7, k;
  for (k = 0; k < 8; k++) p[k] = s[k * ss];
  a0 = p[0] + p[4]; a1 = p[0] - p[4]; a2 = p[6] - (p[2] >> 1); a3 = p[2] + (p[6] >> 1);
  b0 = a0 + a3; b2 = a1 - a2; b4 = a1 + a2; b6 = a0 - a3;
  a0 = -p[3] + p[5] - p[7] - (p[7] >> 1);
  a1 =  p[1] + p[7] - p[3] - (p[3] >> 1);
  a2 = -p[1] + p[7] + p[5] + (p[5] >> 1);
  a3 =  p[3] + p[5] + p[1] + (p[1] >> 1);
  b1 = a0 + (a3 >> 2); b3 = a1 + (a2 >> 2); b5 = a2 - (a1 >> 2); b7 = a3 - (a0 >> 2);
  d[0] = b0 + b7; d[ds] = b2 - b5; d[2 * ds] = b4 + b3; d[3 * ds] = b6 + b1;
  d[4 * ds] = b6 - b1; d[5 * ds] = b4 - b3; d[6 * ds] = b2 + b5; d[7 * ds] = b0 - b7;
}
void jmo_forward8x8(const int in[64], int out[64])
{
  int t[64], i;
  for (i = 0; i < 8; i++) fwd8(in + 8 * i, 1, t + 8 * i, 1);
  for (i = 0; i < 8; i++) fwd8(t + i, 8, out + i, 8);
}
void jmo_inverse8x8(const int in[64], int out[64])
{
  int t[64], i;
  for (i = 0; i < 8; i++) inv8(in + 8 * i, 1, t + 8 * i, 1);
  for (i = 0; i < 8; i++) inv8(t + i, 8, out + i, 8);
}

/* ---------------- tables ---------------- */
/* zig-zag frame scans, (i = horizontal, j = vertical) pairs: block.c:170-176, transform8x8.c:44-53 */
const uint8_t JMO_SNGL_SCAN[16][2] = {
  {0,0},{1,0},{0,1},{0,2},{1,1},{2,0},{3,0},{2,1},{1,2},{0,3},{1,3},{2,2},{3,1},{3,2},{2,3},{3,3}
};
const uint8_t JMO_SNGL_SCAN8x8[64][2] = {
  {0,0},{1,0},{0,1},{0,2},{1,1},{2,0},{3,0},{2,1},{1,2},{0,3},{0,4},{1,3},{2,2},{3,1},{4,0},{5,0},
  {4,1},{3,2},{2,3},{1,4},{0,5},{0,6},{1,5},{2,4},{3,3},{4,2},{5,1},{6,0},{7,0},{6,1},{5,2},{4,3},
  {3,4},{2,5},{1,6},{0,7},{1,7},{2,6},{3,5},{4,4},{5,3},{6,2},{7,1},{7,2},{6,3},{5,4},{4,5},{3,6},
  {2,7},{3,7},{4,6},{5,5},{6,4},{7,3},{7,4},{6,5},{5,6},{4,7},{5,7},{6,6},{7,5},{7,6},{6,7},{7,7}
};
/* coefficient-cost tables for thresholding: block.c:72-77, transform8x8.c:83-93 */
const uint8_t JMO_COEFF_COST4x4[3][16] = {
  {3,2,2,1,1,1,0,0,0,0,0,0,0,0,0,0},
  {9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9},
  {3,2,2,1,1,1,0,0,0,0,0,0,0,0,0,0}
};
const uint8_t JMO_COEFF_COST8x8[2][64] = {
  {3,3,3,3,2,2,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,0,0,0,0,0,0,0,0,
   0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0},
  {9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,
   9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9}
};

/* Flat-matrix quantiser scales (H.264 normAdjust tables) by position class.
 * 4x4 (q_matrix.c:20-36): class 0 = both coords even, 1 = both odd, 2 = mixed. */
static const int Q4[6][3]  = {{13107,5243,8066},{11916,4660,7490},{10082,4194,6554},{9362,3647,5825},{8192,3355,5243},{7282,2893,4559}};
static const int DQ4[6][3] = {{10,16,13},{11,18,14},{13,20,16},{14,23,18},{16,25,20},{18,29,23}};
static inline int cls4(int j, int i) { return ((i & 1) == 0 && (j & 1) == 0) ? 0 : (((i & 1) && (j & 1)) ? 1 : 2); }
/* 8x8 (q_matrix.c:38-167) */
static const int Q8[6][6]  = {{13107,11428,20972,12222,16777,15481},{11916,10826,19174,11058,14980,14290},
                              {10082,8943,15978,9675,12710,11985},{9362,8228,14913,8931,11984,11259},
                              {8192,7346,13159,7740,10486,9777},{7282,6428,11570,6830,9118,8640}};
static const int DQ8[6][6] = {{20,18,32,19,25,24},{22,19,35,21,28,26},{26,23,42,24,33,31},
                              {28,25,45,26,35,33},{32,28,51,30,40,38},{36,32,58,34,46,43}};
static inline int cls8(int j, int i)
{
  int i4 = i & 3, j4 = j & 3;
  if (i4 == 0 && j4 == 0) return 0;
  if ((i & 1) && (j & 1)) return 1;
  if (i4 == 2 && j4 == 2) return 2;
  if ((i4 == 0 && (j & 1)) || ((i & 1) && j4 == 0)) return 3;
  if ((i4 == 0 && j4 == 2) || (i4 == 2 && j4 == 0)) return 4;
  return 5;
}

/* q_params_4x4[pl][intra][qp] for flat matrices and a uniform offset value:
 * set_default_quant4x4 q_matrix.c:566-577 + update_q_offset4x4 q_offsets.c:238-249
 * (OffsetComp = offset << (Q_BITS + qp_per - OffsetBits), OffsetBits = 11). */
void jmo_qparams_4x4(int qp, int intra, int offset_val, jmo_qparam out[16])
{
  int rem = qp % 6, per = qp / 6, i, j;
  (void)intra;
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) {
    out[j * 4 + i].ScaleComp    = Q4[rem][cls4(j, i)];
    out[j * 4 + i].InvScaleComp = DQ4[rem][cls4(j, i)] << 4;
    out[j * 4 + i].OffsetComp   = offset_val << (15 + per - 11);
  }
}
void jmo_qparams_8x8(int qp, int intra, int offset_val, jmo_qparam out[64])
{
  int rem = qp % 6, per = qp / 6, i, j;
  (void)intra;
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) {
    out[j * 8 + i].ScaleComp    = Q8[rem][cls8(j, i)];
    out[j * 8 + i].InvScaleComp = DQ8[rem][cls8(j, i)] << 4;
    out[j * 8 + i].OffsetComp   = offset_val << (16 + per - 11);    /* q_offsets.c CalculateOffset8x8Param: Q_BITS_8 */
  }
}
void jmo_qparams_4x4_m(int qp, const int16_t off[16], jmo_qparam out[16])
{
  int k;
  jmo_qparams_4x4(qp, 0, 0, out);
  for (k = 0; k < 16; k++) out[k].OffsetComp = (int)off[k] << (15 + qp / 6 - 11);
}
void jmo_qparams_8x8_m(int qp, const int16_t off[64], jmo_qparam out[64])
{
  int k;
  jmo_qparams_8x8(qp, 0, 0, out);
  for (k = 0; k < 64; k++) out[k].OffsetComp = (int)off[k] << (16 + qp / 6 - 11);
}

/* quant_4x4_normal quant4x4_normal.c:39-115 / quant_4x4_around quant4x4_around.c:40-127 */
static int quant4x4_core(int tb[16], const jmo_qparam q[16], int qp_per, int cavlc,
                         const uint8_t *scan, const uint8_t *c_cost, int level[17], int run[17],
                         int *coeff_cost, int around, int arw, int fadjust[16])
{
  int q_bits = 15 + qp_per, k, r = 0, n = 0, nonzero = 0;
  for (k = 0; k < 16; k++) {
    int i = scan[2 * k], j = scan[2 * k + 1], idx = j * 4 + i, c = tb[idx];
    if (c != 0) {
      int scaled = iabs_(c) * q[idx].ScaleComp;
      int lev = (scaled + q[idx].OffsetComp) >> q_bits;
      if (lev != 0) {
        if (cavlc) lev = imin_(lev, 2063);                                   /* CAVLC_LEVEL_LIMIT */
        if (around) fadjust[idx] = rshift_rnd_sf(arw * (scaled - (lev << q_bits)), q_bits + 1);
        *coeff_cost += (lev > 1) ? 999999 : c_cost[r];                       /* MAX_VALUE defines.h:124 */
        lev = c < 0 ? -lev : lev;
        tb[idx] = rshift_rnd_sf((lev * q[idx].InvScaleComp) << qp_per, 4);
        level[n] = lev; run[n] = r; n++;
        r = 0; nonzero = 1;
      } else {
        if (around) fadjust[idx] = 0;
        tb[idx] = 0; r++;
      }
    } else {
      if (around) fadjust[idx] = 0;
      r++;
    }
  }
  level[n] = 0;
  return nonzero;
}
int jmo_quant_4x4_normal(int tb[16], const jmo_qparam q[16], int qp_per, int cavlc, const uint8_t *scan,
                         const uint8_t *c_cost, int level[17], int run[17], int *coeff_cost)
{
  return quant4x4_core(tb, q, qp_per, cavlc, scan, c_cost, level, run, coeff_cost, 0, 0, 0);
}
int jmo_quant_4x4_around(int tb[16], const jmo_qparam q[16], int qp_per, int cavlc, const uint8_t *scan,
                         const uint8_t *c_cost, int arw, int level[17], int run[17], int *coeff_cost, int fadjust[16])
{
  return quant4x4_core(tb, q, qp_per, cavlc, scan, c_cost, level, run, coeff_cost, 1, arw, fadjust);
}

/* quant_8x8_normal quant8x8_normal.c:43-107 (no CAVLC level clamp in this variant) */
int jmo_quant_8x8_normal(int tb[64], const jmo_qparam q[64], int qp_per, int cavlc, const uint8_t *scan,
                         const uint8_t *c_cost, int level[65], int run[65], int *coeff_cost)
{
  int q_bits = 16 + qp_per, k, r = 0, n = 0, nonzero = 0;
  (void)cavlc;
  for (k = 0; k < 64; k++) {
    int i = scan[2 * k], j = scan[2 * k + 1], idx = j * 8 + i, c = tb[idx];
    if (c != 0) {
      int scaled = iabs_(c) * q[idx].ScaleComp;
      int lev = (scaled + q[idx].OffsetComp) >> q_bits;
      if (lev != 0) {
        nonzero = 1;
        *coeff_cost += (lev > 1) ? 999999 : c_cost[r];
        lev = c < 0 ? -lev : lev;
        tb[idx] = rshift_rnd_sf((lev * q[idx].InvScaleComp) << qp_per, 6);
        level[n] = lev; run[n] = r; n++; r = 0;
      } else { r++; tb[idx] = 0; }
    } else r++;
  }
  level[n] = 0;
  return nonzero;
}

/* The four 8x8 quantisers in one restatement:
 *   variant 0 quant_8x8_normal      quant8x8_normal.c:43-107     one 64-entry level/run list, frame zig-zag
 *   variant 1 quant_8x8_around      quant8x8_around.c:43-123     + fadjust
 *   variant 2 quant_8x8cavlc_normal quant8x8_normal.c:123-203    four 16-entry lists (scan quarter k -> list k), |level| <= 2063
 *   variant 3 quant_8x8cavlc_around quant8x8_around.c:136-220    + fadjust
 * scan: 64 (i,j) pairs as handed to the function by JM (SNGL_SCAN8x8 or SNGL_SCAN8x8_CAVLC, transform8x8.c).
 * level / run: 68 entries; variants 0/1 use [0..64], variants 2/3 four lists of 17 at [17k ..]. */
int jmo_quant_8x8(int tb[64], const jmo_qparam q[64], int qp_per, int variant, const uint8_t *scan, const uint8_t *c_cost,
                  int arw, int level[68], int run[68], int *coeff_cost, int fadjust[64])
{
  const int q_bits = 16 + qp_per, cavlc = variant >= 2, around = variant & 1;
  int k, nonzero = 0, r[4] = {0, 0, 0, 0}, n[4] = {0, 0, 0, 0};
  for (k = 0; k < 64; k++) {
    const int i = scan[2 * k], j = scan[2 * k + 1], idx = j * 8 + i, c = tb[idx], l = cavlc ? k >> 4 : 0;
    int fadj = 0;
    if (c != 0) {
      const int scaled = iabs_(c) * q[idx].ScaleComp;
      int lev = (scaled + q[idx].OffsetComp) >> q_bits;
      if (lev != 0) {
        if (cavlc) lev = imin_(lev, 2063);
        if (around) fadj = rshift_rnd_sf(arw * (scaled - (lev << q_bits)), q_bits + 1);
        nonzero = 1;
        *coeff_cost += (lev > 1) ? 999999 : c_cost[r[l]];
        lev = c < 0 ? -lev : lev;
        tb[idx] = rshift_rnd_sf((lev * q[idx].InvScaleComp) << qp_per, 6);
        level[17 * l * cavlc + n[l]] = lev; run[17 * l * cavlc + n[l]] = r[l]; n[l]++; r[l] = 0;
      } else { r[l]++; tb[idx] = 0; }
    } else r[l]++;
    if (around) fadjust[idx] = fadj;
  }
  if (cavlc) for (k = 0; k < 4; k++) level[17 * k + n[k]] = 0;
  else level[n[0]] = 0;
  return nonzero;
}
/* the de-interleaved frame scan the CAVLC variants are given (transform8x8.c SNGL_SCAN8x8_CAVLC): list k takes zig-zag positions
 * k, k+4, k+8, ... ; pinned against the table JM passes (tests/test_oracle_golden.py) */
void jmo_scan8x8_cavlc(uint8_t out[64][2])
{
  int k, i;
  for (k = 0; k < 4; k++) for (i = 0; i < 16; i++) { out[16 * k + i][0] = JMO_SNGL_SCAN8x8[4 * i + k][0]; out[16 * k + i][1] = JMO_SNGL_SCAN8x8[4 * i + k][1]; }
}

/* quant_dc4x4_normal quant4x4_normal.c:200-259: one LevelQuantParams for all 16 DC coefficients, q_bits + 1, the block is left
 * holding the LEVELS (the caller runs ihadamard4x4 and dequantises afterwards) */
int jmo_quant_dc4x4_normal(int tb[16], const jmo_qparam *q, int qp_per, int cavlc, int level[17], int run[17])
{
  const int q_bits = 15 + qp_per + 1;
  int k, r = 0, n = 0, nonzero = 0;
  for (k = 0; k < 16; k++) {
    const int idx = JMO_SNGL_SCAN[k][1] * 4 + JMO_SNGL_SCAN[k][0], c = tb[idx];
    if (c != 0) {
      int lev = (iabs_(c) * q->ScaleComp + (q->OffsetComp << 1)) >> q_bits;
      if (lev != 0) {
        if (cavlc) lev = imin_(lev, 2063);
        lev = c < 0 ? -lev : lev;
        tb[idx] = lev; level[n] = lev; run[n] = r; n++; r = 0; nonzero = 1;
      } else { r++; tb[idx] = 0; }
    } else r++;
  }
  level[n] = 0;
  return nonzero;
}

/* residual_transform_quant_luma_8x8 transform8x8.c:522-586 (cavlc = 0) / residual_transform_quant_luma_8x8_cavlc :604-672
 * (cavlc = 1: no check_zero, de-interleaved scan, four lists), one 8x8 block, frame scan, disthres 0.
 * any_residual (may be NULL): check_zero() of the non-CAVLC path; when it is 0 JM leaves fadjust untouched. */
int jmo_rtq_luma_8x8(const jmo_pel orig[64], const jmo_pel pred[64], const jmo_qparam q[64], int qp_per, int cavlc,
                     int adaptive_rounding, int arw, int max_pel, int level[68], int run[68], int *coeff_cost,
                     jmo_pel rec[64], int fadjust[64], int *any_residual)
{
  int res[64], tb[64], rr[64], k, any = 0, nonzero = 0;
  uint8_t scan_c[64][2];
  for (k = 0; k < 64; k++) { res[k] = (int)orig[k] - (int)pred[k]; any |= res[k]; }
  if (any_residual) *any_residual = any != 0;
  if (cavlc || any) {
    jmo_forward8x8(res, tb);
    jmo_scan8x8_cavlc(scan_c);
    nonzero = jmo_quant_8x8(tb, q, qp_per, 2 * cavlc + adaptive_rounding, cavlc ? &scan_c[0][0] : &JMO_SNGL_SCAN8x8[0][0],
                            JMO_COEFF_COST8x8[0], arw, level, run, coeff_cost, fadjust);
  } else level[0] = 0;
  if (nonzero) {
    jmo_inverse8x8(tb, rr);
    for (k = 0; k < 64; k++) rec[k] = (jmo_pel)clip1(max_pel, rshift_rnd_sf(rr[k], 6) + (int)pred[k]);   /* DQ_BITS_8 = 6 */
  } else {
    for (k = 0; k < 64; k++) rec[k] = pred[k];
  }
  return nonzero;
}

/* residual_transform_quant_luma_4x4 block.c:661-725 for one block, flat matrices,
 * default offsets (q_offsets.c:135-162: 682 intra / 342 inter), frame scan, disthres 0. */
int jmo_rtq_luma_4x4(const jmo_pel orig[16], const jmo_pel pred[16], int qp, int intra, int adaptive_rounding,
                     int arw, int max_pel, int level[17], int run[17], int *coeff_cost,
                     jmo_pel rec[16], int fadjust[16])
{
  int res[16], tb[16], rr[16], k, any = 0, nonzero = 0;
  jmo_qparam q[16];
  for (k = 0; k < 16; k++) { res[k] = (int)orig[k] - (int)pred[k]; any |= res[k]; }
  if (any) {                                                                  /* check_zero block.c:627 */
    jmo_qparams_4x4(qp, intra, intra ? 682 : 342, q);
    jmo_forward4x4(res, tb);
    nonzero = quant4x4_core(tb, q, qp / 6, 1, &JMO_SNGL_SCAN[0][0], JMO_COEFF_COST4x4[0], level, run,
                            coeff_cost, adaptive_rounding, arw, fadjust);
  } else level[0] = 0;
  if (nonzero) {
    jmo_inverse4x4(tb, rr);
    for (k = 0; k < 16; k++) rec[k] = (jmo_pel)clip1(max_pel, rshift_rnd_sf(rr[k], 6) + (int)pred[k]);  /* blk_prediction.c:60 */
  } else {
    for (k = 0; k < 16; k++) rec[k] = pred[k];
  }
  return nonzero;
}

/* ---------------- chroma residual: residual_transform_quant_chroma_4x4 block.c:954-1200 ----------------
 * One chroma plane (uv) of one macroblock, 4:2:0 (8x8 samples) or 4:2:2 (8x16), rows of 8 samples.
 *   forward4x4 per 4x4 block (check_zero blocks stay zero)                         :1013-1029
 *   DC: 4:2:0 hadamard2x2 + quant_dc2x2 (quantChroma_normal.c:37) + ihadamard2x2, >> 5       :1031-1055
 *       4:2:2 hadamard4x2 on the transposed DCs + quant_dc4x2 (:110, qp + 3) + ihadamard4x2, (x + 32) >> 6   :1056-1093
 *   AC: quant_ac4x4_normal / _around (quant4x4_normal.c:117 / quant4x4_around.c:129) per block in (b8, b4) = raster order   :1096-1137
 *   thresholding: all AC levels dropped when their summed cost < _CHROMA_COEFF_COST_ = 4 (defines.h:115)      :1139-1171
 *   inverse4x4 of every block with a DC or surviving AC, sample_reconstruct (DQ_BITS 6) / copy of the prediction   :1173-1199
 * cbp_blk: the macroblock's coded-block bits, updated as JM does; returns cr_cbp.  Frame scan, disthres 0. */
int jmo_rtq_chroma(int yuv, int uv, int cr_cbp, int64_t *cbp_blk, const jmo_qparam q_ac[16], const jmo_qparam *q_dc,
                   int qp_per_ac, int qp_per_dc, int cavlc, int adaptive_rounding, int arw, int max_pel,
                   const jmo_pel *orig, const jmo_pel *pred, jmo_pel *rec, int dc_level[9], int dc_run[9],
                   int ac_level[8][16], int ac_run[8][16], int fadjust[128])
{
  const int H = yuv == 2 ? 16 : 8, nblk = H / 2, uv_scale = uv * (yuv == 2 ? 2 : 1);
  static const uint8_t scan422[8][2] = {{0,0},{0,1},{1,0},{0,2},{0,3},{1,1},{1,2},{1,3}};       /* SCAN_YUV422 block.c:88 (j, i) */
  int rres[16][8], nonzero[8] = {0}, k, j, i, coeff_cost = 0, cr_cbp_tmp = 0, dczero = 0, nonezero = 0, any = 0;
  for (k = 0; k < nblk; k++) {
    const int n1 = 4 * (k & 1), n2 = 4 * (k >> 1);
    int in[16], out[16], z = 0;
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) { in[4*j+i] = (int)orig[(n2+j)*8 + n1+i] - (int)pred[(n2+j)*8 + n1+i]; z |= in[4*j+i]; }
    if (z) jmo_forward4x4(in, out); else memset(out, 0, sizeof out);
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) rres[n2+j][n1+i] = out[4*j+i];
  }
  {                                                                            /* ---- DC ---- */
    const int q_bits = 15 + qp_per_dc + 1, ndc = yuv == 2 ? 8 : 4;
    int m[8], t[8], run = 0, n = 0;
    if (yuv == 1) { int in[4] = {rres[0][0], rres[0][4], rres[4][0], rres[4][4]}; jmo_hadamard2x2(in, m); }
    else {
      int in[8];
      for (j = 0; j < 4; j++) { in[j] = rres[4*j][0]; in[4 + j] = rres[4*j][4]; }   /* tblk4x4[i>>2][j>>2] = mb_rres[j][i]: rows = columns 0 / 4 */
      jmo_hadamard4x2(in, m);
    }
    for (k = 0; k < ndc; k++) {
      const int idx = yuv == 1 ? k : scan422[k][0] * 4 + scan422[k][1], c = m[idx];
      if (c != 0) {
        int lev = (iabs_(c) * q_dc->ScaleComp + (q_dc->OffsetComp << 1)) >> q_bits;
        if (lev != 0) {
          if (cavlc) lev = imin_(lev, 2063);
          lev = c < 0 ? -lev : lev;
          m[idx] = (lev * q_dc->InvScaleComp) << qp_per_dc;
          dc_level[n] = lev; dc_run[n] = run; n++; run = 0; dczero = 1;
        } else { run++; m[idx] = 0; }
      } else run++;
    }
    dc_level[n] = 0;
    /* JM forms these masks in 32-bit int arithmetic (block.c:1044 / :1080): 0xff0000 << 8 is negative and sign-extends into the
       64-bit cbp_blk, setting every bit from 24 up for the V plane of 4:2:2 -- restated as is */
    if (dczero) { *cbp_blk |= (int64_t)(int32_t)(yuv == 1 ? 0xf0000u << (uv << 2) : 0xff0000u << (uv << 3)); if (cr_cbp < 1) cr_cbp = 1; }
    if (yuv == 1) {
      jmo_ihadamard2x2(m, t);
      rres[0][0] = t[0] >> 5; rres[0][4] = t[1] >> 5; rres[4][0] = t[2] >> 5; rres[4][4] = t[3] >> 5;
    } else {
      jmo_ihadamard4x2(m, t);                                                    /* t[j*2 + c]: row group j, column group c */
      for (j = 0; j < 4; j++) { rres[4*j][0] = rshift_rnd_sf(t[2*j], 6); rres[4*j][4] = rshift_rnd_sf(t[2*j + 1], 6); }
    }
  }
  for (k = 0; k < nblk; k++) {                                                 /* ---- AC ---- */
    const int n1 = 4 * (k & 1), n2 = 4 * (k >> 1), q_bits = 15 + qp_per_ac;
    int c, run = 0, n = 0, nz = 0;
    for (c = 1; c < 16; c++) {
      const int ii = JMO_SNGL_SCAN[c][0], jj = JMO_SNGL_SCAN[c][1], qi = jj * 4 + ii, v = rres[n2+jj][n1+ii];
      int fadj = 0;
      if (v != 0) {
        const int scaled = iabs_(v) * q_ac[qi].ScaleComp;
        int lev = (scaled + q_ac[qi].OffsetComp) >> q_bits;
        if (lev != 0) {
          if (cavlc) lev = imin_(lev, 2063);
          if (adaptive_rounding) fadj = rshift_rnd_sf(arw * (scaled - (lev << q_bits)), q_bits + 1);
          coeff_cost += (lev > 1) ? 999999 : JMO_COEFF_COST4x4[0][run];
          lev = v < 0 ? -lev : lev;
          rres[n2+jj][n1+ii] = rshift_rnd_sf((lev * q_ac[qi].InvScaleComp) << qp_per_ac, 4);
          ac_level[k][n] = lev; ac_run[k][n] = run; n++; run = 0; nz = 1;
        } else { rres[n2+jj][n1+ii] = 0; run++; }
      } else run++;
      if (adaptive_rounding) fadjust[(n2+jj)*8 + n1+ii] = fadj;
    }
    ac_level[k][n] = 0;
    nonzero[k] = nz;
    if (nz) { *cbp_blk |= (int64_t)1 << (16 + 4 * uv_scale + k); cr_cbp_tmp = 2; nonezero = 1; }
  }
  if (nonezero && coeff_cost < 4) {                                            /* ---- thresholding ---- */
    const int64_t pat = yuv == 1 ? (int64_t)0xf0000 << (uv << 2) : (int64_t)0xff0000 << (uv << 3);
    cr_cbp_tmp = 0;
    for (k = 0; k < nblk; k++)
      if (nonzero[k]) {
        const int n1 = 4 * (k & 1), n2 = 4 * (k >> 1);
        int c;
        nonzero[k] = 0;
        if (!dczero) *cbp_blk &= ~pat;
        ac_level[k][0] = 0;
        for (c = 1; c < 16; c++) { rres[n2 + JMO_SNGL_SCAN[c][1]][n1 + JMO_SNGL_SCAN[c][0]] = 0; ac_level[k][c] = 0; }
      }
  }
  if (cr_cbp_tmp == 2) cr_cbp = 2;
  for (k = 0; k < nblk; k++) {                                                 /* ---- inverse transform, reconstruction ---- */
    const int n1 = 4 * (k & 1), n2 = 4 * (k >> 1);
    if (rres[n2][n1] != 0 || nonzero[k]) {
      int in[16], out[16];
      for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) in[4*j+i] = rres[n2+j][n1+i];
      jmo_inverse4x4(in, out);
      for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) rres[n2+j][n1+i] = out[4*j+i];
      any = 1;
    }
  }
  for (j = 0; j < H; j++) for (i = 0; i < 8; i++)
    rec[j*8+i] = any ? (jmo_pel)clip1(max_pel, rshift_rnd_sf(rres[j][i], 6) + (int)pred[j*8+i]) : pred[j*8+i];
  return cr_cbp;
}

/* residual_transform_quant_luma_16x16 block.c:208-349 (Intra16x16 luma of one macroblock, frame scan): sixteen forward4x4, the DC
 * coefficients through hadamard4x4 -> quant_dc4x4_normal (quant4x4_normal.c:200; also bound when AdaptiveRounding is on, quant4x4.c:56)
 * -> ihadamard4x4 -> rshift_rnd_sf((dc * InvScaleComp[0][0]) << qp_per, 6), the AC coefficients of every block through
 * quant_ac4x4_normal :117 / quant_ac4x4_around (quant4x4_around.c:129), inverse4x4 where the block has a DC or an AC level, and
 * sample_reconstruct with DQ_BITS = 6.  ac_level / ac_run are indexed b8 * 4 + b4 as JM's cofAC.  Returns ac_coef (15 or 0).
 * fadjust: JM hands quant_ac4x4_around the array ARCofAdj4x4[pl][I16MB] WITHOUT the block's row offset (block.c:247), so every block
 * row writes rows 0..3 and the last one wins; restated as is: fadjust[4][16], AC positions only (the DC positions are never written). */
int jmo_rtq_luma_16x16(const jmo_pel orig[256], const jmo_pel pred[256], const jmo_qparam q[16], int qp_per, int cavlc,
                       int adaptive_rounding, int arw, int max_pel, int dc_level[17], int dc_run[17],
                       int ac_level[16][16], int ac_run[16][16], jmo_pel rec[256], int fadjust[64])
{
  int t[16][16], dc[16], hd[16], j, i, jj, ii, c, ac_coef = 0, dc_nonzero;
  const int q_bits = 15 + qp_per;
  for (jj = 0; jj < 4; jj++)
    for (ii = 0; ii < 4; ii++) {
      int in[16], out[16];
      for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) in[4*j+i] = (int)orig[(4*jj+j)*16 + 4*ii+i] - (int)pred[(4*jj+j)*16 + 4*ii+i];
      jmo_forward4x4(in, out);
      for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) t[4*jj+j][4*ii+i] = out[4*j+i];
      dc[jj*4+ii] = out[0];
    }
  jmo_hadamard4x4(dc, hd);
  dc_nonzero = jmo_quant_dc4x4_normal(hd, &q[0], qp_per, cavlc, dc_level, dc_run);
  if (dc_nonzero) {
    int ih[16];
    jmo_ihadamard4x4(hd, ih);
    for (jj = 0; jj < 4; jj++) for (ii = 0; ii < 4; ii++) t[4*jj][4*ii] = rshift_rnd_sf((ih[jj*4+ii] * q[0].InvScaleComp) << qp_per, 6);
  } else
    for (jj = 0; jj < 4; jj++) for (ii = 0; ii < 4; ii++) t[4*jj][4*ii] = 0;
  for (jj = 0; jj < 4; jj++)
    for (ii = 0; ii < 4; ii++) {
      const int b = (2 * (jj >> 1) + (ii >> 1)) * 4 + 2 * (jj & 1) + (ii & 1);
      int run = 0, n = 0, nz = 0;
      for (c = 1; c < 16; c++) {
        const int si = JMO_SNGL_SCAN[c][0], sj = JMO_SNGL_SCAN[c][1], qi = sj * 4 + si, v = t[4*jj+sj][4*ii+si];
        int fadj = 0;
        if (v != 0) {
          const int scaled = iabs_(v) * q[qi].ScaleComp;
          int lev = (scaled + q[qi].OffsetComp) >> q_bits;
          if (lev != 0) {
            if (cavlc) lev = imin_(lev, 2063);
            if (adaptive_rounding) fadj = rshift_rnd_sf(arw * (scaled - (lev << q_bits)), q_bits + 1);
            lev = v < 0 ? -lev : lev;
            t[4*jj+sj][4*ii+si] = rshift_rnd_sf((lev * q[qi].InvScaleComp) << qp_per, 4);
            ac_level[b][n] = lev; ac_run[b][n] = run; n++; run = 0; nz = 1;
          } else { t[4*jj+sj][4*ii+si] = 0; run++; }
        } else run++;
        if (adaptive_rounding) fadjust[sj * 16 + 4*ii + si] = fadj;              /* row sj of the array, whatever the block row */
      }
      ac_level[b][n] = 0;
      if (nz) ac_coef = 15;
      if (t[4*jj][4*ii] != 0 || nz) {
        int in[16], out[16];
        for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) in[4*j+i] = t[4*jj+j][4*ii+i];
        jmo_inverse4x4(in, out);
        for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) t[4*jj+j][4*ii+i] = out[4*j+i];
      }
    }
  for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) rec[j*16+i] = (jmo_pel)clip1(max_pel, rshift_rnd_sf(t[j][i], 6) + (int)pred[j*16+i]);
  return ac_coef;
}
