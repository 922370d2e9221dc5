/*
 * oracle/jmo.h -- CPU restatement of JM 19.0 lencod's data-parallel hot path.
 *
 * TEST INFRASTRUCTURE.  This is the parity oracle: a plain-C restatement of the
 * reference's algorithms, one function per reference function, each citing the
 * reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path (jm_amd/,
 * include/jmhip.h) never links, imports or calls anything in this directory.
 *
 * Pinning: every function here is checked (tests/test_oracle_*.py, -m "not gpu")
 * against golden vectors captured from the reference itself -- the real lencod
 * built by oracle/Makefile.ref and tapped by oracle/ref_tap.c -- and against the
 * known-answer vectors of SURVEY.md Appendix C.
 *
 * Conventions mirrored from the reference:
 *   imgpel  = uint16_t   (lcommon/inc/typedefs.h:36, IMGTYPE 1 lencod/inc/defines.h:37)
 *   distblk = int64_t    (typedefs.h:39), all costs scaled <<5 (JCOST_CALC_SCALEUP,
 *                         LAMBDA_ACCURACY_BITS 5: defines.h:46,130)
 *   motion vectors in quarter-pel units, int16.
 */
#ifndef JMO_H
#define JMO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t jmo_pel;
typedef int64_t  jmo_dist;
#define JMO_DIST_MAX   (((jmo_dist)INT32_MAX) << 5)   /* DISTBLK_MAX = INT_MAX << LAMBDA_ACCURACY_BITS, lencod/inc/defines.h */
#define JMO_PAD_X 32   /* IMG_PAD_SIZE_X lencod/inc/defines.h:121 */
#define JMO_PAD_Y 20   /* IMG_PAD_SIZE_Y lencod/inc/defines.h:122 */
#define JMO_LAMBDA_BITS 5

typedef struct { int16_t x, y; } jmo_mv;

/* A reference picture as the ME sees it: up to 16 quarter-pel sub-planes
 * (StorablePicture.p_curr_img_sub[4][4], lencod/inc/mbuffer.h), each a padded
 * (H+2*PAD_Y) x (W+2*PAD_X) plane.  sub[j][i] points at picture sample (0,0)
 * of the plane with vertical phase j, horizontal phase i; rows are `pitch`
 * samples apart; samples at [-PAD_Y..H+PAD_Y-1][-PAD_X..W+PAD_X-1] are valid. */
typedef struct {
  int width, height;           /* size_x, size_y */
  int pitch;                   /* >= width + 2*PAD_X */
  const jmo_pel *sub[4][4];
} jmo_refpic;

/* ---- motion-estimation tables (lencod/src/mv_search.c:315-442) ---- */
int  jmo_mvbits(int d);                                   /* mv_search.c:366-374 */
void jmo_spiral(int search_range, jmo_mv *spiral);        /* mv_search.c:405-442, full-pel units */
int  jmo_spiral_index(int dx, int dy);                    /* closed form of the same order */

/* ---- block distortion (lencod/src/me_distortion.c) ---- */
int      jmo_hadamard_sad4x4(const int16_t diff[16]);     /* me_distortion.c:175-258 */
int      jmo_hadamard_sad8x8(const int16_t diff[64]);     /* me_distortion.c:266-341 */
/* computeSAD me_distortion.c:349-426 (luma only: ChromaMEEnable=0).
 * cand is the absolute quarter-pel position incl. pos_*_padded. */
jmo_dist jmo_compute_sad (const jmo_refpic *ref, const jmo_pel *orig, int bsx, int bsy,
                          jmo_dist min_mcost, int cand_x, int cand_y);
/* computeSATD me_distortion.c:745-825 */
jmo_dist jmo_compute_satd(const jmo_refpic *ref, const jmo_pel *orig, int bsx, int bsy,
                          int test8x8, jmo_dist min_mcost, int cand_x, int cand_y);

/* weighted / bi-predictive candidate distortions, me_distortion.c:434-740, :833-1180, :1261-1530 (luma only).
 * pred: 0 = BiPred*1 (average), 1 = BiPred*2 (weighted), 2 = *WP (one weighted reference), 3 = plain; metric: 0 SAD, 1 SSE, 2 SATD */
typedef struct { int weight[2], offset, round, shift; } jmo_wp;
jmo_dist jmo_compute_pred_dist(const jmo_refpic *ref1, const jmo_refpic *ref2, const jmo_pel *orig, int bsx, int bsy, int test8x8,
                               int metric, int pred, const jmo_wp *wp, int max_pel, jmo_dist min_mcost,
                               int cand1_x, int cand1_y, int cand2_x, int cand2_y);

/* ---- integer-pel search ---- */
typedef struct {
  int pos_x, pos_y;          /* block position in the picture, full-pel (MEBlock.pos_x/pos_y) */
  int bsx, bsy;              /* block size */
  jmo_mv pred;               /* MV predictor, quarter-pel */
  jmo_mv center;             /* in: search centre mv (quarter-pel, already rounded+clipped:
                                mv_search.c:924-957); out: best mv */
  int search_range;          /* full-pel: min(max_x,max_y)>>2, me_fullsearch.c:49 */
  int lambda_factor;         /* lambda_factor[F_PEL] */
  jmo_dist min_mcost;        /* in: initial bound (DISTBLK_MAX from BlockMotionSearch) */
} jmo_fs_job;

/* full_search_motion_estimation me_fullsearch.c:39-103 (rdopt on => no (0,0) bonus).
 * orig: bsx*bsy contiguous samples (MEBlock.orig_pic[0]).  Returns min_mcost,
 * updates job->center to the best mv.  If sad_evals != NULL it receives the
 * number of candidates that reached the SAD computation (early-out statistics). */
jmo_dist jmo_full_search(const jmo_refpic *ref, const jmo_pel *orig, jmo_fs_job *job,
                         long *sad_evals);

/* Fast full search (lencod/src/me_fullfast.c).
 * jmo_ffs_setup : setup_fast_full_search :269-608 (no WP, no chroma ME) +
 *                 update_full_search_large_blocks :195-260.
 *   cur: the 16x16 luma MB, row-major 256 samples.  center: search centre mv (quarter-pel,
 *   multiple of 4, already clipped :313-327).  block_sad[8][16][max_pos] (type 0 unused),
 *   max_pos=(2R+1)^2, indexed [blocktype][4x4 raster index][spiral pos]. */
void jmo_ffs_setup(const jmo_refpic *ref, const jmo_pel cur[256], int mb_x, int mb_y,
                   jmo_mv center, int search_range, uint32_t *block_sad /*[8][16][max_pos]*/);
/* fast_full_search_motion_estimation :618-689 (rdopt on). Returns min_mcost, *best_mv. */
jmo_dist jmo_ffs_search(const uint32_t *block_sad, int max_pos_table, int blocktype, int block_index,
                        jmo_mv center, jmo_mv pred, int search_range, int lambda_factor,
                        int max_mvd, jmo_dist min_mcost, jmo_mv *best_mv);

/* ---- sub-pel refinement: sub_pel_motion_estimation me_fullsearch.c:186-289 ----
 * rdopt on; metric per stage: 0 = SAD, 2 = SATD(Hadamard) (MEDistortionHPel/QPel).
 * start_hp/start_qp = p_Vid->start_me_refinement_hp/qp (mv_search.c:445-446). */
typedef struct {
  int pos_x, pos_y, bsx, bsy;
  jmo_mv pred;
  jmo_mv mv;                 /* in: int-pel result; out: refined */
  int lambda_h, lambda_q;    /* lambda_factor[H_PEL], [Q_PEL] */
  int metric_h, metric_q;
  int start_hp, start_qp;
  int test8x8;
  jmo_dist min_mcost;        /* in: cost after int-pel (reset to MAX by caller if !start_hp, mv_search.c:971-974) */
} jmo_subpel_job;
jmo_dist jmo_sub_pel_search(const jmo_refpic *ref, const jmo_pel *orig, jmo_subpel_job *job);

/* ---- transforms (lcommon/src/transform.c) ---- */
void jmo_forward4x4(const int in[16], int out[16]);          /* :20-68  */
void jmo_inverse4x4(const int in[16], int out[16]);          /* :70-118 */
void jmo_hadamard4x4(const int in[16], int out[16]);         /* :121-168 */
void jmo_ihadamard4x4(const int in[16], int out[16]);        /* :170-220 */
void jmo_hadamard2x2(const int in[4], int out[4]);           /* :284-297 */
void jmo_ihadamard2x2(const int in[4], int out[4]);          /* :299-312 */
void jmo_hadamard4x2(const int in[8], int out[8]);           /* :220-256, rows [2][4] */
void jmo_ihadamard4x2(const int in[8], int out[8]);          /* :258-298, out transposed [4][2] */
void jmo_forward8x8(const int in[64], int out[64]);          /* :353-448 */
void jmo_inverse8x8(const int in[64], int out[64]);          /* :450-547 */

/* ---- quantisation (lencod/src/quant4x4_normal.c, quant4x4_around.c, quant8x8_normal.c) ---- */
typedef struct { int OffsetComp, ScaleComp, InvScaleComp; } jmo_qparam;   /* LevelQuantParams quant_params.h:17-21 */
/* q_matrix.c:20-37 flat-matrix tables, q_offsets.c defaults */
void jmo_qparams_4x4(int qp, int intra, int offset_bits_val, jmo_qparam out[16]);  /* [j*4+i] */
void jmo_qparams_4x4_m(int qp, const int16_t off[16], jmo_qparam out[16]);     /* per-position offsets (update_q_offset4x4 q_offsets.c:238) */
void jmo_qparams_8x8_m(int qp, const int16_t off[64], jmo_qparam out[64]);     /* update_q_offset8x8 q_offsets.c:251 */
void jmo_qparams_8x8(int qp, int intra, int offset_bits_val, jmo_qparam out[64]);
/* quant_4x4_normal quant4x4_normal.c:39-115 (symbol_mode CAVLC => level clamp 2063).
 * tblock: 16 coeffs row-major [j][i], overwritten with the dequantised values.
 * level/run: 17 entries (0-terminated as in JM); returns nonzero flag; *coeff_cost accumulates. */
int jmo_quant_4x4_normal(int tblock[16], const jmo_qparam qp16[16], int qp_per, int cavlc,
                         const uint8_t *pos_scan /*16x2 (i,j)*/, const uint8_t *c_cost,
                         int level[17], int run[17], int *coeff_cost);
/* quant_4x4_around quant4x4_around.c:40-127 : additionally fills fadjust[16] */
int jmo_quant_4x4_around(int tblock[16], const jmo_qparam qp16[16], int qp_per, int cavlc,
                         const uint8_t *pos_scan, const uint8_t *c_cost, int adapt_rnd_weight,
                         int level[17], int run[17], int *coeff_cost, int fadjust[16]);
int jmo_quant_8x8_normal(int tblock[64], const jmo_qparam qp64[64], int qp_per, int cavlc,
                         const uint8_t *pos_scan /*64x2*/, const uint8_t *c_cost,
                         int level[65], int run[65], int *coeff_cost);
/* quant_8x8_normal / _around / quant_8x8cavlc_normal / _around (variant 0..3), see jmo_tq.c */
int jmo_quant_8x8(int tblock[64], const jmo_qparam qp64[64], int qp_per, int variant, const uint8_t *pos_scan, const uint8_t *c_cost,
                  int adapt_rnd_weight, int level[68], int run[68], int *coeff_cost, int fadjust[64]);
void jmo_scan8x8_cavlc(uint8_t out[64][2]);      /* SNGL_SCAN8x8_CAVLC, transform8x8.c */
/* quant_dc4x4_normal quant4x4_normal.c:200-259 */
int jmo_quant_dc4x4_normal(int tblock[16], const jmo_qparam *q, int qp_per, int cavlc, int level[17], int run[17]);
/* residual_transform_quant_luma_8x8 / _cavlc transform8x8.c:522 / :604, one 8x8 block */
int jmo_rtq_luma_8x8(const jmo_pel orig[64], const jmo_pel pred[64], const jmo_qparam q[64], int qp_per, int cavlc,
                     int adaptive_rounding, int adapt_rnd_weight, int max_pel, int level[68], int run[68], int *coeff_cost,
                     jmo_pel rec[64], int fadjust[64], int *any_residual);
/* residual_transform_quant_luma_16x16 block.c:208-349, one macroblock (see jmo_tq.c) */
int jmo_rtq_luma_16x16(const jmo_pel orig[256], const jmo_pel pred[256], const jmo_qparam q[16], int qp_per, int cavlc,
                       int adaptive_rounding, int arw, int max_pel, int dc_level[17], int dc_run[17],
                       int ac_level[16][16], int ac_run[16][16], jmo_pel rec[256], int fadjust[64]);
/* residual_transform_quant_chroma_4x4 block.c:954-1200, one plane of one macroblock (see jmo_tq.c) */
int jmo_rtq_chroma(int yuv, int uv, int cr_cbp, int64_t *cbp_blk, const jmo_qparam q_ac[16], const jmo_qparam *q_dc,
                   int qp_per_ac, int qp_per_dc, int cavlc, int adaptive_rounding, int adapt_rnd_weight, int max_pel,
                   const jmo_pel *orig, const jmo_pel *pred, jmo_pel *rec, int dc_level[9], int dc_run[9],
                   int ac_level[8][16], int ac_run[8][16], int fadjust[128]);
extern const uint8_t JMO_SNGL_SCAN[16][2];       /* block.c:170 */
extern const uint8_t JMO_SNGL_SCAN8x8[64][2];    /* transform8x8.c */
extern const uint8_t JMO_COEFF_COST4x4[3][16];   /* block.c COEFF_COST4x4 */
extern const uint8_t JMO_COEFF_COST8x8[2][64];

/* residual_transform_quant_luma_4x4 block.c:661-725, one 4x4 block:
 * orig/pred: 16 samples; returns nonzero; rec[16] reconstructed samples
 * (sample_reconstruct lcommon/src/blk_prediction.c:48-62, DQ_BITS 6). */
int jmo_rtq_luma_4x4(const jmo_pel orig[16], const jmo_pel pred[16], int qp, int intra,
                     int adaptive_rounding /*0: quant_4x4_normal, 1: _around*/, int adapt_rnd_weight,
                     int max_pel, int level[17], int run[17], int *coeff_cost,
                     jmo_pel rec[16], int fadjust[16]);

/* ---- sub-pel plane generation: getSubImagesLuma lencod/src/img_luma.c:611-679 ----
 * src: W x H luma (pitch src_pitch).  dst: 16 planes, plane (j,i) at dst + (j*4+i)*plane_stride,
 * each (H+2*PAD_Y) rows x pitch samples, picture origin at row PAD_Y, col PAD_X. */
/* ---- luma intra prediction and the Intra16x16 mode search (lencod/src/intra4x4.c, intra16x16.c; see jmo_intra.c) ---- */
void jmo_intrapred_4x4(const jmo_pel e[13], int mode, int left_available, int up_available, jmo_pel out[16]);
void jmo_intrapred_16x16(const jmo_pel e[33], int mode, int left_available, int up_available, int max_pel, jmo_pel out[256]);
jmo_dist jmo_dist_i16x16(const jmo_pel orig[256], const jmo_pel pred[256], int metric);
jmo_dist jmo_intra16_search(const jmo_pel e[33], int left_available, int up_available, int mode_mask, int metric, int max_pel,
                            const jmo_pel orig[256], jmo_pel pred4[4][256], int *best_mode);

/* get_intrapred_8x8 intra8x8.c:716 (nine modes :148-495) on the low-pass filtered predictor samples Z, A..P, Q..X */
void jmo_intrapred_8x8(const jmo_pel e[25], int mode, int left_available, int up_available, jmo_pel out[64]);
/* intra_chroma_prediction intra_chroma.c:530-778: DC / horizontal / vertical / plane of one chroma plane (8 wide, ch = 8 or 16 rows) */
int jmo_intra_chroma_pred(const jmo_pel *up, const jmo_pel *left, int corner, int up_avail, int left_avail, int upleft_avail,
                          int ch, int max_pel, jmo_pel pred[4][128]);

/* ---- source frame -> coded-size planes: buf2img_basic + pad_borders, lcommon/src/input.c:552-600, :880-925 (see jmo_interp.c) ---- */
void jmo_load_frame(const uint8_t *raw, int src_w, int src_h, int W, int H, int yuv, jmo_pel *y, jmo_pel *u, jmo_pel *v);
/* the general reader: 4:0:0 .. 4:4:4, one or two bytes per sample (little endian), bit depth conversion (buf2img_bitshift :440), source size != picture size (centred / cropped) */
void jmo_load_frame_ex(const uint8_t *raw, int yuv, int src_w, int src_h, int out_w, int out_h, int W, int H, int symbol_bytes,
                       const int src_depth[3], const int out_depth[3], jmo_pel *y, jmo_pel *u, jmo_pel *v);

/* ---- motion-compensated prediction, un-weighted (lencod/src/mc_prediction.c; see jmo_mc.c) ---- */
void jmo_luma_pred(const jmo_refpic *r0, const jmo_refpic *r1, int p_dir, int x, int y, int bsx, int bsy, jmo_mv mv0, jmo_mv mv1, jmo_pel *out);
void jmo_chroma_pred4x4(const jmo_pel *p0, const jmo_pel *p1, int pitch, int W, int H, int yuv, int p_dir, int xc, int yc,
                        const jmo_mv mv0[4][2], const jmo_mv mv1[4][2], jmo_pel out[16]);

/* weighted sample prediction of per-list predictions, mc_prediction.c:38-73 (jmo_wp: weight[list], offset, round, shift as the formula uses them) */
void jmo_weighted_samples(const jmo_pel *p0, const jmo_pel *p1, int n, int p_dir, const jmo_wp *wp, int max_pel, jmo_pel *out);

void jmo_sub_images_chroma(const jmo_pel *src, int pitch, int W, int H, int yuv, jmo_pel *dst);   /* img_chroma.c:338-437, see jmo_mc.c */
void jmo_sub_images_luma(const jmo_pel *src, int src_pitch, int width, int height,
                         int max_pel, jmo_pel *dst, int pitch, long plane_stride);

/* ---- deblocking: DeblockFrame lencod/src/loopFilter.c:63-71, DeblockMb :120-297,
 *      loop_filter_normal.c (strengths :52-292, edge filters :301-757) ----
 * Per-macroblock side information the filter reads from JM's Macroblock (lencod/inc/global.h). */
typedef struct {
  int16_t mb_type;           /* JM enum value: 0 PSKIP/BSKIP_DIRECT, 1 P16x16, 2 P16x8, 3 P8x16, 8 P8x8,
                                9 I4MB, 10 I16MB, 13 I8MB, 14 IPCM (lcommon/inc/types.h) */
  int16_t slice_type;        /* 0 P, 1 B, 2 I, 3 SP, 4 SI (SliceType) */
  int16_t qp;                /* MbQ->qp */
  int16_t qpc[2];            /* MbQ->qpc[uv] */
  int16_t cbp;               /* MbQ->cbp */
  uint32_t cbp_blk;          /* low 16 bits of MbQ->cbp_blk (luma 4x4 coefficient flags) */
  int16_t slice_nr;
  int16_t df_disable_idc;    /* MbQ->DFDisableIdc */
  int16_t df_alpha_c0;       /* MbQ->DFAlphaC0Offset */
  int16_t df_beta;           /* MbQ->DFBetaOffset */
  int16_t transform8x8;      /* MbQ->luma_transform_size_8x8_flag */
  int16_t pad_;
} jmo_db_mb;
/* per 4x4 block: enc_picture->mv_info[y][x] reduced to what GetStrength* compares:
 * mv per list and an integer identity for ref_pic[list] (-1 when ref_idx[list] == -1). */
typedef struct { int16_t mv[2][2]; int32_t ref_id[2]; } jmo_db_motion;
/* Deblock a whole frame in place (frame MBs, no MBAFF, 4:0:0 / 4:2:0 / 4:2:2).
 * imgY/imgU/imgV point at picture sample (0,0); pitches in samples.
 * direct_8x8_inference: active_sps->direct_8x8_inference_flag. */
void jmo_deblock_frame(jmo_pel *imgY, int pitchY, jmo_pel *imgU, jmo_pel *imgV, int pitchC,
                       int width, int height, int yuv_format /*0:400 1:420 2:422*/,
                       const jmo_db_mb *mbs, const jmo_db_motion *motion /*[H/4][W/4]*/,
                       int max_pel_y, int max_pel_c, int direct_8x8_inference);
/* strength of one edge (16 bytes), exposed for kernel-level parity: dir 0 vertical, 1 horizontal */
void jmo_deblock_strength(uint8_t str[16], int dir, int edge, int mb_addr, int mb_w,
                          const jmo_db_mb *mbs, const jmo_db_motion *motion);

/* ---- the RDO-off macroblock pipeline: encode_one_macroblock_low lencod/src/md_low.c:104 and everything it calls (see jmo_mbenc.c) ---- */
#define JMO_MAX_REF 16
typedef struct {
  int32_t width, height;        /* coded luma size (multiples of 16) */
  int32_t slice_type;           /* 0 P, 1 B (jmo_encode_slice_b), 2 I (SliceType, lcommon/inc/types.h) */
  int32_t first_mb, num_mb;     /* the slice: macroblocks [first_mb, first_mb + num_mb) in raster order */
  int32_t qp, qpc;              /* currMB->qp, currMB->qpc[0] (= qpc[1]) */
  int32_t search_range;         /* SearchRange (full-pel) */
  int32_t num_ref;              /* currSlice->listXsize[LIST_0] */
  int32_t lambda_mf[3];         /* p_Vid->lambda_mf[slice_type][qp][F_PEL, H_PEL, Q_PEL] */
  int32_t lambda_mdfp;          /* LAMBDA_FACTOR(p_Vid->lambda_md[slice_type][qp]) */
  int32_t max_mvd;              /* p_Vid->max_mvd, mv_search.c:327 */
  int32_t mv_limit[4];          /* MaxHmvR[4], MaxHmvR[5], MaxVmvR[4], MaxVmvR[5] (quarter-pel limits of the level, conformance.c:604) */
  int32_t inter_valid[8];       /* InterSearch[0][0][mode] */
  int32_t intra4_valid, intra16_valid;   /* enc_mb.valid[I4MB], [I16MB] (mode_decision.c:127-131) */
  int32_t subpel;               /* !DisableSubpelME */
  int16_t off4[3][2][16];       /* quantiser offsets (of 2048) of the 4x4 transform's coefficients, [Y, U, V][inter, intra][j * 4 + i], as this slice type uses them (CalculateOffset4x4Param
                                   q_offsets.c:633-711): the default lists are flat -- 342 inter, intra 682 in I and 342 in P slices --, a q_offset.cfg (OffsetMatrixPresentFlag) gives every
                                   position its own (the shipped file: luma intra DC 1024, intra 742 in I / 400 in P slices) */
  int32_t start_qp;             /* p_Vid->start_me_refinement_qp (mv_search.c:446); start_me_refinement_hp must be 0 */
  int32_t refbits[JMO_MAX_REF]; /* p_Vid->refbits, mv_search.c:376-385 */
  int32_t cabac;                /* currSlice->symbol_mode == CABAC: levels are not clamped to CAVLC_LEVEL_LIMIT (quant4x4_normal.c:84) */
  int32_t search_mode;          /* SearchMode: -1 (or 0 here: same thing) full search, 3 EPZS with EPZSSubPelGrid = 1 and EPZSSubPelME = 1 (needs a jmo_epzs_cfg) */
  int32_t transform8x8;         /* Transform8x8Mode: 0, or 1 = the 8x8 transform beside the 4x4 one (High profile): transform_decision for 16x16 / 16x8 / 8x16, the P8x8
                                   pass with 8x8 blocks only, Intra8x8, 8x8 Hadamard SATD in the sub-pel search of blocks of 8x8 samples and more */
  int16_t off8[2][64];          /* the same for the luma 8x8 transform, [inter, intra][j * 8 + i] (CalculateOffset8x8Param q_offsets.c:720) */
  int32_t intra8_valid;         /* enc_mb.valid[I8MB] (mode_decision.c:127) */
  int32_t qpc_cr_delta;         /* currMB->qpc[1] - currMB->qpc[0] (CrQPOffset != CbQPOffset in the High profiles), normally 0 */
  int32_t yuv_format;           /* 0 or 1: 4:2:0; 2: 4:2:2 (chroma planes width / 2 x height: 8 x 16 samples per macroblock, the 2x4 DC transform, DC quantiser of qpc + 3) */
} jmo_mbenc_cfg;

/* EPZS (SearchMode = 3): the configuration's switches and what EPZSSliceInit (lencod/src/me_epzs_common.c:620) reads from the decoded picture buffer. */
#define JMO_NO_REF (-(1 << 30))
typedef struct {
  int32_t pattern, dual, fixed, aggressive, temporal, spatial_mem, blocktype;   /* EPZSPattern, EPZSDualRefinement, EPZSFixedPredictors, EPZSAggressiveWindow, EPZSTemporal, EPZSSpatialMem, EPZSBlockType */
  int32_t min_scale, med_scale, max_scale, sub_scale;                           /* EPZSMinThresScale, EPZSMedThresScale, EPZSMaxThresScale, EPZSSubPelThresScale */
  int32_t poc_cur, poc_ref[JMO_MAX_REF];                                        /* enc_picture->poc, listX[LIST_0][r]->poc */
  /* listX[LIST_0][0] and [1] as pictures with motion: per 4x4 block the LIST_0 vector (x, y) and the poc of the picture it refers to (JMO_NO_REF: none).
   * [1] may be NULL when the list holds one picture. */
  const int16_t *col_mv[2];
  const int32_t *col_refpoc[2];
  int64_t alias_hits;           /* out: candidates JM skipped only because its 16-bit EPZSMap stamp had wrapped round (BlkCount, me_epzs_int.c:80) */
  int64_t searches;             /* out: EPZS searches run */
} jmo_epzs_cfg;

/* What encode_one_macroblock_low leaves behind for write_macroblock (lencod/src/macroblock.c:2810), one record per macroblock. */
typedef struct {
  int8_t   mb_type;             /* 0 PSKIP, 1 P16x16, 2 P16x8, 3 P8x16, 8 P8x8, 9 I4MB, 10 I16MB, 13 I8MB (MBModeTypes) */
  int8_t   i16mode;             /* currMB->i16mode as find_sad_16x16 left it */
  int8_t   c_ipred_mode;        /* currMB->c_ipred_mode as rdo_low_intra_chroma_decision left it (written for intra macroblocks only) */
  int8_t   transform8x8;        /* currMB->luma_transform_size_8x8_flag.  With it the 64 levels of 8x8 block b8 lie in the frame zig-zag order of the 8x8 scan at
                                   luma[4 * b8 + (s >> 4)][s & 15] (JM's CAVLC lists are that order de-interleaved: list s & 3, place s >> 2) */
  int16_t  cbp;                 /* currMB->cbp */
  int16_t  pad1;
  uint64_t cbp_blk;             /* currMB->cbp_blk */
  int64_t  min_rdcost;          /* currMB->min_rdcost */
  int8_t   b8mode[4];           /* currMB->b8x8[k].mode */
  int8_t   b8ref[4];            /* reference index of each 8x8 block, -1 intra */
  int8_t   ipredmode[16];       /* p_Vid->ipredmode, 4x4 raster */
  int8_t   ipred_syntax[16];    /* currMB->intra_pred_modes[4 * b8 + b4] */
  int16_t  mv[16][2];           /* enc_picture->mv_info[..].mv[LIST_0], 4x4 raster */
  int16_t  luma[16][16];        /* quantised levels in zig-zag scan order, block 4 * b8 + b4 (cofAC order); Intra16x16: AC levels at [1..15] */
  int16_t  luma_dc[16];         /* Intra16x16 DC levels, scan order */
  int16_t  chroma_dc[2][8];     /* [uv][scan position]: four levels with 4:2:0, eight (SCAN_YUV422 order, block.c:88) with 4:2:2 */
  int16_t  chroma_ac[2][8][16]; /* [uv][4x4 block in raster order of the plane: four (4:2:0) or eight (4:2:2: cofAC[4 + 2 uv + (k >> 2)][k & 3])][1..15] */
  /* B slices (zero otherwise) */
  int16_t  mv1[16][2];          /* enc_picture->mv_info[..].mv[LIST_1], 4x4 raster */
  int8_t   b8ref1[4];           /* ... .ref_idx[LIST_1] of each 8x8 block */
  int8_t   b8pdir[4];           /* currMB->b8x8[k].pdir: 0 list 0, 1 list 1, 2 both, -1 intra */
  int8_t   b8bipred[4];         /* currMB->b8x8[k].bipred: 0, or 1 / 2 = the vectors of the bi-predictive search (currSlice->bipred_mv[bipred - 1]) */
  int8_t   pad2[4];
} jmo_mb_record;                /* 1296 bytes */

typedef struct {                /* intermediate values, for localising a divergence (tests only) */
  int64_t motion_cost[8][4];    /* p_Vid->motion_cost[mode][LIST_0][0][block] */
  int16_t all_mv[8][16][2];     /* currSlice->all_mv[LIST_0][0][mode][by][bx] */
  int32_t best_mode, pad;
  int64_t motion_cost1[8][4];   /* p_Vid->motion_cost[mode][LIST_1][0][block] (B slices) */
} jmo_mb_debug;

/* B slices (jmo_mbenc_b.inc): what the slice has beside cfg (cfg->num_ref = listXsize[LIST_0], cfg->inter_valid = InterSearch[1][..]: [0] = BSliceDirect) */
typedef struct jmo_b_cfg_s {
  int32_t num_ref1;             /* currSlice->listXsize[LIST_1] */
  int32_t direct_8x8_inference; /* active_sps->direct_8x8_inference_flag */
  int32_t col_long_term;        /* listX[LIST_1][0]->is_long_term */
  int32_t bipred_me;            /* BiPredMotionEstimation */
  int32_t bipred_search[4];     /* BiPredSearch16x16 / 16x8 / 8x16 / 8x8 (p_Vid->bipred_enabled[1..4], slice.c:410) */
  int32_t bipred_refinements, bipred_range, bipred_subpel;   /* BiPredMERefinements, BiPredMESearchRange, BiPredMESubPel */
  const int8_t *col_ref;        /* listX[LIST_1][0]->mv_info[..].ref_idx[list] at [(y4 * w4 + x4) * 2 + list] */
  const int16_t *col_mv;        /* ... .mv[list] {x, y} at [((y4 * w4 + x4) * 2 + list) * 2] */
  /* DirectModeType 0 (temporal direct, Get_Direct_MV_Temporal mv_direct.c:40; frame pictures): */
  int32_t direct_temporal;      /* !currSlice->direct_spatial_mv_pred_flag */
  int32_t poc_cur, poc_l0[16], poc_l1_0;   /* enc_picture->poc, listX[LIST_0][i]->poc, listX[LIST_1][0]->poc: compute_colocated's mvscale (mbuffer.c:3122) and the mapping of the
                                              co-located block's reference picture into list 0 */
  const int32_t *col_refpoc;    /* listX[LIST_1][0]->mv_info[..].ref_pic[list]->poc at [(y4 * w4 + x4) * 2 + list] (only read where col_ref >= 0) */
} jmo_b_cfg;

/* refs / refc: list 0, refs1 / refc1: list 1; mv / ref_idx: list 0 of enc_picture->mv_info, mv1 / ref_idx1: list 1 */
int jmo_encode_slice_b(const jmo_mbenc_cfg *cfg, const jmo_b_cfg *bcfg, const jmo_pel *cur_y, const jmo_pel *cur_u, const jmo_pel *cur_v,
                       const jmo_refpic *refs, const jmo_pel *const *refc, const jmo_refpic *refs1, const jmo_pel *const *refc1,
                       jmo_pel *rec_y, jmo_pel *rec_u, jmo_pel *rec_v, int16_t *mv, int8_t *ref_idx, int16_t *mv1, int8_t *ref_idx1, int8_t *ipredmode,
                       jmo_mb_record *out, jmo_mb_debug *dbg);

int jmo_encode_slice(const jmo_mbenc_cfg *cfg, const jmo_pel *cur_y, const jmo_pel *cur_u, const jmo_pel *cur_v,
                     const jmo_refpic *refs, const jmo_pel *const *refc, jmo_pel *rec_y, jmo_pel *rec_u, jmo_pel *rec_v,
                     int16_t *mv, int8_t *ref_idx, int8_t *ipredmode, jmo_mb_record *out, jmo_mb_debug *dbg);
/* the same with the EPZS inputs (ez may be NULL when cfg->search_mode != 3) */
int jmo_encode_slice_ex(const jmo_mbenc_cfg *cfg, jmo_epzs_cfg *ez, const jmo_pel *cur_y, const jmo_pel *cur_u, const jmo_pel *cur_v,
                        const jmo_refpic *refs, const jmo_pel *const *refc, jmo_pel *rec_y, jmo_pel *rec_u, jmo_pel *rec_v,
                        int16_t *mv, int8_t *ref_idx, int8_t *ipredmode, jmo_mb_record *out, jmo_mb_debug *dbg);

#ifdef __cplusplus
}
#endif
/* ---- B slices: spatial direct mode (jmo_direct.c; lencod/src/mv_direct.c:522 Get_Direct_MV_Spatial_Normal) ---- */
void jmo_direct_spatial(const int8_t avail[3], const int8_t nref[3][2], const int16_t nmv[3][2][2], int col_long_term,
                        const int8_t col_ref[16][2], const int16_t col_mv[16][2][2],
                        int8_t ref_out[16][2], int8_t pdir_out[16], int16_t mv_out[16][2][2]);

#endif
