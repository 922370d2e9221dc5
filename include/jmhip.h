/*
 * jmhip.h -- C ABI of libjmhip: the MI355X (gfx950) implementation of JM 19.0 lencod's
 * data-parallel inner loop (integer / sub-pel motion estimation, 4x4 integer transform +
 * quantisation + reconstruction, sub-pel reference planes, in-loop deblocking).
 *
 * Plain C: opaque context, POD structs, pointers and sizes.  No C++/torch types.  The
 * library owns all device memory; the caller owns every host buffer.  Every entry point
 * returns 0 on success or a negative JMHIP_E* code, and never calls exit(); the message is
 * available from jmhip_last_error().  One host thread per context; all work of a context is
 * issued on one HIP stream (the caller's, if given at creation).  Entry points taking host
 * buffers are synchronous (they return after the results are in the host buffer); the
 * `_dev` twins take device pointers, enqueue on the context's stream and return at once.
 *
 * There is NO CPU fallback: without a HIP device jmhip_create() fails.
 *
 * Each entry point names the reference interface it stands in for (JM 19.0, paths
 * relative to the reference tree).  How a JM maintainer binds them: INTEGRATION.md.
 *
 * Sample type at the host boundary is JM's imgpel = uint16_t (lcommon/inc/typedefs.h:36);
 * on the device 8-bit video is kept as uint8 (bit_depth 8 is the only depth in this round).
 * Motion vectors are quarter-pel int16; costs are JM's distblk scaled by 32
 * (JCOST_CALC_SCALEUP, lencod/inc/defines.h:46) and fit int32 for 8-bit video.
 */
#ifndef JMHIP_H
#define JMHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define JMHIP_OK            0
#define JMHIP_EINVAL       -1   /* bad argument */
#define JMHIP_ENODEV       -2   /* no usable HIP device / not gfx950 */
#define JMHIP_ENOMEM       -3
#define JMHIP_EHIP         -4   /* a HIP runtime call failed */
#define JMHIP_EUNSUPPORTED -5
#define JMHIP_EREACH       -6   /* jmhip_seq_batch with EPZS: a search reached past what the queue's order covers (see there); the launch's pictures are void */

#define JMHIP_PAD_X 32          /* IMG_PAD_SIZE_X, lencod/inc/defines.h:121 */
#define JMHIP_PAD_Y 20          /* IMG_PAD_SIZE_Y, lencod/inc/defines.h:122 */
#define JMHIP_NPART 41          /* 1+2+2+4+8+8+16 partitions of the 7 inter block types */
#define JMHIP_MAX_SEARCH_RANGE 64

typedef struct jmhip_ctx jmhip_ctx;

typedef struct {
  int32_t device;          /* HIP device ordinal */
  int32_t width, height;   /* coded luma size, multiples of 16 (p_Vid->width / height) */
  int32_t yuv_format;      /* 0 = 4:0:0, 1 = 4:2:0, 2 = 4:2:2 (ColorFormat, lcommon/inc/types.h) */
  int32_t bit_depth;       /* 8 */
  int32_t search_range;    /* SearchRange in full pels; <= JMHIP_MAX_SEARCH_RANGE */
  int32_t num_ref_slots;   /* reference pictures kept resident (NumberReferenceFrames) */
  void   *stream;          /* hipStream_t to use, or NULL for the default stream */
} jmhip_config;

int         jmhip_create(jmhip_ctx **out, const jmhip_config *cfg);
void        jmhip_destroy(jmhip_ctx *ctx);
const char *jmhip_last_error(const jmhip_ctx *ctx);     /* ctx may be NULL: error of the last failed create */
int         jmhip_synchronize(jmhip_ctx *ctx);           /* wait for the context's stream and for every picture in flight (jmhip_seq_*): each is taken out of flight and its
                                                            device error word is read (the first error is returned); an entry that streams its records to the host keeps
                                                            answering jmhip_seq_record for its picture afterwards */
/* geometry of the resident padded planes: pitch (bytes), rows, bytes between the 16 sub-planes */
int         jmhip_plane_geometry(const jmhip_ctx *ctx, int32_t *pitch, int32_t *rows, int64_t *plane_stride);

/* The HIP stream every later call of this context launches on (hipStream_t; NULL = the default stream).  The context itself keeps no
 * work in flight across the switch: ordering between the old and the new stream is the caller's (events), which is what lets independent
 * stages of one picture -- e.g. the chroma and the luma residual paths -- run side by side.  One host thread per context, as before. */
int jmhip_set_stream(jmhip_ctx *ctx, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * Frames
 * ------------------------------------------------------------------------------------------ */
/* Current (source) luma: p_Vid->pCurImg, read by get_original_block (lencod/src/mv_search.c:786)
 * and setup_fast_full_search (lencod/src/me_fullfast.c:333-337). */
int jmhip_set_current(jmhip_ctx *ctx, const uint16_t *luma, int32_t pitch_samples);
int jmhip_set_current_dev(jmhip_ctx *ctx, const uint8_t *d_luma, int32_t pitch_bytes);

/* The source picture straight from the file's bytes: read_one_frame's buf2img (lcommon/src/input.c:792-868, buf2img_basic :552-600; 8-bit,
 * source size == output size) followed by pad_borders (:880-925; image.c:1243-1244) -- `raw` is one frame as it lies in the YUV file, planar
 * Y then U then V at src_w x src_h (chroma sub-sampled per the context's yuv_format); the coded-size planes (every sample right of / below
 * the picture repeats its left / upper neighbour) stay on the device: the luma plane becomes the current picture of the motion search
 * (as after jmhip_set_current), jmhip_current_planes_dev hands all three to the transform / quantisation stage as the originals, and
 * jmhip_get_current_planes copies them out as imgpel (tight pitches W and W / 2: what p_Vid->pImgOrg[0..2] hold). */
int jmhip_set_current_frame(jmhip_ctx *ctx, const uint8_t *raw, int32_t src_w, int32_t src_h);
int jmhip_set_current_frame_dev(jmhip_ctx *ctx, const uint8_t *d_raw, int32_t src_w, int32_t src_h);
/* The general reader: any of read_one_frame's planar cases -- 4:0:0 / 4:2:0 / 4:2:2 / 4:4:4, 8 .. 14 (16) bit samples in one or two bytes (little endian),
 * source bit depth == / > / < the coded one (buf2img_basic lcommon/src/input.c:552, buf2img_bitshift :440 with rshift_rnd, chosen as initInput :41-53 does), a file frame larger
 * or smaller than the picture (cropped / centred, :609-650), pad_borders (:880-925) -- into imgpel (uint16_t) planes of the coded size with tight pitches (coded_w; chroma by
 * yuv_format), independent of the context's own picture size and bit depth (the context supplies device and stream).  Bit-identical to the reference including its one oddity:
 * imgpel-sized samples of an equal-sized picture are copied with ONE memcpy (:568-570), so a two-byte picture whose width is not a multiple of 16 arrives sheared in the coded-width
 * planes -- that is what JM encodes, and what this returns.  Interleaved / RGB / TIFF input stays the host's (deinterleave :102, ReadTIFFImage). */
typedef struct {
  int32_t yuv_format;                   /* 0 4:0:0, 1 4:2:0, 2 4:2:2, 3 4:4:4 */
  int32_t src_w, src_h;                 /* source->width[0], height[0]: the frame in the file */
  int32_t out_w, out_h;                 /* output->width[0], height[0]: the picture to code */
  int32_t coded_w, coded_h;             /* p_Vid->width, height: out size rounded up to whole macroblocks (pad_borders' target) */
  int32_t symbol_bytes;                 /* source->pic_unit_size_shift3: 1 or 2 */
  int32_t src_depth[3], out_depth[3];   /* source->bit_depth[k], output->bit_depth[k] */
} jmhip_frame_format;
int jmhip_load_frame(jmhip_ctx *ctx, const jmhip_frame_format *f, const uint8_t *raw, uint16_t *y, uint16_t *u, uint16_t *v);           /* host frame in, host planes out */
int jmhip_load_frame_dev(jmhip_ctx *ctx, const jmhip_frame_format *f, const uint8_t *d_raw, uint16_t *d_y, uint16_t *d_u, uint16_t *d_v);   /* device to device, asynchronous */
/* the same from imgpel planes that already have the coded size (p_Vid->pCurImg, p_Vid->pImgOrg[1], [2] after pad_borders, lcommon/src/input.c:880);
 * asynchronous on the context's stream (the samples are copied into pinned staging before the call returns) */
int jmhip_set_current_planes(jmhip_ctx *ctx, const uint16_t *y, int32_t pitch_y, const uint16_t *u, const uint16_t *v, int32_t pitch_c);
int jmhip_current_planes_dev(jmhip_ctx *ctx, const uint8_t **d_y, int32_t *pitch_y, const uint8_t **d_u, const uint8_t **d_v, int32_t *pitch_c);
int jmhip_get_current_planes(jmhip_ctx *ctx, uint16_t *y, uint16_t *u, uint16_t *v);

/* Reference picture `slot` := reconstructed luma; builds the 16 quarter-pel planes on the device.
 * Replaces getSubImagesLuma(p_Vid, s) (lencod/src/img_luma.c:611-679), reached from
 * UnifiedOneForthPix (lencod/src/image.c:2187) when a picture enters the DPB (mbuffer.c:2313). */
int jmhip_set_reference(jmhip_ctx *ctx, int32_t slot, const uint16_t *luma, int32_t pitch_samples);
int jmhip_set_reference_dev(jmhip_ctx *ctx, int32_t slot, const uint8_t *d_luma, int32_t pitch_bytes);
/* Copy the planes back as JM lays them out: out[(j*4+i)] is plane p_curr_img_sub[j][i], each
 * (height+2*PAD_Y) rows of (width+2*PAD_X) imgpel, top-left = padded origin.  For host-side MC. */
int jmhip_get_subplanes(jmhip_ctx *ctx, int32_t slot, uint16_t *out);
/* Device address of the 16 planes of a slot (uint8, see jmhip_plane_geometry). */
const uint8_t *jmhip_subplanes_dev(jmhip_ctx *ctx, int32_t slot);

/* ------------------------------------------------------------------------------------------
 * Integer-pel motion estimation
 *
 * A job is one search window: one macroblock, one reference, one search centre, and the
 * subset of the macroblock's 41 partitions that are searched around that centre, each with
 * its own MV predictor.  It computes, for every position of the (2R+1)^2 window, the sixteen
 * 4x4 SADs (setup_fast_full_search, me_fullfast.c:492-556), aggregates them to the selected
 * partitions (update_full_search_large_blocks :195-260) and returns per partition
 *     argmin_pos  (SAD << 5) + lambda * (mvbits[cand_x - pred_x] + mvbits[cand_y - pred_y])
 * with ties resolved to the lowest index of JM's spiral (mv_search.c:405-442).  That is the
 * result of
 *   full_search_motion_estimation      (me_fullsearch.c:39-103, Macroblock.IntPelME, one job per
 *                                       distinct search centre; max_mvd = 0), and of
 *   fast_full_search_motion_estimation (me_fullfast.c:618-689, one job per macroblock/reference,
 *                                       all partitions, max_mvd = p_Vid->max_mvd)
 * for RDOptimization != 0 (no (0,0) bonus, bit-exactness checklist SURVEY.md 8a.4) and an
 * initial min_mcost of DISTBLK_MAX (mv_search.c:871-872).
 *
 * Partition order p = 0..40 (blocktype, top-left 4x4 block in raster units bx,by):
 *   0: 16x16 | 1,2: 16x8 (by=0,2) | 3,4: 8x16 (bx=0,2) | 5..8: 8x8 raster |
 *   9..16: 8x4 (by=0..3, bx=0,2 raster) | 17..24: 4x8 (by=0,2, bx=0..3 raster) | 25..40: 4x4 raster
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int16_t  mb_x, mb_y;              /* macroblock position in luma samples (pix_x, opix_y) */
  int16_t  center_x, center_y;      /* search centre MV, quarter-pel, already rounded to full-pel and clipped
                                       (mv_search.c:931-957 / me_fullfast.c:313-327) */
  int16_t  search_range;            /* full pels, <= config.search_range */
  int16_t  max_mvd;                 /* 0: no guard; else candidates with max|mvd| >= max_mvd-1 are skipped (me_fullfast.c:638,671) */
  int32_t  lambda;                  /* lambda_factor[F_PEL] */
  uint64_t part_mask;               /* bit p set: search partition p */
  int16_t  pred[JMHIP_NPART][2];    /* MV predictor per partition, quarter-pel */
  int16_t  reserved_[2];
} jmhip_me_job;                     /* 192 bytes */

typedef struct { int16_t mv_x, mv_y; int32_t cost; } jmhip_me_best;     /* quarter-pel MV, min_mcost */
typedef struct { jmhip_me_best best[JMHIP_NPART]; } jmhip_me_result;    /* 328 bytes; entries outside part_mask are untouched */

int jmhip_me_fullsearch(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *jobs, int32_t njobs, jmhip_me_result *results);
int jmhip_me_fullsearch_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *d_jobs, int32_t njobs, jmhip_me_result *d_results);

/* The BlockSAD tables themselves, for a host that keeps JM's own argmin (setup_fast_full_search as
 * bound to Macroblock.p_SetupFastFullPelSearch, lencod/inc/global.h:469): for each job (only mb_x,
 * mb_y, center, search_range are read) writes table[7][16][max_pos] uint16 in JM's order
 * [blocktype-1][4x4 raster index][spiral position], max_pos = (2*search_range+1)^2. */
int jmhip_me_sad_tables(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *jobs, int32_t njobs, uint16_t *tables);

/* ------------------------------------------------------------------------------------------
 * Candidate-list distortion and sub-pel refinement
 * ------------------------------------------------------------------------------------------ */
#define JMHIP_METRIC_SAD  0          /* ERROR_SAD  */
#define JMHIP_METRIC_SSE  1          /* ERROR_SSE: jmhip_distortion only */
#define JMHIP_METRIC_SATD 2          /* ERROR_SATD (Hadamard) */

/* computeSAD / computeSATD (me_distortion.c:349-426 / :745-825; MEBlock.computePred{F,H,Q}Pel,
 * lencod/inc/global.h:316-318) evaluated without the early exit: full distortion << 5. */
typedef struct {
  int16_t pos_x, pos_y;             /* block position, luma samples (MEBlock.pos_x/pos_y) */
  int16_t bsx, bsy;                 /* block size */
  int16_t cand_x, cand_y;           /* candidate MV relative to the block, quarter-pel */
  int16_t metric, test8x8;
} jmhip_cand;
int jmhip_me_eval(jmhip_ctx *ctx, int32_t slot, const jmhip_cand *cands, int32_t n, int32_t *dist);

/* The weighted and bi-predictive candidate distortions (me_distortion.c; MEBlock.computeBiPred{F,H,Q}Pel, lencod/inc/global.h:319-321,
 * and MEBlock.computePred{F,H,Q}Pel when MEBlock.apply_weights, mv_search.c:741-768), full distortion << 5 without the early exit:
 *   JMHIP_PRED_AVG     computeBiPred{SAD,SSE,SATD}1  :525 / :1353 / :943    p = (r1 + r2 + 1) >> 1
 *   JMHIP_PRED_BI_WP   computeBiPred{SAD,SSE,SATD}2  :624 / :1438 / :1038   p = clip1(((w1*r1 + w2*r2 + round) >> shift) + offset)
 *                      round = 2 * Slice.wp_luma_round, shift = Slice.luma_log_weight_denom + 1, weights = MEBlock.weight1 / weight2 / offsetBi
 *   JMHIP_PRED_UNI_WP  compute{SAD,SSE,SATD}WP       :434 / :1261 / :833    p = clip1(((w1*r1 + round) >> shift) + offset)
 *                      round = Slice.wp_luma_round, shift = Slice.luma_log_weight_denom, weights = MEBlock.weight_luma / offset_luma
 *   JMHIP_PRED_UNI     compute{SAD,SSE,SATD}         :349 / :1190 / :745    p = r1
 * All three metrics.  Luma only (MEBlock.ChromaMEEnable == 0).  computeBiPredSATD2's 8x8 path is reproduced with the source-pointer slip
 * of me_distortion.c:1167 (every further row of an 8x8 sub-block starts one source sample earlier), because the reference's results carry it. */
#define JMHIP_PRED_AVG    0
#define JMHIP_PRED_BI_WP  1
#define JMHIP_PRED_UNI_WP 2
#define JMHIP_PRED_UNI    3
typedef struct {
  int16_t pos_x, pos_y;             /* block position, luma samples */
  int16_t bsx, bsy;
  int16_t cand_x[2], cand_y[2];     /* candidate MV per reference, relative to the block, quarter-pel ([1] unused for the UNI kinds) */
  int8_t  slot[2];                  /* reference slot per list ([1] unused for the UNI kinds) */
  int8_t  metric, test8x8;          /* JMHIP_METRIC_*; test8x8: 8x8 Hadamard sub-blocks */
  int8_t  pred;                     /* JMHIP_PRED_* */
  int8_t  shift;                    /* 0..8 */
  int16_t weight[2], offset, round;
  int16_t reserved_;
} jmhip_pred_cand;                  /* 32 bytes */
int jmhip_me_eval_pred(jmhip_ctx *ctx, const jmhip_pred_cand *cands, int32_t n, int32_t *dist);
int jmhip_me_eval_pred_dev(jmhip_ctx *ctx, const jmhip_pred_cand *d_cands, int32_t n, int32_t *d_dist);

/* sub_pel_motion_estimation (me_fullsearch.c:186-289, Macroblock.SubPelME) for RDOptimization != 0:
 * up to 9 half-pel then 9 quarter-pel candidates around `mv`. */
typedef struct {
  int16_t pos_x, pos_y, bsx, bsy;
  int16_t pred_x, pred_y;
  int16_t mv_x, mv_y;               /* in: integer-pel result */
  int32_t lambda_h, lambda_q;       /* lambda_factor[H_PEL], [Q_PEL] */
  int8_t  metric_h, metric_q;       /* JMHIP_METRIC_* */
  int8_t  start_hp, start_qp;       /* p_Vid->start_me_refinement_hp / _qp (mv_search.c:445-446) */
  int8_t  test8x8, reserved_[3];
  int32_t min_mcost;                /* cost carried in when start_hp != 0; ignored (DISTBLK_MAX) otherwise */
} jmhip_subpel_job;                 /* 36 bytes */
int jmhip_me_subpel(jmhip_ctx *ctx, int32_t slot, const jmhip_subpel_job *jobs, int32_t n, jmhip_me_best *results);
int jmhip_me_subpel_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_subpel_job *d_jobs, int32_t n, jmhip_me_best *d_results);

/* BlockMotionSearch's hand-over from IntPelME to SubPelME (mv_search.c:960-981), device-resident: every
 * partition searched by jobs[i] is refined starting from int_results[i].best[p]; out[i].best[p] receives
 * the final MV and cost.  test8x8 = transform8x8_mode for block types 1-4 (mv_search.c:1630,1770). */
typedef struct {
  int32_t lambda_h, lambda_q;
  int8_t  metric_h, metric_q, start_hp, start_qp;
  int32_t transform8x8_mode;
} jmhip_refine_params;
int jmhip_me_refine_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *d_jobs, int32_t njobs, const jmhip_me_result *d_int_results,
                        const jmhip_refine_params *prm, jmhip_me_result *d_out);

/* ------------------------------------------------------------------------------------------
 * 4x4 luma residual: transform, quantise, dequantise, inverse transform, reconstruct
 *
 * residual_transform_quant_luma_4x4 (lencod/src/block.c:661-725, Macroblock.
 * residual_transform_quant_luma_4x4 global.h:465) = forward4x4 (lcommon/src/transform.c:20) +
 * Slice.quant_4x4 (quant_4x4_normal lencod/src/quant4x4_normal.c:39 or quant_4x4_around
 * quant4x4_around.c:40) + inverse4x4 (transform.c:70) + sample_reconstruct
 * (lcommon/src/blk_prediction.c:48), batched over independent 4x4 blocks.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int32_t OffsetComp, ScaleComp, InvScaleComp; } jmhip_qparam;   /* LevelQuantParams, lcommon/inc/quant_params.h:17-21 */
typedef struct {
  jmhip_qparam q[16];               /* p_Quant->q_params_4x4[pl][intra][qp][j][i] at index j*4+i */
  int32_t qp_per;                   /* p_Quant->qp_per_matrix[qp] */
  int32_t cavlc;                    /* symbol_mode == CAVLC: clamp |level| to 2063 */
  int32_t adaptive_rounding;        /* 0: quant_4x4_normal, 1: quant_4x4_around */
  int32_t adapt_rnd_weight;         /* p_Vid->AdaptRndWeight */
  int32_t max_pel;                  /* p_Vid->max_imgpel_value */
  int32_t reserved_[3];
} jmhip_tq_params;                  /* one per launch: uniform (plane, intra, qp) */
typedef struct {
  int16_t level[16];                /* ACLevel, zig-zag order, 0-terminated when fewer than 16 */
  uint8_t run[16];                  /* ACRun */
  int32_t coeff_cost;               /* contribution to *coeff_cost (999999 per |level| > 1) */
  uint8_t nonzero;                  /* return value */
  uint8_t any_residual;             /* check_zero(): 0 => JM leaves fadjust untouched */
  uint8_t ncoef, reserved_;
  uint8_t rec[16];                  /* reconstructed samples, row-major */
  int16_t fadjust[16];              /* ARCofAdj4x4 update (quant_4x4_around only), row-major */
} jmhip_tq_out;                     /* 104 bytes */
/* orig / pred: n blocks x 16 samples row-major (uint8). */
int jmhip_tq_luma4x4(jmhip_ctx *ctx, const jmhip_tq_params *prm, const uint8_t *orig, const uint8_t *pred,
                     int32_t nblocks, jmhip_tq_out *out);
int jmhip_tq_luma4x4_dev(jmhip_ctx *ctx, const jmhip_tq_params *prm, const uint8_t *d_orig, const uint8_t *d_pred,
                         int32_t nblocks, jmhip_tq_out *d_out);
/* the bare transforms, batched (n blocks x 16 int32, row-major): forward4x4 / inverse4x4 */
int jmhip_forward4x4(jmhip_ctx *ctx, const int32_t *in, int32_t nblocks, int32_t *out);
int jmhip_inverse4x4(jmhip_ctx *ctx, const int32_t *in, int32_t nblocks, int32_t *out);
int jmhip_forward8x8(jmhip_ctx *ctx, const int32_t *in, int32_t nblocks, int32_t *out);   /* transform.c:353 */
int jmhip_inverse8x8(jmhip_ctx *ctx, const int32_t *in, int32_t nblocks, int32_t *out);   /* transform.c:450 */

/* 8x8 luma residual: residual_transform_quant_luma_8x8 (lencod/src/transform8x8.c:522-586) and, with prm.cavlc,
 * residual_transform_quant_luma_8x8_cavlc (:604-672) = forward8x8 (lcommon/src/transform.c:353) + Slice.quant_8x8 /
 * quant_8x8cavlc (quant_8x8_normal lencod/src/quant8x8_normal.c:43, quant_8x8cavlc_normal :123, quant_8x8_around
 * quant8x8_around.c:43, quant_8x8cavlc_around :136) + inverse8x8 (transform.c:450) + sample_reconstruct (DQ_BITS_8 = 6),
 * frame scan (SNGL_SCAN8x8 / SNGL_SCAN8x8_CAVLC), COEFF_COST8x8[0]. */
typedef struct {
  jmhip_qparam q[64];               /* p_Quant->q_params_8x8[pl][intra][qp][j][i] at index j*8+i */
  int32_t qp_per, cavlc, adaptive_rounding, adapt_rnd_weight, max_pel;
  int32_t reserved_[3];
} jmhip_tq8_params;                 /* 800 bytes */
typedef struct {
  int16_t level[68];                /* cavlc 0: one 0-terminated list at [0..64]; cavlc 1: four lists of 17 at [17k..] (cofAC[k][0]) */
  uint8_t run[68];
  int32_t coeff_cost;
  uint8_t nonzero, any_residual;    /* any_residual: check_zero() of the non-CAVLC path (0 => JM leaves fadjust untouched) */
  uint8_t ncoef[4];                 /* entries in each list */
  uint8_t reserved_[2];
  uint8_t rec[64];                  /* reconstructed samples, row-major */
  int16_t fadjust[64];              /* ARCofAdj8x8 update (adaptive rounding only), row-major */
} jmhip_tq8_out;                    /* 408 bytes */
int jmhip_tq_luma8x8(jmhip_ctx *ctx, const jmhip_tq8_params *prm, const uint8_t *orig, const uint8_t *pred, int32_t nblocks, jmhip_tq8_out *out);
int jmhip_tq_luma8x8_dev(jmhip_ctx *ctx, const jmhip_tq8_params *prm, const uint8_t *d_orig, const uint8_t *d_pred, int32_t nblocks, jmhip_tq8_out *d_out);

/* Chroma residual of one plane of a macroblock: residual_transform_quant_chroma_4x4 (lencod/src/block.c:954-1200, slot
 * Macroblock.residual_transform_quant_chroma_4x4[uv], global.h:467) = forward4x4 per 4x4 block, the DC path (4:2:0: hadamard2x2 +
 * quant_dc2x2 + ihadamard2x2; 4:2:2: hadamard4x2 + quant_dc4x2 at qp+3 + ihadamard4x2; lencod/src/quantChroma_normal.c:37 / :110),
 * quant_ac4x4_normal / _around per block (quant4x4_normal.c:117 / quant4x4_around.c:129), the _CHROMA_COEFF_COST_ thresholding,
 * inverse4x4 and reconstruction; cbp_blk / cr_cbp are updated exactly as JM updates Macroblock.cbp_blk and the returned cr_cbp
 * (including the sign extension of JM's 32-bit DC mask for the V plane of 4:2:2).  Frame scan, disthres 0. */
typedef struct {
  jmhip_qparam q_ac[16];            /* p_Quant->q_params_4x4[uv+1][intra][cur_qp][j][i] at j*4+i */
  jmhip_qparam q_dc;                /* 4:2:0: the [0][0] entry of the same table; 4:2:2: of the table at cur_qp + 3 */
  int32_t qp_per_ac, qp_per_dc;     /* qp_per_matrix[cur_qp], qp_per_matrix[cur_qp (+3 for 4:2:2)] */
  int32_t yuv_format;               /* 1 = 4:2:0 (8x8 samples per plane), 2 = 4:2:2 (8x16) */
  int32_t cavlc, adaptive_rounding, adapt_rnd_weight, max_pel;
  int32_t reserved_[2];
} jmhip_tqc_params;                 /* 240 bytes; uniform per launch */
typedef struct { int64_t cbp_blk; int32_t cr_cbp; int32_t uv; } jmhip_tqc_mb;      /* per item, in/out: Macroblock.cbp_blk, cr_cbp; plane 0 = U, 1 = V */
typedef struct {
  int16_t dc_level[9];              /* cofDC[uv+1][0], 0-terminated */
  uint8_t dc_run[9];                /* cofDC[uv+1][1] */
  uint8_t dc_nonzero;
  int16_t ac_level[8][16];          /* cofAC[4 + b8 + uv_scale][b4][0], block k = 4*b8 + b4 = raster order of the plane's 4x4 blocks */
  uint8_t ac_run[8][16];
  uint8_t ac_ncoef[8];
  uint8_t rec[128];                 /* reconstructed samples, rows of 8 (4:2:0: the first 64) */
  int16_t fadjust[128];             /* ARCofAdj4x4 update (adaptive rounding; AC positions only, as in JM) */
  uint8_t reserved_[4];
} jmhip_tqc_out;                    /* 808 bytes */
/* orig / pred: n items x 128 samples (rows of 8, uint8). */
int jmhip_tq_chroma(jmhip_ctx *ctx, const jmhip_tqc_params *prm, jmhip_tqc_mb *mbs, const uint8_t *orig, const uint8_t *pred,
                    int32_t nitems, jmhip_tqc_out *out);
int jmhip_tq_chroma_dev(jmhip_ctx *ctx, const jmhip_tqc_params *prm, jmhip_tqc_mb *d_mbs, const uint8_t *d_orig, const uint8_t *d_pred,
                        int32_t nitems, jmhip_tqc_out *d_out);

/* Intra16x16 luma of whole macroblocks: Macroblock.residual_transform_quant_luma_16x16 (global.h:467) = residual_transform_quant_luma_16x16
 * (lencod/src/block.c:208-349): sixteen forward4x4, the DC coefficients through hadamard4x4 -> Slice.quant_dc4x4 (quant_dc4x4_normal,
 * quant4x4_normal.c:200) -> ihadamard4x4 -> their own dequantisation (block.c:294), the AC coefficients through Slice.quant_ac4x4
 * (quant_ac4x4_normal quant4x4_normal.c:117 / quant_ac4x4_around quant4x4_around.c:129), inverse4x4, sample_reconstruct.
 * prm: jmhip_tq_params with the INTRA quantiser of the macroblock's qp; orig / pred: n x 256 samples row-major (pred = mpr_16x16[i16mode]). */
typedef struct {
  uint8_t rec[256];                 /* reconstructed macroblock, rows of 16 */
  int16_t fadjust[4][16];           /* what JM's call leaves in rows 0..3 of ARCofAdj4x4[pl][I16MB] (adaptive rounding): it passes the array
                                       without the block's row offset (block.c:247), so these are block row 3's values; the DC position of
                                       each block ([0][0], [0][4], ...) is never written by JM -- ignore it */
  int16_t ac_level[16][16];         /* cofAC[b8][b4][0] at index b8 * 4 + b4, 0-terminated when fewer than 15 */
  int16_t dc_level[17];             /* cofDC[pl][0], 0-terminated */
  uint8_t ac_run[16][16];           /* cofAC[b8][b4][1] */
  uint8_t ac_ncoef[16];
  uint8_t dc_run[17];               /* cofDC[pl][1] */
  uint8_t dc_nonzero;
  uint8_t ac_coef;                  /* return value: 15 when any block has an AC level, else 0 */
  uint8_t reserved_[3];
} jmhip_tq16_out;                   /* 1224 bytes */
int jmhip_tq_luma16x16(jmhip_ctx *ctx, const jmhip_tq_params *prm, const uint8_t *orig, const uint8_t *pred, int32_t nmbs, jmhip_tq16_out *out);
int jmhip_tq_luma16x16_dev(jmhip_ctx *ctx, const jmhip_tq_params *prm, const uint8_t *d_orig, const uint8_t *d_pred, int32_t nmbs, jmhip_tq16_out *d_out);

/* The DC transforms of lcommon/src/transform.c, batched over blocks of int32 (row-major):
 *   HADAMARD4x4 :121 / IHADAMARD4x4 :170   16 values (Intra16x16 luma DC)
 *   HADAMARD4x2 :220 / IHADAMARD4x2 :258   8 values, rows [2][4]; the inverse returns JM's transposed [4][2] layout (4:2:2 chroma DC)
 *   HADAMARD2x2 :301 / IHADAMARD2x2 :316   4 values {dc00, dc01, dc10, dc11} (4:2:0 chroma DC) */
#define JMHIP_DC_HADAMARD4x4  0
#define JMHIP_DC_IHADAMARD4x4 1
#define JMHIP_DC_HADAMARD4x2  2
#define JMHIP_DC_IHADAMARD4x2 3
#define JMHIP_DC_HADAMARD2x2  4
#define JMHIP_DC_IHADAMARD2x2 5
int jmhip_dc_transform(jmhip_ctx *ctx, int32_t kind, const int32_t *in, int32_t nblocks, int32_t *out);
/* quant_dc4x4_normal (lencod/src/quant4x4_normal.c:200-259, Slice.quant_dc4x4): `blocks` (n x 16 int32, row-major) holds the
 * transformed DC coefficients on entry and the quantised LEVELS on return, as JM leaves them for ihadamard4x4. */
typedef struct { int16_t level[17]; uint8_t run[17]; uint8_t nonzero; } jmhip_dc_out;   /* 52 bytes */
int jmhip_quant_dc4x4(jmhip_ctx *ctx, const jmhip_qparam *q, int32_t qp_per, int32_t cavlc, int32_t *blocks, int32_t nblocks, jmhip_dc_out *out);

/* The block distortions of JM's mode decision, batched: VideoParameters.distortion4x4 / distortion8x8 (lencod/inc/global.h:1470-1471,
 * bound in lencod/src/me_distortion.c:148-166) = distortion{4x4,8x8}{SAD,SSE,SATD} (:38-146; HadamardSAD4x4 :175, HadamardSAD8x8 :266).
 * diff: n blocks of size*size int16 differences, row-major; out[i] = dist_scale(value) = value << 5.  (distortion8x8SADthres, the
 * early-exit variant, returns a partial sum when it exits and is not provided.) */
int jmhip_distortion(jmhip_ctx *ctx, int32_t metric, int32_t size, const int16_t *diff, int32_t nblocks, int64_t *out);

/* ------------------------------------------------------------------------------------------
 * Luma intra prediction and the Intra16x16 mode search
 *
 * jmhip_intrapred4x4: get_intrapred_4x4 (lencod/src/intra4x4.c:521-561; modes 0..8 = VERT, HOR, DC, DIAG_DOWN_LEFT, DIAG_DOWN_RIGHT,
 *   VERT_RIGHT, HOR_DOWN, VERT_LEFT, HOR_UP) over the predictor samples set_intrapred_4x4 (:421) leaves in currMB->intra4x4_pred[pl]:
 *   edge[0] = above left, edge[1..8] = above and above right, edge[9..12] = left.  left / up: the availability flags the DC mode takes.
 * jmhip_intra16_search: find_sad_16x16_JM (lencod/src/intra16x16.c:463-517; Slice.find_sad_16x16, bound in rdopt.c:301): the predictions
 *   of the modes in mode_mask (bit k = VERT_PRED_16, HOR_PRED_16, DC_PRED_16, PLANE_16; get_intrapred_16x16 :307) over
 *   currMB->intra16x16_pred[pl] (edge[0] = above left, [1..16] above, [17..32] left), their cost Slice.distI16x16 (metric: JMHIP_METRIC_*;
 *   distI16x16_sad / _sse / _satd :331-452) and the choice by strict '<' in ascending mode order.  8-bit samples.
 * ------------------------------------------------------------------------------------------ */
typedef struct { uint8_t edge[13]; uint8_t mode, left, up; } jmhip_ip4_blk;                    /* 16 bytes */
typedef struct { uint8_t edge[33]; uint8_t left, up, mode_mask, metric; uint8_t reserved_[3]; } jmhip_i16_mb;   /* 40 bytes */
typedef struct {
  int64_t cost;                     /* best cost (distblk, << 5); DISTBLK_MAX when mode_mask is empty */
  int32_t mode;                     /* what JM leaves in currMB->i16mode (DC_PRED_16 when nothing was evaluated) */
  int32_t reserved_;
  uint8_t pred[4][256];             /* mpr_16x16[pl][k] of the evaluated modes, rows of 16 */
} jmhip_i16_out;                    /* 1040 bytes */
int jmhip_intrapred4x4(jmhip_ctx *ctx, const jmhip_ip4_blk *blocks, int32_t n, uint8_t *out /* n x 16 */);
int jmhip_intra16_search(jmhip_ctx *ctx, const jmhip_i16_mb *mbs, const uint8_t *orig /* n x 256 */, int32_t n, jmhip_i16_out *out);
int jmhip_intra16_search_dev(jmhip_ctx *ctx, const jmhip_i16_mb *d_mbs, const uint8_t *d_orig, int32_t n, jmhip_i16_out *d_out);

/* ------------------------------------------------------------------------------------------
 * Motion-compensated prediction, un-weighted, frame pictures
 *
 * luma:   luma_prediction (lencod/src/mc_prediction.c:144-236; bound to p_Dpb->pf_luma_prediction, lencod.c:367):
 *         OneComponentLumaPrediction :122-136 copies block_size_y rows of block_size_x samples from the quarter-pel plane
 *         the vector's phase selects, from ONE UMVLine4X origin (refbuf.h:22-26); two lists: (a + b + 1) >> 1 (:82-93).
 * chroma: chroma_prediction_4x4 (:568-650) with ChromaMCBuffer = 1 (OneComponentChromaPrediction4x4_retrieve :361-411):
 *         per sample row and sample pair the vector of the luma 4x4 block above them, two samples from the chroma sub-image
 *         of the vector's phase (getSubImagesChroma, lencod/src/img_chroma.c:338-437) through UMVLine8X_chroma (refbuf.h:61-65).
 *         The device interpolates from the integer chroma planes on the fly -- the same values, without 64 (4:2:0) / 32 (4:2:2)
 *         stored sub-images per plane.
 * A reference slot's chroma planes are set with jmhip_set_reference_chroma (the luma sub-planes with jmhip_set_reference).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int16_t x, y;              /* block position in the picture, luma samples (currMB->pix_x + block_x, opix_y + block_y) */
  uint8_t w, h;              /* block size: 4, 8 or 16 each */
  uint8_t dir;               /* p_dir: 0 = list 0, 1 = list 1, 2 = both */
  uint8_t reserved_;
  int8_t  slot[2];           /* reference slot of list 0 / list 1 (ignored for an unused list) */
  int16_t mv[2][2];          /* [list][x, y], quarter-pel */
  int16_t reserved2_;
} jmhip_mc_luma_blk;         /* 20 bytes */
typedef struct {
  int16_t x, y;              /* 4x4 block position in the picture, chroma samples (pix_c_x + block_x, opix_c_y + block_y) */
  uint8_t dir, plane;        /* p_dir; 0 = U, 1 = V */
  int8_t  slot[2];
  int16_t mv[2][4][2][2];    /* [list][sample row][sample pair (0,1) / (2,3)][x, y]: the vectors the reference reads from all_mv */
} jmhip_mc_chroma_blk;       /* 72 bytes */
/* Weighted sample prediction (weighted_mc_prediction / weighted_bi_prediction, mc_prediction.c:38-73; luma_prediction :203-228,
 * chroma_prediction_4x4 :615-640), one record per block, same for luma and chroma:
 *   one list (dir 0 / 1):  clip1(((weight[dir] * p + round) >> shift) + offset)
 *                          weight = Slice.wp_weight[dir][ref][comp], offset = wp_offset[dir][ref][comp], round = wp_luma_round / wp_chroma_round,
 *                          shift = luma_log_weight_denom / chroma_log_weight_denom
 *   both lists (dir 2):    clip1(((weight[0] * p0 + weight[1] * p1 + round) >> shift) + offset)
 *                          weight[l] = Slice.wbp_weight[l][ref0][ref1][comp], offset = (wp_offset[0][ref0][comp] + wp_offset[1][ref1][comp] + 1) >> 1,
 *                          round = 2 * wp_*_round, shift = *_log_weight_denom + 1 */
typedef struct {
  int16_t weight[2];
  int16_t offset, round;
  int8_t  shift;             /* 0..8 */
  int8_t  reserved_[3];
} jmhip_mc_weights;          /* 12 bytes */
/* out: luma n x 256 bytes (the w x h samples row-major at the start of each record); chroma n x 16 bytes */
int jmhip_set_reference_chroma(jmhip_ctx *ctx, int32_t slot, const uint16_t *u, const uint16_t *v, int32_t pitch_samples);
int jmhip_set_reference_chroma_dev(jmhip_ctx *ctx, int32_t slot, const uint8_t *d_u, const uint8_t *d_v, int32_t pitch_bytes);
/* K6: getSubImagesChroma (lencod/src/img_chroma.c:338-437) of plane 0 = U / 1 = V of the reference in `slot`, for callers that keep JM's
 * host-side chroma prediction: out[suby][subx][ch + 2 pad_y][cw + 2 pad_x] imgpel, suby < 8 (4:2:0) / 4 (4:2:2), subx < 8,
 * pad_x = 16, pad_y = 10 (4:2:0) / 20 (4:2:2)  (lencod.c:2366-2376). */
int jmhip_get_chroma_subplanes(jmhip_ctx *ctx, int32_t slot, int32_t plane, uint16_t *out);
int jmhip_mc_luma(jmhip_ctx *ctx, const jmhip_mc_luma_blk *blocks, int32_t n, uint8_t *out);
int jmhip_mc_luma_dev(jmhip_ctx *ctx, const jmhip_mc_luma_blk *d_blocks, int32_t n, uint8_t *d_out);
int jmhip_mc_chroma(jmhip_ctx *ctx, const jmhip_mc_chroma_blk *blocks, int32_t n, uint8_t *out);
int jmhip_mc_chroma_dev(jmhip_ctx *ctx, const jmhip_mc_chroma_blk *d_blocks, int32_t n, uint8_t *d_out);
/* the same with weighted prediction: weights[i] belongs to blocks[i]; weights == NULL is the un-weighted call */
int jmhip_mc_luma_wp(jmhip_ctx *ctx, const jmhip_mc_luma_blk *blocks, const jmhip_mc_weights *weights, int32_t n, uint8_t *out);
int jmhip_mc_luma_wp_dev(jmhip_ctx *ctx, const jmhip_mc_luma_blk *d_blocks, const jmhip_mc_weights *d_weights, int32_t n, uint8_t *d_out);
int jmhip_mc_chroma_wp(jmhip_ctx *ctx, const jmhip_mc_chroma_blk *blocks, const jmhip_mc_weights *weights, int32_t n, uint8_t *out);
int jmhip_mc_chroma_wp_dev(jmhip_ctx *ctx, const jmhip_mc_chroma_blk *d_blocks, const jmhip_mc_weights *d_weights, int32_t n, uint8_t *d_out);

/* Device-resident glue between the stages of a P picture coded as 16x16 macroblocks (no host round trip):
 * jmhip_mc_mb16_dev   luma_prediction of every window job's 16x16 partition with the vector the refinement left in
 *                     results[job].best[0], written as sixteen 4x4 blocks of 16 samples each in picture block-raster order
 *                     (block (by, bx) at index by * blocks_per_row + bx): the layout jmhip_tq_luma4x4_dev reads.  job.mb_y - y_offset
 *                     is the macroblock's row in that block array (y_offset: rows of halo above a band, else 0).
 * jmhip_tq_rec_to_plane_dev  the reconstructed samples of jmhip_tq_out records (same block order) assembled into a plane.
 * jmhip_mc_mb16_chroma_dev   chroma_prediction_4x4 of both planes of every job's macroblock with the same vector, written as items
 *                     2 * job + plane of 128 samples (rows of 8; 4:2:0 uses the first 64): the layout jmhip_tq_chroma_dev reads.
 * jmhip_tqc_rec_to_planes_dev  the reconstructed samples of those items' jmhip_tqc_out records put into the U and V planes. */
int jmhip_mc_mb16_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *d_jobs, const jmhip_me_result *d_results, int32_t njobs,
                      int32_t y_offset, int32_t blocks_per_row, uint8_t *d_pred_blocks);
int jmhip_tq_rec_to_plane_dev(jmhip_ctx *ctx, const jmhip_tq_out *d_out, int32_t nblocks, int32_t blocks_per_row, uint8_t *d_plane, int32_t pitch_bytes);
/* The three luma calls above in ONE launch (the prediction stays in registers): jmhip_mc_mb16_dev -> jmhip_tq_luma4x4_dev on
 * (d_orig_blocks, that prediction) -> jmhip_tq_rec_to_plane_dev.  Same records in d_out (block order, every block of the listed
 * macroblocks), same samples in d_plane (row 0 = picture row y_offset); d_pred_blocks may be NULL (else it receives the prediction).
 * Replaces, per macroblock coded P16x16: luma_prediction (mc_prediction.c:144) + residual_transform_quant_luma_4x4 (block.c:661-725)
 * of its sixteen blocks with the reconstruction written to enc_picture. */
int jmhip_mb16_recon_luma_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_tq_params *prm, const jmhip_me_job *d_jobs, const jmhip_me_result *d_results,
                              int32_t njobs, int32_t y_offset, int32_t blocks_per_row, const uint8_t *d_orig_blocks, jmhip_tq_out *d_out,
                              uint8_t *d_pred_blocks, uint8_t *d_plane, int32_t pitch_bytes);
int jmhip_mc_mb16_chroma_dev(jmhip_ctx *ctx, int32_t slot, const jmhip_me_job *d_jobs, const jmhip_me_result *d_results, int32_t njobs, uint8_t *d_pred_items);
int jmhip_tqc_rec_to_planes_dev(jmhip_ctx *ctx, const jmhip_me_job *d_jobs, const jmhip_tqc_out *d_out, int32_t njobs, int32_t y_offset,
                                uint8_t *d_u, uint8_t *d_v, int32_t pitch_bytes);

/* Intra8x8 luma prediction: get_intrapred_8x8 (lencod/src/intra8x8.c:716-760; the nine modes :148-495).  edge = currMB->intra8x8_pred[pl][0..24]
 * = Z, A..P (16 samples above and above-right), Q..X (8 samples left) as set_intrapred_8x8 (:497-601) leaves them, i.e. after
 * LowPassForIntra8x8Pred (:85-140); mode 0..8 (VERT_PRED .. HOR_UP_PRED); left / up = the availability flags get_intrapred_8x8 is given
 * (only the DC mode reads them).  out: n x 64 samples, row-major 8x8: Slice.mpr_8x8[pl][mode]. */
typedef struct {
  uint8_t edge[25];
  uint8_t mode, left, up;
} jmhip_ip8_blk;             /* 28 bytes */
int jmhip_intrapred8x8(jmhip_ctx *ctx, const jmhip_ip8_blk *blks, int32_t n, uint8_t *out);

/* Chroma intra prediction: intra_chroma_prediction (lencod/src/intra_chroma.c:530-778; Slice.intra_chroma_prediction, bound in slice.c:1135),
 * frame macroblocks, 4:2:0 (8x8) and 4:2:2 (8x16): the DC (per 4x4 block, :590-686), horizontal, vertical and plane (:718-747) predictions of
 * both planes.  The caller gathers the neighbour samples the way the function does (getNeighbour + UseConstrainedIntraPred, :556-574):
 *   up[plane][i]   = imgUV[plane][pix_c.pos_y][pix_c.pos_x + i], i < 8       left[plane][j] = imgUV[plane][pix_a.pos_y + j][pix_a.pos_x], j < 8 / 16
 *   corner[plane]  = imgUV[plane][pix_d.pos_y][pix_d.pos_x]
 * out: n x 1024 bytes = [mode DC_PRED_8 0 / HOR_PRED_8 1 / VERT_PRED_8 2 / PLANE_8 3][plane][16 rows x 8]; a mode whose neighbours are missing
 * (the reference leaves Slice.mpr_16x16 alone then) is zeros.  rdo_low_intra_chroma_decision (RDOptimization = 0) stays with the host. */
typedef struct {
  uint8_t up[2][8];
  uint8_t left[2][16];
  uint8_t corner[2];
  uint8_t up_avail, left_avail, upleft_avail;
  uint8_t reserved_[3];
} jmhip_ic_mb;               /* 56 bytes */
int jmhip_intra_chroma(jmhip_ctx *ctx, const jmhip_ic_mb *mbs, int32_t n, uint8_t *out);
int jmhip_intra_chroma_dev(jmhip_ctx *ctx, const jmhip_ic_mb *d_mbs, int32_t n, uint8_t *d_out);

/* ------------------------------------------------------------------------------------------
 * In-loop deblocking of a whole frame
 *
 * DeblockFrame(p_Vid, imgY, imgUV) (lencod/src/loopFilter.c:63-71; DeblockMb :120-297, strengths and
 * edge filters lencod/src/loop_filter_normal.c), frame pictures without MBAFF.  Results equal JM's
 * raster-order in-place filtering.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int16_t  mb_type;          /* Macroblock.mb_type (JM enum value) */
  int16_t  slice_type;       /* p_Slice->slice_type: 0 P, 1 B, 2 I, 3 SP, 4 SI */
  int16_t  qp;               /* Macroblock.qp */
  int16_t  qpc[2];           /* Macroblock.qpc[] */
  int16_t  cbp;              /* Macroblock.cbp */
  uint32_t cbp_blk;          /* Macroblock.cbp_blk & 0xFFFF */
  int16_t  slice_nr;
  int16_t  df_disable_idc, df_alpha_c0, df_beta;     /* DFDisableIdc, DFAlphaC0Offset, DFBetaOffset */
  int16_t  transform8x8;     /* luma_transform_size_8x8_flag */
  int16_t  reserved_;
} jmhip_db_mb;               /* 28 bytes */
typedef struct { int16_t mv[2][2]; int32_t ref_id[2]; } jmhip_db_motion;   /* per 4x4: mv[list][x,y], identity of ref_pic[list] or -1 */

int jmhip_deblock_frame(jmhip_ctx *ctx, uint16_t *imgY, int32_t pitchY, uint16_t *imgU, uint16_t *imgV, int32_t pitchC,
                        const jmhip_db_mb *mbs, const jmhip_db_motion *motion, int32_t direct_8x8_inference);
int jmhip_deblock_frame_dev(jmhip_ctx *ctx, uint8_t *d_Y, int32_t pitchY, uint8_t *d_U, uint8_t *d_V, int32_t pitchC,
                            const jmhip_db_mb *d_mbs, const jmhip_db_motion *d_motion, int32_t direct_8x8_inference);

/* ------------------------------------------------------------------------------------------
 * The RDO-off macroblock pipeline of a whole slice
 *
 * encode_one_macroblock_low (lencod/src/md_low.c:104-687) for every macroblock of a P or I slice, in the wavefront order x + 2y
 * that its neighbour dependencies allow: motion vector prediction (GetMotionVectorPredictorNormal lcommon/src/mv_prediction.c:194 over
 * get_neighbors lencod/src/mv_search.c:268), BlockMotionSearch (mv_search.c:857: full_search_motion_estimation me_fullsearch.c:39 around
 * each block's own rounded predictor with the (0,0) bonus of the RDO-off encoder, sub_pel_motion_estimation :186, the skip vector's
 * cost mv_search.c:983-998), list_prediction_cost (mode_decision.c:275), submacroblock_mode_decision_low (mode_decision_P8x8.c:681),
 * mode_decision_for_I4x4_MB (rd_intra_jm.c:386, rd_intra_jm_low.c:39), find_sad_16x16_JM (intra16x16.c:463), luma_residual_coding
 * (macroblock.c:1182) / set_coeff_and_recon_8x8_p_slice (rdopt.c:1326), rdo_low_intra_chroma_decision (intra_chroma.c:460),
 * chroma_residual_coding (macroblock.c:1439), the skip test (md_low.c:658).  What the call leaves behind per macroblock is what
 * write_macroblock (macroblock.c:2810) and DeblockFrame read: one jmhip_mb_record.  The host keeps the entropy coder.
 *
 * Scope: frame macroblocks, 4:2:0 or 4:2:2 (the context's yuv_format), 8 bit, 4x4 and 8x8 transform, no adaptive rounding / weighted prediction / rate control,
 * SearchMode -1, 0 or 3 (EPZS),
 * unconstrained intra prediction, num_ref * window bytes within the LDS (5 references at SearchRange 32, 16 at 16); anything else
 * returns JMHIP_EUNSUPPORTED and the caller keeps JM's own function.
 * The source picture is the one jmhip_set_current_frame loaded; the references are slots filled by jmhip_set_reference[_chroma] or
 * jmhip_reference_from_recon.  The reconstruction stays on the device (jmhip_recon_planes_dev), so do the records' loop-filter
 * side information (jmhip_deblock_picture[_dev]).
 * ------------------------------------------------------------------------------------------ */
#define JMHIP_MB_MAX_REF 16
#define JMHIP_SEQ_MAX_DEPTH 16        /* pictures in flight at most (jmhip_seq_open) */
typedef struct {
  int8_t   mb_type;             /* 0 PSKIP, 1 P16x16, 2 P16x8, 3 P8x16, 8 P8x8, 9 I4MB, 10 I16MB, 13 I8MB (MBModeTypes, lencod/inc/defines.h:170-185) */
  int8_t   i16mode;             /* currMB->i16mode as find_sad_16x16 left it */
  int8_t   c_ipred_mode;        /* currMB->c_ipred_mode as rdo_low_intra_chroma_decision left it (written for intra macroblocks) */
  int8_t   transform8x8;        /* currMB->luma_transform_size_8x8_flag.  With it the 64 levels of 8x8 block b8 lie in the 8x8 scan's zig-zag order at
                                   luma[4 * b8 + (s >> 4)][s & 15]: JM keeps them as one 64-entry list (CABAC, cofAC[b8][0]) or de-interleaved into four lists
                                   (CAVLC, quant_8x8cavlc_normal: list s & 3, place s >> 2); Intra8x8: ipredmode per 4x4 block, ipred_syntax at [4 * b8] */
  int16_t  cbp;                 /* currMB->cbp */
  int16_t  reserved1_;
  uint64_t cbp_blk;             /* currMB->cbp_blk */
  int64_t  min_rdcost;          /* currMB->min_rdcost */
  int8_t   b8mode[4];           /* currMB->b8x8[k].mode */
  int8_t   b8ref[4];            /* reference index of each 8x8 block (enc_picture->mv_info[..].ref_idx[LIST_0]), -1 intra */
  int8_t   ipredmode[16];       /* p_Vid->ipredmode, 4x4 raster */
  int8_t   ipred_syntax[16];    /* currMB->intra_pred_modes[4 * b8 + b4] */
  int16_t  mv[16][2];           /* enc_picture->mv_info[..].mv[LIST_0], 4x4 raster */
  int16_t  luma[16][16];        /* quantised levels in zig-zag scan order, block 4 * b8 + b4 (cofAC order); Intra16x16: AC levels at [1..15] */
  int16_t  luma_dc[16];         /* Intra16x16 DC levels (cofDC[0]), scan order */
  int16_t  chroma_dc[2][8];     /* cofDC[1 + uv] at the levels' scan positions: four with 4:2:0, eight with 4:2:2 (SCAN_YUV422, block.c:88) */
  int16_t  chroma_ac[2][8][16]; /* levels at [1..15] of the plane's 4x4 blocks in raster order: 4:2:0 cofAC[4 + uv][k] (k < 4), 4:2:2 cofAC[4 + 2 uv + (k >> 2)][k & 3] (k < 8) */
  /* B slices (slice_type 1; zero otherwise).  mb_type 0 is then B_Skip / Direct16x16 (Direct when cbp != 0), a sub-mode 0 of P8x8 a direct 8x8 block */
  int16_t  mv1[16][2];          /* enc_picture->mv_info[..].mv[LIST_1], 4x4 raster */
  int8_t   b8ref1[4];           /* ... .ref_idx[LIST_1] of each 8x8 block (-1: the block does not use list 1); b8ref is -1 where it does not use list 0 */
  int8_t   b8pdir[4];           /* currMB->b8x8[k].pdir: 0 list 0, 1 list 1, 2 both (the average of the two predictions), -1 intra */
  int8_t   b8bipred[4];         /* currMB->b8x8[k].bipred: 0, or 1 / 2 = the block's vectors are those of the bi-predictive search (currSlice->bipred_mv[bipred - 1]: write_motion_info
                                   and the prediction take them from there) */
  int8_t   reserved2_[4];
} jmhip_mb_record;              /* 1296 bytes */

typedef struct {
  int32_t slice_type;           /* 0 P, 1 B, 2 I */
  int32_t first_mb, num_mb;     /* macroblocks [first_mb, first_mb + num_mb) in raster order */
  int32_t slice_nr;             /* Macroblock.slice_nr (loop filter side information) */
  int32_t qp, qpc;              /* currMB->qp, currMB->qpc[0] (= qpc[1]) */
  int32_t search_range;         /* SearchRange (full-pel), <= the context's */
  int32_t num_ref;              /* currSlice->listXsize[LIST_0] */
  int32_t ref_slot[JMHIP_MB_MAX_REF];  /* device slot of listX[LIST_0][r]; B slices: listX[LIST_1][r] follows at [num_ref + r] (num_ref + num_ref1 <= 16) */
  int32_t ref_id[JMHIP_MB_MAX_REF];    /* identity of that picture for the loop filter's ref_pic comparison (B slices: list 1 at [num_ref + r], as ref_slot) */
  int32_t lambda_mf[3];         /* p_Vid->lambda_mf[slice_type][qp][F_PEL, H_PEL, Q_PEL]: JM's double arithmetic, never recomputed */
  int32_t lambda_mdfp;          /* LAMBDA_FACTOR(p_Vid->lambda_md[slice_type][qp]) */
  int32_t max_mvd;              /* p_Vid->max_mvd (mv_search.c:327) */
  int32_t mv_limit[4];          /* MaxHmvR[4], MaxHmvR[5], MaxVmvR[4], MaxVmvR[5] */
  int32_t inter_valid[8];       /* enc_mb.valid[mode] (InterSearch) */
  int32_t intra4_valid, intra16_valid;
  int32_t subpel;               /* !DisableSubpelME */
  int32_t start_qp;             /* p_Vid->start_me_refinement_qp (mv_search.c:446); start_me_refinement_hp must be 0 */
  int32_t refbits[JMHIP_MB_MAX_REF];   /* p_Vid->refbits */
  jmhip_qparam q_luma[2][16];   /* p_Quant->q_params_4x4[0][intra][qp][j][i] at [intra][j * 4 + i] */
  jmhip_qparam q_chroma[2][2][16];  /* p_Quant->q_params_4x4[1 + uv][intra][qpc + chroma scale] at [uv][intra][j * 4 + i] */
  int32_t df_disable_idc, df_alpha_c0, df_beta;   /* Macroblock.DFDisableIdc, DFAlphaC0Offset, DFBetaOffset of the slice */
  int32_t num_slices;           /* 0 or 1: one slice.  n > 1: the call covers n consecutive slices of num_mb macroblocks each (the last one ends with the
                                   picture), same parameters, slice_nr counting up -- SliceMode 1 pictures: the slices' wavefronts run side by side */
  int32_t symbol_mode;          /* currSlice->symbol_mode: 0 = CAVLC (levels clamped to CAVLC_LEVEL_LIMIT = 2063: quant4x4_normal.c:84, :160, :233, quantChroma_normal.c:69), 1 = CABAC (no clamp) */
  int32_t search_mode;          /* 0: full_search_motion_estimation (JM's SearchMode -1); 1: fast_full_search_motion_estimation (SearchMode 0, me_fullfast.c:618: one search
                                   centre per macroblock and reference, the (0,0) vector first; RestrictSearchRange 2; refused when mv_limit cuts into the range -- mv_limit[3] - 4 R < 4 R etc.: JM's
                                   centre then leaves the sample grid and it reads the pos_00 an earlier macroblock left, me_fullfast.c:326-327, :354-365); 3: EPZS (SearchMode 3) with EPZSSubPelGrid = 1 and EPZSSubPelME = 1,
                                   the shipped settings: EPZS_integer_motion_estimation lencod/src/me_epzs_int.c:42, its sub-macroblock variant :437,
                                   EPZS_sub_pel_motion_estimation me_epzs_sub.c:30 (start_qp must be 1) */
  int32_t qpc_cr_delta;         /* currMB->qpc[1] - currMB->qpc[0]: not 0 when CrQPOffset != CbQPOffset (High profiles); q_chroma[1] / q_chroma_dc[1] are the Cr tables at that QP */
  int32_t num_ref1;             /* B slices: currSlice->listXsize[LIST_1] (0 otherwise) */
  /* EPZS only (me_epzs_common.c:423 EPZSStructInit, :620 EPZSSliceInit); ignored with search_mode 0 */
  int32_t epzs_pattern, epzs_dual, epzs_fixed, epzs_aggressive, epzs_temporal, epzs_spatial_mem, epzs_blocktype;   /* EPZSPattern (0..5), EPZSDualRefinement (0..6),
                                   EPZSFixedPredictors (0..3), EPZSAggressiveWindow, EPZSTemporal, EPZSSpatialMem, EPZSBlockType */
  int32_t epzs_min_scale, epzs_med_scale, epzs_max_scale, epzs_sub_scale;   /* EPZSMinThresScale, EPZSMedThresScale, EPZSMaxThresScale, EPZSSubPelThresScale */
  int32_t b_switches;           /* B slices: bit 0 active_sps->direct_8x8_inference_flag; bit 1 BiPredMotionEstimation, bits 2..4 BiPredSearch16x16 / 16x8 / 8x16 (BiPredSearch8x8
                                   must be 0), bits 8..11 BiPredMERefinements, bits 16..23 BiPredMESearchRange, bits 24..25 BiPredMESubPel; bit 5: DirectModeType 0 (temporal direct:
                                   needs bit 0, poc_cur, poc_ref[] of list 0 and -- at [num_ref] -- of list 1's first picture, and reference pictures that the pipeline coded
                                   with non-negative ref_id: their blocks' vectors and reference picture ids are kept with the slots; Get_Direct_MV_Temporal lencod/src/mv_direct.c:40,
                                   compute_colocated mbuffer.c:3122), else spatial.  WeightedBiprediction 0; the B picture is not used for reference.
                                   Get_Direct_MV_Spatial_Normal mv_direct.c:522, the bslice branches of
                                   encode_one_macroblock_low md_low.c:174-263 / :374-431, submacroblock_mode_decision_low mode_decision_P8x8.c:681, list_prediction_cost
                                   mode_decision.c:275, BIDPartitionCost mv_search.c:1159, BiPredBlockMotionSearch :1033 */
  int32_t poc_cur;              /* enc_picture->poc */
  int32_t poc_ref[JMHIP_MB_MAX_REF];   /* listX[LIST_0][r]->poc: EPZS scales its predictors by picture distances; the temporal predictors are the vectors
                                   jmhip_reference_from_recon kept with slots ref_slot[0] / [1] (a slot loaded by jmhip_set_reference has none: zero vectors).  Every picture of
                                   an EPZS sequence -- its I pictures too -- is launched with search_mode 3 and its picture order counts: a picture launched otherwise leaves
                                   "no motion" with its slot, not JM's mv_info */
  /* High profile (ignored with transform8x8 0) */
  int32_t transform8x8;         /* p_Inp->Transform8x8Mode: 0, or 1 = the 8x8 transform beside the 4x4 one: transform_decision (macroblock.c:1347) for 16x16 / 16x8 / 8x16,
                                   the tr8x8 pass of P8x8 (mode_decision_P8x8.c:681), Intra8x8 (transform8x8.c:241), 8x8 Hadamard SATD in the sub-pel search of
                                   blocks of 8x8 samples and more (mv_search.c:1624, :1768); needs inter_valid[4] in P slices */
  int32_t intra8_valid;         /* enc_mb.valid[I8MB] (mode_decision.c:127) */
  jmhip_qparam q_luma8[2][64];  /* p_Quant->q_params_8x8[0][intra][qp][j][i] at [intra][j * 8 + i] */
  /* 4:2:2 (a context with yuv_format 2; ignored otherwise) */
  jmhip_qparam q_chroma_dc[2][2];   /* p_Quant->q_params_4x4[1 + uv][intra][qpc + 3][0][0] at [uv][intra]: the 2x4 chroma DC transform is quantised with qpc + 3 (block.c:1059-1072) */
} jmhip_slice_params;

int jmhip_encode_slice(jmhip_ctx *ctx, const jmhip_slice_params *prm, jmhip_mb_record *out /* host, num_mb records */);
/* How many workgroups (= compute units) a slice's launch may occupy; 0 = the default: a third more than the slice's widest wavefront, between 64 and 256 (80 for a 1080p slice).  A slice is a dependency chain: at
 * 1080p at most 60 macroblocks can be in flight, 27 on average, so one stream leaves most of the chip idle.  A server that encodes several
 * sequences at once (one context and one HIP stream each -- JM itself has no such mode: one lencod process per sequence) gives each context its
 * share, e.g. 32 for eight 1080p streams; the records are the same for any value (tests/test_gpu_mbenc.py). */
int jmhip_set_pipeline_workgroups(jmhip_ctx *ctx, int32_t workgroups);
int jmhip_encode_slice_dev(jmhip_ctx *ctx, const jmhip_slice_params *prm, jmhip_mb_record *d_out /* device, or NULL: kept inside only */);
/* The same, streamed, for a host that codes macroblocks in raster order while the device is still encoding (JM's write_macroblock after every
 * encode_one_macroblock): _begin launches and returns; _record waits until macroblock mb_addr's record is complete in pinned host memory and
 * returns a pointer to it (valid until the next _begin); _end waits for the launch and reports its errors.  One slice at a time. */
int jmhip_encode_slice_begin(jmhip_ctx *ctx, const jmhip_slice_params *prm);
int jmhip_slice_record(jmhip_ctx *ctx, int32_t mb_addr, const jmhip_mb_record **rec);
int jmhip_encode_slice_end(jmhip_ctx *ctx);
/* the reconstruction the slices of the current picture left on the device (before / after jmhip_deblock_picture_dev) */
int jmhip_recon_planes_dev(jmhip_ctx *ctx, uint8_t **d_y, int32_t *pitch_y, uint8_t **d_u, uint8_t **d_v, int32_t *pitch_c);
int jmhip_get_recon(jmhip_ctx *ctx, uint16_t *y, int32_t pitch_y, uint16_t *u, uint16_t *v, int32_t pitch_c);   /* as imgpel, pitches in samples */
/* The loop filter's side information the slices of the current picture left on the device: one jmhip_db_mb per macroblock (raster) and one
 * jmhip_db_motion per 4x4 block (raster over the picture's 4x4 grid, PicWidthInMbs * 4 per row).  A host that splits a picture's slices over
 * several devices exchanges these rows together with the reconstruction's before the picture is deblocked (DeblockFrame filters across slice
 * edges unless DFDisableIdc = 2: lencod/src/loopFilter.c:159-165 reads the neighbouring slice's macroblocks). */
int jmhip_deblock_side_info_dev(jmhip_ctx *ctx, jmhip_db_mb **d_mbs, jmhip_db_motion **d_motion);
/* One picture's slices dealt to several devices (SURVEY.md 8e / 8b viii: JM shards a picture only by slice, lencod/src/slice.c:431; BASELINE configs[3]): one process, n
 * contexts of the same picture size -- one per device (or several on one) --, context r has coded the slices that make up macroblock rows [r * band_mb_rows,
 * (r + 1) * band_mb_rows) of the current picture (jmhip_encode_slice[_dev / _begin]; the last band may be shorter).  The call gives EVERY context every band: the rows of
 * the un-deblocked reconstruction (Y, U, V), of the loop filter's side information (jmhip_db_mb, jmhip_db_motion) and of what jmhip_reference_from_recon keeps with the slot
 * beside the samples -- the picture's motion (EPZS's temporal predictors read the co-located block below a block: the next band's first row, me_epzs_common.c:1575-1602)
 * and its "does not move" map (a B picture's spatial direct mode) --, so that each can run jmhip_deblock_picture_dev
 * (DeblockFrame filters across slice edges unless DFDisableIdc = 2: loopFilter.c:159-165) and jmhip_reference_from_recon on the whole picture -- the next picture's
 * search windows reach into the neighbouring bands.  Peer-to-peer copies (hipMemcpyPeerAsync over xGMI between devices), each destination's copies on its own stream behind
 * an event of the source's stream; asynchronous like the launches around it.  (One process per GPU: the same exchange is one RCCL all-gather, jm_amd/shard.py BandGather.) */
int jmhip_allgather_bands(jmhip_ctx *const *ctx, int32_t n, int32_t band_mb_rows);
/* DeblockFrame on that reconstruction with the side information the slices' macroblocks left on the device */
int jmhip_deblock_picture_dev(jmhip_ctx *ctx, int32_t direct_8x8_inference);
/* getSubImagesLuma (+ the integer chroma planes) of the reconstruction into a reference slot, without leaving the device */
int jmhip_reference_from_recon(jmhip_ctx *ctx, int32_t slot);

/* ------------------------------------------------------------------------------------------
 * Consecutive pictures of ONE sequence in flight side by side
 *
 * JM codes a sequence picture after picture on one thread: code_a_picture (lencod/src/image.c:1183), then DeblockFrame over the whole picture (:236), then
 * UnifiedOneForthPix (:2187, getSubImagesLuma) when the picture enters the DPB, then the next picture.  What picture n + 1 needs of picture n is local: macroblock
 * (X, r) reads reference samples within 2 SearchRange (+ the sub-pel taps) of itself, and DeblockMb / the six-tap stencil of a macroblock are final as soon as its
 * right and lower neighbours are filtered.  So with IPPP coding, RDOptimization 0 and no rate control -- where nothing but the reference picture flows from one
 * picture to the next -- the pictures can follow each other at a distance of a few macroblock diagonals instead of a whole picture, every record and every sample
 * as JM leaves them.  (JM itself has no such mode; the hand-over is jm_amd/csrc/mbpipe_post.inc.)
 *
 * jmhip_seq_open gives the context `depth` entries (1 .. JMHIP_SEQ_MAX_DEPTH): a source picture, records, loop-filter side information and a HIP stream each.  A picture is
 *   jmhip_seq_set_frame[_dev](entry, ...)      the source picture, as jmhip_set_current_frame[_dev]
 *   jmhip_seq_encode(entry, prm, out_slot, ..) the launch: jmhip_encode_slice_dev + jmhip_deblock_picture_dev + jmhip_reference_from_recon(out_slot) in one, asynchronous;
 *                                              a reference slot (prm->ref_slot[]) that an earlier jmhip_seq_encode is still writing is followed macroblock by macroblock
 *   jmhip_seq_record(entry, mb_addr, &rec)     (to_host != 0) the record of one macroblock as soon as it is complete in pinned host memory, as jmhip_slice_record
 *   jmhip_seq_wait(entry)                      the picture is done: records (jmhip_seq_records[_dev]), the filtered reconstruction and the sub-pel planes of out_slot
 *                                              (jmhip_seq_recon_dev / jmhip_seq_get_recon, jmhip_subplanes_dev); reports the launch's errors
 * The caller deals the pictures to the entries in turn (picture k -> entry k % depth) and the reconstructions to slots that no picture in flight refers to
 * (num_ref + depth slots in turn always do; a conflicting launch is simply held back until the readers are done).  An entry's buffers -- its records among them --
 * are reused by the entry's next jmhip_seq_set_frame / jmhip_seq_encode.
 * Scope: what jmhip_encode_slice_dev accepts, the whole picture per launch: one slice, or its slices of num_mb macroblocks each (num_slices, SliceMode 1) -- coded in the
 * PICTURE's wavefront order rather than with the slices' wavefronts side by side, because a macroblock's loop filter follows its left and upper neighbours' also across a
 * slice's edge.  The full searches (search_mode 0, 1) reach 2 SearchRange into the reference, so a
 * macroblock waits once, for the macroblock 5 to its right and 5 below it (SearchRange 32); an EPZS search (search_mode 3) goes wherever its predictor takes it inside the
 * level's vector range, so every search asks for what it is about to read when its centre is known; the temporal predictors read the motion kept with the slots of
 * references 0 / 1 directly (every picture of such a sequence, its I pictures too, is launched with search_mode 3 and its picture order counts).
 * A B picture (slice_type 1, non-reference: what jmhip_encode_slice accepts) is in flight too: its launch waits -- stream order -- for the launches that make its references,
 * runs beside the pictures that follow those (a P picture does not need the B pictures before it), its loop filter follows on a stream the context keeps for B pictures, its
 * filtered reconstruction is out_slot's integer planes (jmhip_seq_get_recon / jmhip_seq_recon_dev; no sub-pel planes, no post flags: nobody reads it as a reference; give it a
 * slot that holds no reference of a picture still to be launched).  Anything else is JMHIP_EUNSUPPORTED and the caller codes that picture the usual way (the slots are shared with the other entry points: use them
 * after jmhip_seq_wait of the entry that wrote them, or after jmhip_synchronize).
 * jmhip_seq_open returns once entry 0 exists; the others (a stream and a pinned record buffer each: 10 - 12 ms) are made on a thread of the library's that starts when the first
 * launch is queued and is joined by the first call that names one of them (or by jmhip_seq_close / jmhip_destroy); JMHIP_SEQ_OPEN_INLINE=1: all inside jmhip_seq_open.
 * ------------------------------------------------------------------------------------------ */
int jmhip_seq_open(jmhip_ctx *ctx, int32_t depth, int32_t workgroups_per_picture /* 0: 256 / depth, at most 80; always cut to 248 / (depth - 1), so that the oldest picture in flight can never be
                   kept off the chip by the workgroups of later ones waiting for it.  HIP serves a process's streams from GPU_MAX_HW_QUEUES hardware queues (default 4): with
                   more pictures in flight than that, set GPU_MAX_HW_QUEUES = 2 * depth in the environment before HIP starts, or the launches take turns.
                   The count is in workgroups that fill a compute unit (eight waves).  An EPZS P picture with up to five references is launched as four-wave workgroups, two
                   to a compute unit (k_mb_pipe_epzs4*), and takes twice the count: sixteen such pictures in flight (depth 16, 2 x 16 workgroups each) are the fastest form
                   measured, 5.4 ms per 1080p picture (profiles/r04_epzs_four_wave.txt) */);
int jmhip_seq_b_workgroups(jmhip_ctx *ctx, int32_t workgroups /* workgroups of a B picture in flight (0: as the other pictures).  A B picture's macroblock takes several times a P
                   picture's and nothing waits for a B picture inside the device: a caller that keeps its B pictures in entries of their own (say 4 entries for the P pictures
                   with 16 workgroups each, 6 for the B pictures with 32) fills the chip with them */);
int jmhip_seq_close(jmhip_ctx *ctx);
int jmhip_seq_set_frame(jmhip_ctx *ctx, int32_t entry, const uint8_t *raw, int32_t src_w, int32_t src_h);
int jmhip_seq_set_frame_dev(jmhip_ctx *ctx, int32_t entry, const uint8_t *d_raw, int32_t src_w, int32_t src_h);
int jmhip_seq_set_planes(jmhip_ctx *ctx, int32_t entry, const uint16_t *y, int32_t pitch_y, const uint16_t *u, const uint16_t *v, int32_t pitch_c);   /* as jmhip_set_current_planes */
int jmhip_seq_encode(jmhip_ctx *ctx, int32_t entry, const jmhip_slice_params *prm, int32_t out_slot, int32_t direct_8x8_inference, int32_t to_host,
                     jmhip_mb_record *d_out /* device: a copy of the picture's records queued behind the launch (the entry's own are overwritten by its next picture), or NULL */);
/* with jmhip_enable_timing: milliseconds of the entry's last launch (HIP events on the entry's stream around k_mb_pipe); waits for it */
int jmhip_seq_kernel_ms(jmhip_ctx *ctx, int32_t entry, float *ms);
int jmhip_seq_record(jmhip_ctx *ctx, int32_t entry, int32_t mb_addr, const jmhip_mb_record **rec);
int jmhip_seq_wait(jmhip_ctx *ctx, int32_t entry);
int jmhip_seq_records(jmhip_ctx *ctx, int32_t entry, jmhip_mb_record *out /* host, the launch's num_mb records */);
int jmhip_seq_records_dev(jmhip_ctx *ctx, int32_t entry, jmhip_mb_record **d_records);
int jmhip_seq_recon_dev(jmhip_ctx *ctx, int32_t slot, uint8_t **d_y, int32_t *pitch_y, uint8_t **d_u, uint8_t **d_v, int32_t *pitch_c);
int jmhip_seq_get_recon(jmhip_ctx *ctx, int32_t slot, uint16_t *y, int32_t pitch_y, uint16_t *u, uint16_t *v, int32_t pitch_c);   /* as imgpel, after jmhip_seq_wait */

/* The same hand-over, n consecutive P pictures in ONE launch.  With a launch per picture (jmhip_seq_encode) every picture owns its share of the chip's compute units for as
 * long as it runs, although a picture's wavefront is narrow at both ends and wide in the middle; here the launch's persistent workgroups draw the macroblocks of all n
 * pictures from one queue, ordered by wavefront index + lag x position in the batch, so that every workgroup always finds the oldest macroblock that can run, whichever
 * picture it belongs to (lag = reach_x + 2 reach_y + 1 = 16 wavefront steps at SearchRange 32: every macroblock's queue position is behind everything it waits for -- its
 * neighbours, the macroblock 5 right / 5 below it in each reference, the pictures that still read the slot it overwrites --, which is what makes the launch deadlock-free).
 * Picture k: its source in device memory in the file's layout (as jmhip_seq_set_frame_dev), the slot its filtered reconstruction and sub-pel planes go to, its references
 * (slots an earlier picture of the batch goes to, or slots that were complete before the call: jmhip_set_reference*, jmhip_reference_from_recon, an earlier jmhip_seq_encode /
 * jmhip_seq_batch), and device memory for its PicSizeInMbs records.  prm: the parameters every picture shares (JM: one P picture after another at one QP with
 * RDOptimization 0 and no rate control differ in nothing but their references); its ref_slot / ref_id are not read.  A slot is reused inside the batch as with the entries:
 * the picture that overwrites it starts once the last picture that read it is done, so num_ref + (pictures that overlap, PicHeightInMbs / 8 at 1080p) slots keep the
 * queue dense.  Asynchronous on the context's stream: jmhip_synchronize (which reports the launch's errors), then the records, jmhip_seq_get_recon / jmhip_seq_recon_dev,
 * jmhip_subplanes_dev.  Scope: search_mode 0, 1 and 3, P slices, the whole picture (one slice or num_slices of them), num_ref <= 8, PicSizeInMbs < 65536, n <= 4096; needs jmhip_seq_open
 * (any depth).  Anything else JMHIP_EUNSUPPORTED.  Results: those of coding the pictures one after another (tests/test_gpu_seq.py).
 * EPZS (search_mode 3, P slices; round 5): a full search reads its reference no further than SearchRange from the predictor it was staged for, EPZS wherever its predictors point
 * (me_epzs_int.c:42: up to the level's vector range), so no lag covers every case.  The queue is ordered for a reach of SearchRange + 27 samples (lag 13 at SearchRange 32, 16 with several references;
 * JMHIP_EPZS_BATCH_LAG overrides) and every search asks for what IT reaches as with jmhip_seq_encode -- and waits only for a macroblock whose ticket is known to be handed
 * out (an earlier place in the queue than its own, or all tickets up to that place drawn already: such a wait always ends).  A search that reaches further sets the launch's
 * error word instead of waiting: jmhip_synchronize then returns JMHIP_EREACH, NOTHING of the launch is valid (records, slots), and the caller codes the same pictures with
 * jmhip_seq_encode, whose launches wait for each other without a queue order to respect. */
typedef struct {
  const uint8_t *d_raw;         /* device: the source picture as it lies in the file (8 bit planar) */
  int32_t src_w, src_h;         /* its size (<= the context's; padded as JM's reader pads: PadImage lencod/src/input.c:257) */
  int32_t out_slot;
  int32_t ref_slot[JMHIP_MB_MAX_REF], ref_id[JMHIP_MB_MAX_REF];   /* as jmhip_slice_params; the first prm->num_ref count */
  int32_t poc_offset;           /* EPZS: this picture's order counts are prm's poc_cur / poc_ref[] + poc_offset (IPPP: 2 (1 + FrameSkip) x its place in the batch); in what was padding */
  jmhip_mb_record *d_records;   /* device: PicSizeInMbs records */
} jmhip_seq_picture;
int jmhip_seq_batch(jmhip_ctx *ctx, const jmhip_slice_params *prm, int32_t direct_8x8_inference, int32_t n, const jmhip_seq_picture *pics);
/* EPZS launches of several pictures: the queue lag in wavefront keys (0: the library's, 3 ceil((SearchRange + 27) / 16) + 1, three more with several references).  A caller whose launch came back with JMHIP_EREACH may
 * try the same pictures once more with a larger lag -- pictures further apart reach less of each other; a picture's whole wavefront, PicWidthInMbs + 2 (PicHeightInMbs - 1), is
 * picture after picture inside one launch and never gives up -- before it falls back to a launch per picture. */
int jmhip_seq_batch_lag(jmhip_ctx *ctx, int32_t lag);
/* The device memory a launch of n pictures needs beside the slots (per picture: the source picture, edge records, flags, loop-filter side information; the launch's descriptors
 * and ticket table) is kept from launch to launch and grows with the first launch that needs more: a caller that knows its run length reserves it once, after jmhip_seq_open,
 * instead of paying a hipMalloc inside its first long launch (JM: the counterpart of init_global_buffers, lencod.c, which allocates per sequence, not per picture). */
int jmhip_seq_batch_reserve(jmhip_ctx *ctx, int32_t n);

/* ------------------------------------------------------------------------------------------
 * Timing helper: elapsed milliseconds of the last `_dev` launch of each kind, measured with
 * hipEvents on the context's stream (bench.py uses it for the roofline object).
 * kind: 0 subplanes, 1 me_fullsearch, 2 me_subpel, 3 tq, 4 deblock, 5 encode_slice.
 * ------------------------------------------------------------------------------------------ */
int jmhip_enable_timing(jmhip_ctx *ctx, int32_t on);
int jmhip_last_kernel_ms(jmhip_ctx *ctx, int32_t kind, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* JMHIP_H */
