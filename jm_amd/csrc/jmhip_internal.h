// jmhip_internal.h -- shared by the translation units of libjmhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/jmhip.h"

#define JMHIP_NKINDS 6

// A picture in flight (jmhip_seq_*, mbpipe_host.inc): everything one launch of the macroblock pipeline owns, so that launches of consecutive pictures can run side by side
#define JMHIP_SEQ_LAUNCH_EVENTS 256
#define JMHIP_SEQ_SLOT_READERS 40
struct jmhip_seq_entry {
  hipStream_t stream;
  hipEvent_t done;       // recorded behind the entry's launch (a B picture: behind its loop filter)
  hipEvent_t coded;      // a B picture in flight: behind the coding launch, before the loop filter
  hipEvent_t t0, t1;     // jmhip_enable_timing: around the launch
  int timed;
  uint8_t *d_raw;        // the source picture as the file holds it (jmhip_seq_set_frame)
  uint8_t *d_cur, *d_cur_c;   // source planes (as jmhip_ctx::d_cur / d_cur_c)
  void *d_edge; unsigned *d_done, *d_sync;
  void *d_records, *d_dbmb, *d_dbmo;
  int *d_ez_state;       // EPZS: the macroblocks' state columns of the entry's picture (allocated on first use)
  void *h_records; unsigned *h_flags; void *d_h_records; unsigned *d_h_flags;
  unsigned epoch;
  int first_mb, num_mb;  // macroblocks of the launch (streamed records)
  int out_slot, num_ref, refs[JMHIP_MB_MAX_REF];
  int in_flight;         // launched and not yet waited for (jmhip_seq_wait)
  int streaming;
  volatile int made;     // the entry exists (entries beyond the first are made by jmhip_seq_open's thread, one after the other: whoever names entry k waits for THAT one)
};

struct jmhip_ctx {
  jmhip_config cfg;
  hipStream_t stream;
  int W, H;              // luma size
  int Wp, Hp;            // padded: W + 64, H + 40
  int pitch;             // bytes per row of a padded plane (multiple of 64)
  int64_t plane_stride;  // bytes between sub-planes (multiple of 256)
  int cw, ch;            // chroma plane size
  uint8_t *d_cur;        // W x H current luma, pitch cur_pitch
  int cur_pitch;
  uint8_t *d_cur_c;      // current chroma U then V, cw x ch bytes each (allocated by jmhip_set_current_frame)
  uint8_t **d_sub;       // [num_ref_slots] -> 16 planes
  uint8_t **d_refc;      // [num_ref_slots] -> integer chroma planes U then V, cw x ch bytes each (allocated on first use)
  uint8_t *d_stage;      // staging for host uploads (W x H luma, u8)
  uint8_t *h_stage;      // pinned host staging
  size_t h_stage_bytes;
  void *d_scratch; size_t scratch_bytes;     // grows on demand: jobs/results/tables for host entry points
  void *d_scratch2; size_t scratch2_bytes;
  int16_t *d_spiral;     // [(2R+1)^2][2] spiral offsets for R = cfg.search_range
  unsigned *d_me_declined; // [0..1] jobs k_me_fs_fast left to k_me_fullsearch, ping-pong by launch parity; [4] job error flag, [5] index of the first bad job
  int me_jobs_checked;     // a job check kernel has run since the error word was last read
  unsigned me_launches;
  void *d_db_prep;       // deblocking: 192-byte strength/parameter record per macroblock (k_deblock_prep)
  unsigned *d_db_sync;   // deblocking row pipeline: ticket, error
  int db_launched;       // the row pipeline ran since its error word was last read
  void *d_db_hand;       // deblocking row pipeline: 24 8-byte hand-over granules per macroblock
  uint8_t *d_db_flags;   // deblocking segment walks: per macroblock flags (nmb bytes), then store_bottom (nmb bytes)
  void *d_db_tasks;      // deblocking segment walks: task list (1024 x int2)
  int db_no_prefill, db_sparse_pct, refine_per_block, mb_prof_mode;   // JMHIP_DEBLOCK_NO_PREFILL, JMHIP_DEBLOCK_SPARSE_PCT (default 40), JMHIP_REFINE_PER_BLOCK, JMHIP_MB_PROF: read once, in jmhip_create
  int force_db_diag;     // JMHIP_DEBLOCK_DIAG=1: one launch per diagonal instead of the row pipeline (A/B testing)
  // the macroblock pipeline (mbpipe.hip), allocated on first use
  uint8_t *d_rec;        // reconstruction of the current picture: Y (cur_pitch x H), then U, V (cw x ch, pitch cw)
  void *d_mb_edge;       // per macroblock: the samples / vectors / modes its right and lower neighbours read (136 bytes, write-through)
  unsigned *d_mb_done;   // per macroblock: epoch of the launch that finished it
  unsigned *d_mb_sync;   // [0] ticket, [1] error word, [2..9] the bands' tickets
  int *d_mb_order;       // wavefront order of the slice's macroblocks
  int mb_order_first, mb_order_num, mb_order_per, mb_order_bands;
  int mb_band_start[9];
  unsigned mb_epoch;
  int mb_launched;       // a pipeline launch has happened since its error word was last read
  int mb_alloc_done;     // mb_alloc (mbpipe_host.inc) has completed: every buffer of the pipeline exists
  int mb_grid;           // jmhip_set_pipeline_workgroups (0: the device's compute units)
  int num_cus;           // hipDeviceProp_t::multiProcessorCount (256 on a whole MI355X; a partitioned one -- CPX / DPX -- has fewer): the persistent launches are sized by it
  void *d_mb_records;    // jmhip_mb_record per macroblock of the picture
  void *d_mb_dbmb, *d_mb_dbmo;   // loop-filter side information written by the pipeline
  void *h_mb_records;    // pinned host staging for jmhip_encode_slice
  unsigned *h_mb_flags;  // pinned, device-visible: per macroblock the epoch whose record is complete in h_mb_records (streaming)
  void *d_h_mb_records; unsigned *d_h_mb_flags;   // the device's addresses of the two
  int mb_streaming, mb_stream_first, mb_stream_num;
  void *d_mb_prof;       // JMHIP_MB_PROF=1: time stamps per macroblock
  // EPZS inside the pipeline (search_mode 3), allocated on its first use
  int *d_ez_state;       // per macroblock: its columns of p_EPZS->distortion / p_motion (28 + 112 x 16 ints)
  int *d_ez_col;         // co-located vectors of the running launch, per 4x4 block
  void *d_mot;           // the current picture's motion per 4x4 block {packed vector, poc referred to}: what jmhip_reference_from_recon keeps with the slot
  void **d_slot_mot;     // [num_ref_slots] the same of the pictures in the slots ({0, none} after jmhip_set_reference*)
  uint8_t *d_colz;       // the current picture's answer, per 4x4 block, to a B picture's spatial direct mode ("the co-located block does not move": PicView::colz_out)
  uint64_t *d_colm, **d_slot_colm;   // ... and the blocks' vectors | reference picture ids << 32 (temporal direct), the current picture's and the slots'
  uint8_t **d_slot_colz; // [num_ref_slots] the same of the pictures in the slots (jmhip_reference_from_recon keeps it with the slot; pictures in flight write theirs there)
  // pictures in flight (jmhip_seq_open)
  int seq_depth, seq_grid;
  jmhip_seq_entry *seq;
  uint8_t **d_slot_recy; // [num_ref_slots] the slot's picture before interpolation: luma, pitch cur_pitch (its chroma is d_refc[slot]); filtered in place
  unsigned **d_slot_post;// [num_ref_slots] per macroblock: slot_tag once the macroblock is filtered and interpolated
  unsigned *slot_tag;    // [num_ref_slots] tag of the slot's current / last picture made by a sequence launch
  int *slot_entry;       // [num_ref_slots] entry that makes / made the slot's picture, -1: filled by jmhip_set_reference* / jmhip_reference_from_recon
  // who still reads or writes a slot, launch by launch (an entry's `done` event is re-recorded by the entry's next launch: with B pictures in flight the host is entries
  // ahead of the device, and waiting for "the entry" would be waiting for the wrong picture): every sequence launch takes an event of its own from a pool
  hipEvent_t *launch_ev; int launch_ev_next;                    // JMHIP_SEQ_LAUNCH_EVENTS of them, taken in turn (one is reused after that many launches: complete by then, or waited for)
  hipEvent_t *slot_wev;                                         // [num_ref_slots] the launch that writes / wrote the slot's picture (null: none)
  hipEvent_t *slot_rev; int *slot_nrev;                         // [num_ref_slots][JMHIP_SEQ_SLOT_READERS] launches reading the slot's picture since; their number (-1: more than fit -- wait for everything)
  hipEvent_t seq_ev;     // orders the context's own stream before an entry's
  int seq_b_grid;        // workgroups of a B picture in flight (0: seq_grid)
  void *seq_maker; int seq_maker_rc; volatile int seq_maker_go, seq_maker_done;      // jmhip_seq_open: the std::thread that makes entries 1 .. depth - 1 while the first picture is coded (joined by whoever names such an entry), its result
  hipStream_t bdb_stream; hipEvent_t bdb_ev; int bdb_used;     // B pictures in flight: the stream their loop filters run on one after the other, and the last one's event
  // several pictures in ONE launch (jmhip_seq_batch): per picture a source picture, edge records, flags and loop-filter side information at a fixed stride; the
  // pictures' descriptors and the ticket order of the last batch
  uint8_t *d_batch; int batch_cap; void *d_batch_tab; size_t batch_tab_bytes; unsigned batch_epoch;
  int batch_lag;         // jmhip_seq_batch_lag: the EPZS launches' queue lag (0: the library's)
  int *d_batch_ez; size_t batch_ez_bytes;      // jmhip_seq_batch with EPZS: the pictures' PicView::ez_state
  int timing;
  int force_generic;     // JMHIP_FORCE_GENERIC=1: never use the tuned ME kernel (A/B testing)
  hipEvent_t ev0[JMHIP_NKINDS], ev1[JMHIP_NKINDS];
  int ev_valid[JMHIP_NKINDS];
  char err[512];
};

extern char g_jmhip_create_err[512];

int jmhip_fail(jmhip_ctx *ctx, int code, const char *fmt, ...);
// a process may hold contexts on several devices (jmhip_allgather_bands): the entry points of the macroblock pipeline select theirs first
#define JMHIP_DEVICE(ctx) do { if ((ctx)->cfg.device != 0 || g_jmhip_multi_device) (void)hipSetDevice((ctx)->cfg.device); } while (0)
extern int g_jmhip_multi_device;
#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return jmhip_fail(ctx, JMHIP_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

int jmhip_scratch(jmhip_ctx *ctx, int which, size_t bytes, void **out);
void jmhip_time_begin(jmhip_ctx *ctx, int kind);
void jmhip_time_end(jmhip_ctx *ctx, int kind);

// kernels' launchers (defined next to the kernels)
int jmhip_check_deblock_error(jmhip_ctx *ctx);
int jmhip_check_job_error(jmhip_ctx *ctx);      // me_fullsearch.hip: device-side validation of job records
void jmhip_launch_check_me_jobs(jmhip_ctx *ctx, const jmhip_me_job *d_jobs, int n);
void jmhip_launch_check_subpel_jobs(jmhip_ctx *ctx, const jmhip_subpel_job *d_jobs, int n);
void jmhip_mb_free(jmhip_ctx *ctx);   // mbpipe.hip
int jmhip_seq_sync_all(jmhip_ctx *ctx);   // mbpipe.hip: every picture in flight is done (their error words are left for jmhip_seq_wait)
int jmhip_check_mb_error(jmhip_ctx *ctx);   // mbpipe.hip: the pipeline's sticky error word, read and cleared
void jmhip_mb_slot_motion_reset(jmhip_ctx *ctx, int slot);   // mbpipe.hip: a slot loaded from outside the pipeline carries no motion
int jmhip_launch_subplanes(jmhip_ctx *ctx, const uint8_t *d_luma, int pitch, uint8_t *d_planes);
int jmhip_launch_deblock_rows(jmhip_ctx *ctx, uint8_t *d_Y, int pitchY, uint8_t *d_U, uint8_t *d_V, int pitchC,
                              const jmhip_db_mb *d_mbs, const jmhip_db_motion *d_motion, int direct8x8);
void jmhip_launch_refine_mb(jmhip_ctx *ctx, int slot, const jmhip_me_job *d_jobs, int njobs, const jmhip_me_result *d_int,
                            const jmhip_refine_params *prm, jmhip_me_result *d_out);
