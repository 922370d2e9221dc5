/*
 * jm_adapter.c -- host-side C glue that puts libjmhip (include/jmhip.h) behind JM 19.0 lencod's own call surface.
 *
 * JM is plain C and so is this file: it is compiled against JM's public headers and linked into the UNMODIFIED
 * reference encoder objects with GNU ld's --wrap, so that every call the encoder makes to one of the hot-path
 * functions below lands here, is served by the MI355X through the C ABI, and returns to JM's sequential control
 * flow (mode decision, MV prediction, entropy coding) with exactly the values JM's own function would have
 * produced.  The encoder.cfg / YUV-in -> Annex-B .264-out contract is untouched; the bitstream is bit-identical
 * (tests/test_lencod_dropin.py compares md5s with CPU JM).
 *
 *   JM function (reference file:line)                              bound through                      served by
 *   getSubImagesLuma            lencod/src/img_luma.c:611          UnifiedOneForthPix image.c:2187     jmhip_set_reference + jmhip_get_subplanes
 *   full_search_motion_estimation lencod/src/me_fullsearch.c:39    Macroblock.IntPelME  mv_search.c:139-175   jmhip_me_fullsearch (one window job)
 *   sub_pel_motion_estimation   lencod/src/me_fullsearch.c:186     Macroblock.SubPelME                 jmhip_me_subpel
 *   setup_fast_full_search      lencod/src/me_fullfast.c:269       Macroblock.p_SetupFastFullPelSearch jmhip_me_sad_tables (BlockSAD tables; JM keeps its argmin)
 *   residual_transform_quant_luma_4x4 lencod/src/block.c:661       Macroblock.residual_transform_quant_luma_4x4   jmhip_tq_luma4x4 ("dct_4x4" + quant_4x4)
 *   residual_transform_quant_luma_8x8 (+_cavlc) lencod/src/transform8x8.c:522 / :604   Macroblock.residual_transform_quant_luma_8x8   jmhip_tq_luma8x8
 *   residual_transform_quant_chroma_4x4 lencod/src/block.c:954     Macroblock.residual_transform_quant_chroma_4x4[uv]   jmhip_tq_chroma
 *   DeblockFrame                lencod/src/loopFilter.c:63         image.c:236                         jmhip_deblock_frame
 *   encode_one_slice            lencod/src/slice.c:431             image.c:210                         (hook only: uploads the current picture, then calls JM's own)
 *
 * A call whose configuration the device path does not implement (weighted prediction, chroma ME, SSE metric,
 * RDOptimization=0's (0,0) bonus, field/MBAFF pictures, 4:4:4, bit depth > 8, search range > 64) is passed to JM's own
 * function (__real_*) and counted; the counters are printed at exit.  JMHIP_ADAPTER=off passes everything through;
 * JMHIP_ADAPTER_PARTS=interp,fs,subpel,ffs,tq4,tq8,tqc,deblock selects a subset.  There is no CPU restatement in here: either
 * the GPU serves a call or JM's own code does.
 *
 * This is per-call, synchronous offload: it demonstrates the drop-in boundary and bit-exactness inside the real
 * encoder.  Throughput comes from the batched entry points (bench.py); see INTEGRATION.md for how a maintainer
 * batches macroblock rows of ME behind the same slots.
 */
#define _GNU_SOURCE                                         /* fopencookie */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/types.h>
#include <unistd.h>
#include "global.h"
#include "image.h"
#include "mbuffer.h"
#include "mb_access.h"
#include "mv_search.h"
#include "me_distortion.h"
#include "me_fullsearch.h"
#include "me_fullfast.h"
#include "loop_filter.h"
#include "img_luma.h"
#include "block.h"
#include "transform8x8.h"
#include "quant4x4.h"
#include "quant8x8.h"
#include "slice.h"
#include "intra16x16.h"
#include "intra4x4.h"
#include "intra8x8.h"
#include "me_epzs_common.h"
#include "conformance.h"
#include "jmhip.h"

#define MAX_SLOTS 32

static struct {
  int         init_done, off;
  int         part_interp, part_fs, part_subpel, part_ffs, part_deblock, part_tq4, part_tq8, part_tqc, part_tq16, part_mcl, part_mcc, part_eval, part_ip4, part_i16, part_interpc, part_evalp, part_load, part_ic, part_ip8, part_mbpipe, part_evalbatch;
  int         in_real_me;             /* inside one of JM's own search functions: its computeSAD / computeSATD calls stay on the host */
  jmhip_ctx  *ctx;
  jmhip_ctx  *ctxs[8];                /* JMHIP_DEVICES=d0,d1,...: one context per entry (ctxs[0] = ctx); a SliceMode 1 picture's slices are dealt to them */
  int         nctx;
  int         W, H, fmt, R, nslots;
  StorablePicture *slot_pic[MAX_SLOTS];
  unsigned    slot_tick[MAX_SLOTS], tick;
  int         slot_chroma[MAX_SLOTS]; /* the slot's integer chroma planes are on the device (jmhip_set_reference_chroma) */
  uint16_t   *planes;                 /* 16 x (H+40) x (W+64) imgpel, jmhip_get_subplanes output */
  uint16_t   *cplanes;                /* chroma sub-images of one plane, jmhip_get_chroma_subplanes output (allocated on first use) */
  uint16_t   *tables;                 /* 7 x 16 x max_pos uint16, jmhip_me_sad_tables output */
  jmhip_db_mb     *dbmb;
  jmhip_db_motion *dbmo;
  long        n_interp, n_fs, n_subpel, n_ffs, n_deblock, n_cur, n_passed, n_tq4, n_tq8, n_tqc, n_tq16, n_mcl, n_mcc, n_eval, n_ip4, n_i16, n_interpc, n_evalp, n_load, n_ic, n_ip8, n_eval_hits, n_eval_calls, n_eval_cands;
} G;

static void pipe_report(void);
static double now_s(void);
/* JMHIP_INIT_PROF=1: where a run's first hundreds of milliseconds go (profiles/r05_init_prof.sh) -- stamps in ms since the process started */
static double IP_t0;
static int IP_on;
static void init_prof(const char *what) { if (IP_on) fprintf(stderr, "jmhip adapter: init %8.1f ms  %s\n", 1e3 * (now_s() - IP_t0), what); }
__attribute__((constructor)) static void init_prof_start(void) { IP_t0 = now_s(); IP_on = getenv("JMHIP_INIT_PROF") != NULL; }
static double T_slice, T_pad, T_deblock, T_interp;
/* per picture (the first 64): when encode_one_slice was entered and left, and the time inside DeblockFrame / getSubImagesLuma (JMHIP_ADAPTER_TIMELINE=1 prints them) */
static double TL_in[64], TL_out[64], TL_db[64], TL_ip[64], TL_begun[64], TL_first[64], TL_last[64], TL_ended[64], TL_wait[64], TL_db_at[64], TL_ip_at[64];
static int TL_n;
static int pipe_deblock(VideoParameters *p_Vid, imgpel **imgY, imgpel ***imgUV);
static int pipe_reference(StorablePicture *s);
static int pipe_config_ok(VideoParameters *p_Vid);
static int pipe_active(void);
static void flight_begin(VideoParameters *p_Vid);
static void flight_launch(VideoParameters *p_Vid, const jmhip_slice_params *prm);

/* Pictures in flight (jmhip_seq_*).  JM codes picture after picture; with IPPP, RDOptimization 0 and no rate control nothing but the reference picture
 * flows from one picture to the next, and everything else a picture's launch needs is known beforehand: its source frame lies in the input file, its parameters are those
 * of the last picture of its type.  So while JM entropy-codes picture k the adapter has the next pictures launched already -- each follows its reference a few macroblock
 * diagonals behind, inside the device (mbpipe_post.inc) -- and when JM arrives at picture k + 1 its records are waiting.  Nothing is taken on trust: at the picture's
 * first macroblock the parameters JM really has are built as always and compared byte for byte with the ones the launch was given, and the source planes JM really holds
 * are compared with the frame that was sent; any difference voids the pictures in flight and the picture is launched the ordinary way. */
#define FL_MAX 16
static struct {
  int depth, nring;                   /* entries of jmhip_seq_open; device slots the reference pictures (I, P) use in turn */
  int nb, nring_b;                    /* NumberBFrames; the slots behind the first nring that the (non-reference) B pictures use in turn */
  long refcount, bcount;              /* reference / B pictures met so far */
  int st_n, st_slot[JMHIP_MB_MAX_REF + 1], window;   /* the stored reference pictures as the sliding window holds them, most recent first: slot and ... */
  long st_disp[JMHIP_MB_MAX_REF + 1]; /* ... place in display order */
  int poc0;                           /* picture order count of picture 0 */
  long first_frame_no; int regular;   /* the file's frame of picture 0; pictures so far arrived in the regular order I P B.. P B.. (display place fl_disp) */
  int on;
  int fd, have_file, src_w, src_h, cur_frame_no, frame_step;
  long header, frame_bytes, start_frame;
  long pic, total;                    /* pictures of the sequence met so far; pictures it will have */
  struct { long pic; int valid, slot; jmhip_slice_params prm; uint8_t *raw; } sub[FL_MAX];      /* by entry: the launch it holds */
  jmhip_slice_params tmpl[3];         /* by slice type: what the last picture of that type was launched with */
  int have_tmpl[3], nref_prev;
  int slot_poc[MAX_SLOTS], poc_last, poc_step;   /* EPZS: the picture order count of the picture in each slot of the ring; the last picture's and the step to the one before it */
  int cur_entry, cur_slot;
  long n_ahead, n_hit, n_void;
} F;

static void adapter_report(void)
{
  if (!G.init_done) return;
  pipe_report();
  if (G.n_eval) fprintf(stderr, "jmhip adapter: candidate distortions: %ld requests served by %ld jmhip_me_eval calls (%ld candidates evaluated, %ld requests answered from the cache)\n", G.n_eval, G.n_eval_calls, G.n_eval_cands, G.n_eval_hits);
  fprintf(stderr, "jmhip adapter: on the MI355X: %ld getSubImagesLuma, %ld full_search_motion_estimation, %ld sub_pel_motion_estimation, "
                  "%ld setup_fast_full_search, %ld DeblockFrame (%ld current pictures uploaded); passed to JM's own code: %ld calls; "
                  "transform/quant blocks on the MI355X: %ld 4x4, %ld 8x8, %ld chroma planes; prediction blocks on the MI355X: %ld luma, %ld chroma; "
                  "Intra16x16 macroblocks on the MI355X: %ld; candidate distortions (computeSAD / computeSATD) on the MI355X: %ld; "
                  "intra predictions on the MI355X: %ld 4x4 blocks, %ld Intra16x16 mode searches; getSubImagesChroma on the MI355X: %ld; "
                  "weighted / bi-predictive candidate distortions on the MI355X: %ld; source pictures padded on the MI355X: %ld; chroma intra predictions on the MI355X: %ld macroblocks; Intra8x8 predictions on the MI355X: %ld blocks\n",
          G.n_interp, G.n_fs, G.n_subpel, G.n_ffs, G.n_deblock, G.n_cur, G.n_passed, G.n_tq4, G.n_tq8, G.n_tqc, G.n_mcl, G.n_mcc, G.n_tq16, G.n_eval,
          G.n_ip4, G.n_i16, G.n_interpc, G.n_evalp, G.n_load, G.n_ic, G.n_ip8);
  init_prof("exit: the report is written");
}

/* The process is over: JM has closed its files, the report above is written.  What would follow -- jmhip_destroy (33 ms at 1080p: pinned buffers, streams) and then the HIP
 * runtime's own exit handlers (139 ms) -- frees what the kernel driver reclaims from a dead process anyway, and is more than a quarter of a two-picture run's wall time
 * (profiles/r05_init_prof.txt).  A caller that wants that time back asks for it: with JMHIP_ADAPTER_FAST_EXIT=1, after a NORMAL end (status 0) with nothing in flight, every
 * stdio stream is flushed and the process leaves with _exit -- exit handlers registered before the adapter's (other libraries', coverage or sanitizer reports, the HIP runtime's
 * own shutdown) then never run, which is why it is not the default.  The full teardown is what happens otherwise, and always under a profiler that collects
 * at exit (rocprofv3: ROCP_TOOL_LIBRARIES / LD_PRELOAD), in the -pg build (gmon.out is written by an exit handler) and after any failure. */
static void adapter_exit(int status, void *unused)
{
  int k_, fast;
  (void)unused;
  if (!G.init_done) return;
  adapter_report();
#ifdef JMHIP_ADAPTER_KEEP_EXIT_HANDLERS
  fast = 0;
#else
  { const char *fe = getenv("JMHIP_ADAPTER_FAST_EXIT"); fast = status == 0 && !G.off && fe && fe[0] == '1' && !getenv("JMHIP_ADAPTER_FULL_EXIT") && !getenv("ROCP_TOOL_LIBRARIES") && !getenv("LD_PRELOAD"); }
#endif
  if (fast) {
    for (k_ = 0; k_ < G.nctx; k_++) if (G.ctxs[k_] && jmhip_synchronize(G.ctxs[k_]) != JMHIP_OK) fast = 0;      /* nothing may be left running on the device */
    if (fast) { init_prof("exit: leaving without the teardown"); fflush(NULL); _exit(0); }
  }
  for (k_ = 1; k_ < G.nctx; k_++) if (G.ctxs[k_]) jmhip_destroy(G.ctxs[k_]);
  if (G.ctx) jmhip_destroy(G.ctx);
  init_prof("exit: jmhip_destroy done");
  G.ctx = NULL;
}

static void adapter_die(const char *what, int rc)
{
  fprintf(stderr, "jmhip adapter: %s failed (%d): %s\n", what, rc, jmhip_last_error(G.ctx));
  exit(70);                                              /* no silent CPU fallback for a call the device accepted */
}

/* is `name` one of the comma-separated tokens of `list` (NULL list = everything) */
/* ------------------------------------------------------------------ the host's own per-picture chores, once the macroblocks come from the device
 * (profiles/host_gprof.sh): none of them changes a byte JM writes.
 *
 * part "nulltrace": the reference is built with TRACE on (lencod/inc/defines.h:25): per syntax element trace2out (vlc.c:1364-1398) formats a line with
 * some seventy stdio calls and an fflush -- 80 000 times per 1080p picture, which is most of the host's time per macroblock once the decisions come
 * from the device.  When the configuration sends the trace to /dev/null, the trace FILE is replaced by a sink inside the process and the stdio calls
 * JM's objects make ON THAT FILE (fprintf, putc, fputc, fflush, bound with --wrap like everything else here) return at once: the bytes were going
 * nowhere.  Every other FILE, and any other TraceFile, goes to libc untouched. */
#include <stdarg.h>
static FILE *null_sink;
static ssize_t null_write(void *cookie, const char *buf, size_t size) { (void)cookie; (void)buf; return (ssize_t)size; }
static void trace_to_null_sink(InputParameters *p_Inp)
{
  cookie_io_functions_t io;
  FILE *f;
  if (!p_Enc || !p_Enc->p_trace || strcmp(p_Inp->TraceFile, "/dev/null")) return;
  memset(&io, 0, sizeof io);
  io.write = null_write;
  if (!(f = fopencookie(NULL, "w", io))) return;
  fclose(p_Enc->p_trace);
  p_Enc->p_trace = f;
  null_sink = f;
}
extern int __real_putc(int, FILE *);
extern int __real_fputc(int, FILE *);
extern int __real_fflush(FILE *);
extern int __vfprintf_chk(FILE *, int, const char *, va_list);
#define IS_SINK(f) (null_sink && (f) == null_sink)            /* a NULL FILE (or any other) reaches libc while no sink exists */
int __wrap_putc(int c, FILE *f) { return IS_SINK(f) ? c : __real_putc(c, f); }
int __wrap_fputc(int c, FILE *f) { return IS_SINK(f) ? c : __real_fputc(c, f); }
int __wrap_fflush(FILE *f) { return IS_SINK(f) ? 0 : __real_fflush(f); }
int __wrap_fprintf(FILE *f, const char *fmt, ...)             /* a build without _FORTIFY_SOURCE calls fprintf itself */
{
  va_list ap;
  int r;
  if (IS_SINK(f)) return 0;
  va_start(ap, fmt);
  r = vfprintf(f, fmt, ap);
  va_end(ap);
  return r;
}
int __wrap___fprintf_chk(FILE *f, int flag, const char *fmt, ...)
{
  va_list ap;
  int r;
  if (IS_SINK(f)) return 0;
  va_start(ap, fmt);
  r = __vfprintf_chk(f, flag, fmt, ap);
  va_end(ap);
  return r;
}
/* part "lazyzero": JM's padded picture planes (get_mem2Dpel_pad .. get_mem5Dpel_pad, lcommon/src/memalloc.c:881-1022) are malloc + memset (mem_calloc, memalloc.h:204).
 * alloc_storable_picture (mbuffer.c:438-456) makes 16 luma and 128 chroma sub-image planes that way for EVERY stored picture -- 0.84 GB of memset and a quarter of a million
 * page faults per 2160p picture (0.21 GB at 1080p) -- and with the macroblock pipeline nobody on the host ever reads them (the device keeps its own).  The same
 * pointer structures with calloc for the samples: the memory is as zero as JM makes it, but a plane nobody touches costs an mmap and a munmap instead of being written
 * (M_MMAP_THRESHOLD is pinned so that every plane of a megabyte or more is mapped afresh).  JM's own free_mem*_pad free them: same layout, same allocator. */
#include <malloc.h>
static int lz_on(void)
{
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("JMHIP_ADAPTER"), *parts = getenv("JMHIP_ADAPTER_PARTS"), *p;
    on = !(e && !strcmp(e, "off"));
    if (on && parts) { on = 0; for (p = parts; (p = strstr(p, "lazyzero")); p += 8) if ((p == parts || p[-1] == ',') && (p[8] == 0 || p[8] == ',')) on = 1; }
    if (on) mallopt(M_MMAP_THRESHOLD, 1 << 20);
  }
  return on;
}
static void lz_nomem(void) { fprintf(stderr, "jmhip adapter: out of memory (picture planes)\n"); exit(70); }
static int lz_2d(imgpel ***a, int d0, int d1, int py, int px)
{
  const int h = d0 + 2 * py, w = d1 + 2 * px;
  imgpel *cur;
  int i;
  if (!(*a = (imgpel **)malloc((size_t)h * sizeof(imgpel *)))) lz_nomem();
  if (!((*a)[0] = (imgpel *)calloc((size_t)h * w, sizeof(imgpel)))) lz_nomem();
  (*a)[0] += px;
  cur = (*a)[0];
  for (i = 1; i < h; i++) { cur += w; (*a)[i] = cur; }
  *a = &(*a)[py];
  return h * (int)(sizeof(imgpel *) + w * sizeof(imgpel));
}
static int lz_3d(imgpel ****a, int d0, int d1, int d2, int py, int px)
{
  int i, n = d0 * (int)sizeof(imgpel **);
  if (!(*a = (imgpel ***)malloc((size_t)d0 * sizeof(imgpel **)))) lz_nomem();
  for (i = 0; i < d0; i++) n += lz_2d((*a) + i, d1, d2, py, px);
  return n;
}
static int lz_4d(imgpel *****a, int d0, int d1, int d2, int d3, int py, int px)
{
  int i, n = d0 * (int)sizeof(imgpel ***);
  if (!(*a = (imgpel ****)malloc((size_t)d0 * sizeof(imgpel ***)))) lz_nomem();
  n += lz_3d(*a, d0 * d1, d2, d3, py, px);
  for (i = 1; i < d0; i++) (*a)[i] = (*a)[i - 1] + d1;
  return n;
}
extern int __real_get_mem2Dpel_pad(imgpel ***, int, int, int, int);
extern int __real_get_mem3Dpel_pad(imgpel ****, int, int, int, int, int);
extern int __real_get_mem4Dpel_pad(imgpel *****, int, int, int, int, int, int);
extern int __real_get_mem5Dpel_pad(imgpel ******, int, int, int, int, int, int, int);
int __wrap_get_mem2Dpel_pad(imgpel ***a, int d0, int d1, int py, int px) { return lz_on() ? lz_2d(a, d0, d1, py, px) : __real_get_mem2Dpel_pad(a, d0, d1, py, px); }
int __wrap_get_mem3Dpel_pad(imgpel ****a, int d0, int d1, int d2, int py, int px) { return lz_on() ? lz_3d(a, d0, d1, d2, py, px) : __real_get_mem3Dpel_pad(a, d0, d1, d2, py, px); }
int __wrap_get_mem4Dpel_pad(imgpel *****a, int d0, int d1, int d2, int d3, int py, int px) { return lz_on() ? lz_4d(a, d0, d1, d2, d3, py, px) : __real_get_mem4Dpel_pad(a, d0, d1, d2, d3, py, px); }
int __wrap_get_mem5Dpel_pad(imgpel ******a, int d0, int d1, int d2, int d3, int d4, int py, int px)
{
  int i, n = d0 * (int)sizeof(imgpel ****);
  if (!lz_on()) return __real_get_mem5Dpel_pad(a, d0, d1, d2, d3, d4, py, px);
  if (!(*a = (imgpel *****)malloc((size_t)d0 * sizeof(imgpel ****)))) lz_nomem();
  n += lz_4d(*a, d0 * d1, d2, d3, d4, py, px);
  for (i = 1; i < d0; i++) (*a)[i] = (*a)[i - 1] + d1;
  return n;
}

/* part "readframe": buf2img_basic (lcommon/src/input.c:552) widens the file's bytes to imgpel one memcpy at a time (12 ms per 1080p picture); the same
 * assignment as a plain loop the compiler vectorises, for the case JM's first branch of the 8-bit path covers (equal sizes); everything else is JM's. */
extern void buf2img_basic(imgpel **imgX, unsigned char *buf, int size_x, int size_y, int o_size_x, int o_size_y, int symbol_size_in_bytes, int bitshift);
static void buf2img_bytes(imgpel **imgX, unsigned char *buf, int size_x, int size_y, int o_size_x, int o_size_y, int symbol_size_in_bytes, int bitshift)
{
  int i, j;
  if (symbol_size_in_bytes != 1 || sizeof(imgpel) != 2 || size_x != o_size_x || size_y != o_size_y) {
    buf2img_basic(imgX, buf, size_x, size_y, o_size_x, o_size_y, symbol_size_in_bytes, bitshift);
    return;
  }
  for (j = 0; j < o_size_y; j++) {
    imgpel *restrict d = imgX[j];
    const unsigned char *restrict q = buf + (size_t)j * size_x;
    for (i = 0; i < o_size_x; i++) d[i] = (imgpel)q[i];
  }
}

static int has_part(const char *list, const char *name)
{
  const size_t n = strlen(name);
  const char *p = list;
  if (!list) return 1;
  while (*p) {
    const char *e = strchr(p, ',');
    const size_t len = e ? (size_t)(e - p) : strlen(p);
    if (len == n && !strncmp(p, name, n)) return 1;
    if (!e) break;
    p = e + 1;
  }
  return 0;
}

/* Create the context on first use; returns 0 when the device path cannot serve this sequence at all. */
static int adapter_on(VideoParameters *p_Vid)
{
  if (!G.init_done) {
    InputParameters *p_Inp = p_Vid->p_Inp;
    const char *e = getenv("JMHIP_ADAPTER"), *parts = getenv("JMHIP_ADAPTER_PARTS");
    jmhip_config cfg;
    int rc;
    G.init_done = 1;
    on_exit(adapter_exit, NULL);
    G.part_interp = has_part(parts, "interp"); G.part_fs = has_part(parts, "fs"); G.part_subpel = has_part(parts, "subpel");
    G.part_ffs = has_part(parts, "ffs"); G.part_deblock = has_part(parts, "deblock");
    G.part_tq4 = has_part(parts, "tq4"); G.part_tq8 = has_part(parts, "tq8"); G.part_tqc = has_part(parts, "tqc");
    G.part_mcl = has_part(parts, "mcl"); G.part_mcc = has_part(parts, "mcc"); G.part_tq16 = has_part(parts, "tq16");
    G.part_eval = has_part(parts, "eval"); G.part_ip4 = has_part(parts, "ip4"); G.part_i16 = has_part(parts, "i16");
    G.part_interpc = has_part(parts, "interpc"); G.part_evalp = has_part(parts, "evalp"); G.part_load = has_part(parts, "load"); G.part_ic = has_part(parts, "ic"); G.part_ip8 = has_part(parts, "ip8"); G.part_mbpipe = has_part(parts, "mbpipe"); G.part_evalbatch = has_part(parts, "evalbatch");
    if (e && !strcmp(e, "off")) { G.off = 1; return 0; }
    if (p_Vid->bitdepth_luma != 8 || p_Vid->bitdepth_chroma != 8 || p_Vid->yuv_format > YUV422 ||
        p_Inp->PicInterlace != FRAME_CODING || p_Inp->MbInterlace != FRAME_CODING) {
      fprintf(stderr, "jmhip adapter: configuration outside the device path (bit depth / 4:4:4 / interlace): JM's own code runs\n");
      G.off = 1; return 0;
    }
    G.W = p_Vid->width; G.H = p_Vid->height; G.fmt = p_Vid->yuv_format;
    G.R = imax(p_Inp->search_range[0], p_Inp->search_range[1]);
    if (G.R > JMHIP_MAX_SEARCH_RANGE) { G.part_fs = G.part_ffs = 0; G.R = JMHIP_MAX_SEARCH_RANGE; }
    if (G.R < 1) G.R = 1;
    G.nslots = imin(MAX_SLOTS, p_Vid->max_num_references + 2);
    {                                                       /* pictures in flight: JMHIP_ADAPTER_FLIGHT = 2 .. 16 entries (0 or 1: off).  Default 4 for the full searches -- the host's
                                                               entropy coder is what bounds such a sequence, and the first pictures of a run get more workgroups each -- and 16 for EPZS with
                                                               its up to five references per picture, where the device is: its P pictures run as four-wave workgroups, two to a compute unit, and
                                                               sixteen of them fill the chip (24 1080p pictures of configs[2]: 16 - 58 ms per later P picture with eight in flight, the device
                                                               running dry again and again; 17 - 23 ms with sixteen: profiles/r04_e2e_configs2_flight.txt) */
      const char *fl = getenv("JMHIP_ADAPTER_FLIGHT");
      F.depth = fl ? atoi(fl) : (p_Inp->SearchMode[0] == EPZS ? 16 : p_Inp->NumberBFrames ? 8 : 4);       /* (B pictures: ~6 x a P picture's time on the device, and nobody waits for them) */
      /* (SliceMode 1: the picture's slices go up in one launch either way; with pictures in flight the launch runs in the picture's wavefront order -- not with JMHIP_DEVICES,
         where the slices are dealt to several contexts) */
      if (F.depth < 2 || !G.part_mbpipe || p_Inp->rdopt != 0 || (p_Inp->slice_mode != NO_SLICES && (p_Inp->slice_mode != FIXED_MB || getenv("JMHIP_DEVICES"))) ||
          (p_Inp->NumberBFrames != 0 && (p_Inp->intra_period != 0 || p_Inp->idr_period != 0 || p_Inp->SearchMode[0] == EPZS)) ||
          (p_Inp->SearchMode[0] != FULL_SEARCH && p_Inp->SearchMode[0] != FAST_FULL_SEARCH && p_Inp->SearchMode[0] != EPZS)) F.depth = 0;
      if (F.depth > FL_MAX) F.depth = FL_MAX;
      /* a run of n pictures never has more than n - 1 of them in flight behind its first: entries beyond that only cut every picture's share of the workgroups (jmhip_seq_open
         deals 248 / (depth - 1) at most) -- configs[2]'s three-picture run: its first P picture alone on 2 x 16 four-wave workgroups 93 ms, 75 with the chip to itself
         (profiles/r06_first_pictures.txt) */
      if (!fl && F.depth > 2 && p_Inp->no_frames > 0 && F.depth > imax(2, p_Inp->no_frames - 1)) F.depth = imax(2, p_Inp->no_frames - 1);
      if (F.depth) {
        F.nb = p_Inp->NumberBFrames;
        if (F.nb && F.depth > 10) F.depth = 10;
        F.nring_b = F.nb ? F.depth + 1 : 0;                  /* a B picture's filtered reconstruction waits in its slot until JM has fetched it */
        F.nring = imin(MAX_SLOTS - F.nring_b, p_Vid->max_num_references + F.depth + 1);
        G.nslots = F.nring + F.nring_b;
        setenv("GPU_MAX_HW_QUEUES", F.depth > 8 ? "24" : "16", 0);       /* before HIP starts: a hardware queue per picture in flight (the runtime's default is 4) */
      }
    }
    memset(&cfg, 0, sizeof cfg);
    cfg.device = getenv("JMHIP_DEVICE") ? atoi(getenv("JMHIP_DEVICE")) : 0;
    cfg.width = G.W; cfg.height = G.H; cfg.yuv_format = G.fmt; cfg.bit_depth = 8; cfg.search_range = G.R; cfg.num_ref_slots = G.nslots;
    {
      /* JMHIP_DEVICES=0,1,...: a context per listed device (a device may be listed more than once); used when a picture comes in slices (SliceMode 1) */
      const char *dl = getenv("JMHIP_DEVICES");
      int devs[8], nd = 0, k;
      if (dl && G.part_mbpipe && p_Inp->rdopt == 0 && p_Inp->slice_mode == FIXED_MB)
        while (*dl && nd < 8) { devs[nd++] = atoi(dl); dl = strchr(dl, ','); if (!dl) break; dl++; }
      if (nd > 1) cfg.device = devs[0];
      init_prof("adapter_on: before jmhip_create");
      rc = jmhip_create(&G.ctx, &cfg);
      init_prof("adapter_on: jmhip_create done");
      if (rc != JMHIP_OK) {                                /* the library has no CPU fallback; say so and stop */
        fprintf(stderr, "jmhip adapter: jmhip_create failed (%d): %s\n", rc, jmhip_last_error(NULL));
        exit(70);
      }
      G.ctxs[0] = G.ctx; G.nctx = 1;
      for (k = 1; k < nd; k++) {
        cfg.device = devs[k];
        if ((rc = jmhip_create(&G.ctxs[k], &cfg)) != JMHIP_OK) { fprintf(stderr, "jmhip adapter: jmhip_create on device %d failed (%d): %s\n", devs[k], rc, jmhip_last_error(NULL)); exit(70); }
        G.nctx = k + 1;
      }
    }
    G.planes = (uint16_t *)malloc((size_t)16 * (G.W + 2 * JMHIP_PAD_X) * (G.H + 2 * JMHIP_PAD_Y) * sizeof(uint16_t));
    G.tables = (uint16_t *)malloc((size_t)7 * 16 * (2 * G.R + 1) * (2 * G.R + 1) * sizeof(uint16_t));
    G.dbmb = (jmhip_db_mb *)calloc((size_t)(G.W / 16) * (G.H / 16), sizeof(jmhip_db_mb));
    G.dbmo = (jmhip_db_motion *)calloc((size_t)(G.W / 4) * (G.H / 4), sizeof(jmhip_db_motion));
    if (!G.planes || !G.tables || !G.dbmb || !G.dbmo) { fprintf(stderr, "jmhip adapter: out of memory\n"); exit(70); }
    if (has_part(parts, "nulltrace")) trace_to_null_sink(p_Inp);
    if (has_part(parts, "readframe") && p_Vid->buf2img == buf2img_basic) p_Vid->buf2img = buf2img_bytes;
  }
  return !G.off;
}

/* ------------------------------------------------------------------ reference pictures <-> device slots */
static int slot_find(StorablePicture *s)
{
  int k;
  for (k = 0; k < G.nslots; k++) if (G.slot_pic[k] == s) { G.slot_tick[k] = ++G.tick; return k; }
  return -1;
}
static int slot_take(StorablePicture *s)
{
  int k, best = 0;
  if ((k = slot_find(s)) >= 0) return k;
  for (k = 0; k < G.nslots; k++) {
    if (!G.slot_pic[k]) { best = k; break; }
    if (G.slot_tick[k] < G.slot_tick[best]) best = k;
  }
  G.slot_pic[best] = s; G.slot_tick[best] = ++G.tick; G.slot_chroma[best] = 0;
  return best;
}
/* slot holding the sub-pel planes of `s`; a picture the slot ring has dropped is rebuilt from its luma */
static int slot_of_reference(StorablePicture *s)
{
  int k = slot_find(s);
  if (k < 0) {
    int rc;
    k = slot_take(s);
    rc = jmhip_set_reference(G.ctx, k, s->imgY[0], (int)(s->imgY[1] - s->imgY[0]));
    if (rc) adapter_die("jmhip_set_reference", rc);
  }
  return k;
}

/* ------------------------------------------------------------------ K5: sub-pel planes */
extern void __real_getSubImagesLuma(VideoParameters *, StorablePicture *);
void __wrap_getSubImagesLuma(VideoParameters *p_Vid, StorablePicture *s)
{
  int rc, k, j, i, y;
  const int Wp = s->size_x + 2 * JMHIP_PAD_X, Hp = s->size_y + 2 * JMHIP_PAD_Y;
  { const double t0_ = now_s(); if (adapter_on(p_Vid) && pipe_reference(s)) { T_interp += now_s() - t0_; if (TL_n) { TL_ip[TL_n - 1] += now_s() - t0_; TL_ip_at[TL_n - 1] = t0_; } return; } }   /* the macroblock pipeline's picture: the planes are made on the device and stay there */
  if (!adapter_on(p_Vid) || !G.part_interp || s->size_x != G.W || s->size_y != G.H ||
      s->size_x_padded != Wp || s->size_y_padded != Hp) {
    G.n_passed++;
    __real_getSubImagesLuma(p_Vid, s);
    if (G.ctx && !G.off) { k = slot_find(s); if (k >= 0) G.slot_pic[k] = NULL; }     /* device copy (if any) is stale now */
    return;
  }
  k = slot_take(s);
  G.slot_chroma[k] = 0;                                     /* the picture's samples are new: its chroma planes go up again when first needed */
  if ((rc = jmhip_set_reference(G.ctx, k, s->imgY[0], (int)(s->imgY[1] - s->imgY[0])))) adapter_die("jmhip_set_reference", rc);
  if ((rc = jmhip_get_subplanes(G.ctx, k, G.planes))) adapter_die("jmhip_get_subplanes", rc);
  for (j = 0; j < 4; j++)                                   /* JM's host-side MC / RDO keep reading p_curr_img_sub */
    for (i = 0; i < 4; i++) {
      const uint16_t *src = G.planes + (size_t)(j * 4 + i) * Wp * Hp;
      imgpel **dst = s->p_curr_img_sub[j][i];
      for (y = 0; y < Hp; y++) memcpy(&dst[y - JMHIP_PAD_Y][-JMHIP_PAD_X], src + (size_t)y * Wp, (size_t)Wp * sizeof(imgpel));
    }
  G.n_interp++;
}

/* ------------------------------------------------------------------ K6: chroma sub-images
 * getSubImagesChroma (lencod/src/img_chroma.c:338; UnifiedOneForthPix calls it right after getSubImagesLuma when ChromaMCBuffer = 1). */
static int slot_with_chroma(StorablePicture *s);
extern void __real_getSubImagesChroma(VideoParameters *, StorablePicture *);
void __wrap_getSubImagesChroma(VideoParameters *p_Vid, StorablePicture *s)
{
  const int fmt = p_Vid->yuv_format, ny = fmt == YUV422 ? 4 : 8, py = p_Vid->pad_size_uv_y, px = p_Vid->pad_size_uv_x;
  const int Wp = s->size_x_cr + 2 * px, Hp = s->size_y_cr + 2 * py;
  int uv, j, i, y, k, rc;
  if (adapter_on(p_Vid) && slot_find(s) >= 0 && G.slot_chroma[slot_find(s)] == 2) return;   /* the macroblock pipeline's picture (see pipe_reference) */
  if (!adapter_on(p_Vid) || !G.part_interpc || (fmt != YUV420 && fmt != YUV422) || p_Vid->p_Inp->OnTheFlyFractMCP ||
      s->size_x != G.W || s->size_y != G.H || px != (JMHIP_PAD_X >> 1) || py != (fmt == YUV422 ? JMHIP_PAD_Y : JMHIP_PAD_Y >> 1)) {
    G.n_passed++;
    __real_getSubImagesChroma(p_Vid, s);
    return;
  }
  if (!G.cplanes && !(G.cplanes = (uint16_t *)malloc((size_t)ny * 8 * Wp * Hp * sizeof(uint16_t)))) { fprintf(stderr, "jmhip adapter: out of memory\n"); exit(70); }
  k = slot_with_chroma(s);
  for (uv = 0; uv < 2; uv++) {
    if ((rc = jmhip_get_chroma_subplanes(G.ctx, k, uv, G.cplanes))) adapter_die("jmhip_get_chroma_subplanes", rc);
    for (j = 0; j < ny; j++)
      for (i = 0; i < 8; i++) {
        const uint16_t *src = G.cplanes + (size_t)(j * 8 + i) * Wp * Hp;
        imgpel **dst = s->p_img_sub[uv + 1][j][i];
        for (y = 0; y < Hp; y++) memcpy(&dst[y - py][-px], src + (size_t)y * Wp, (size_t)Wp * sizeof(imgpel));
      }
  }
  G.n_interpc++;
}

/* ------------------------------------------------------------------ current picture: uploaded once per coded picture */
extern int __real_encode_one_slice(VideoParameters *, int, int);
int __wrap_encode_one_slice(VideoParameters *p_Vid, int SliceGroupId, int TotalCodedMBs)
{
  const double t0_ = now_s();
  int n_;
  if (TotalCodedMBs == 0 && adapter_on(p_Vid) && (G.part_fs || G.part_subpel || G.part_ffs || G.part_eval || G.part_evalp) && !pipe_active() && p_Vid->structure == FRAME) {
    int rc = jmhip_set_current(G.ctx, p_Vid->pCurImg[0], (int)(p_Vid->pCurImg[1] - p_Vid->pCurImg[0]));
    if (rc) adapter_die("jmhip_set_current", rc);
    G.n_cur++;
  }
  if (TotalCodedMBs == 0 && TL_n < 64) TL_in[TL_n++] = t0_;
  if (TotalCodedMBs == 0 && TL_n <= 2) init_prof("encode_one_slice: entered (a picture's first slice)");
  n_ = __real_encode_one_slice(p_Vid, SliceGroupId, TotalCodedMBs);
  if (TL_n <= 2) init_prof("encode_one_slice: left");
  T_slice += now_s() - t0_;
  if (TL_n) TL_out[TL_n - 1] = now_s();
  return n_;
}

/* partition index of the ABI (jmhip.h) from JM's (blocktype, block_x, block_y) with block_* in 4x4 units */
static int partition_of(const MEBlock *b)
{
  const int bx = b->block_x, by = b->block_y;
  switch (b->blocktype) {
  case 1: return 0;
  case 2: return 1 + (by >> 1);
  case 3: return 3 + (bx >> 1);
  case 4: return 5 + (by >> 1) * 2 + (bx >> 1);
  case 5: return 9 + by * 2 + (bx >> 1);
  case 6: return 17 + (by >> 1) * 4 + bx;
  case 7: return 25 + by * 4 + bx;
  default: return -1;
  }
}
static int me_common_ok(Macroblock *currMB, MEBlock *mv_block)
{
  VideoParameters *p_Vid = currMB->p_Vid;
  Slice *currSlice = currMB->p_Slice;
  return adapter_on(p_Vid) && p_Vid->structure == FRAME && !currSlice->mb_aff_frame_flag && !mv_block->ChromaMEEnable &&
         !mv_block->apply_weights && G.n_cur > 0;
}

/* ------------------------------------------------------------------ K1-K3: full search (Macroblock.IntPelME) */
extern distblk __real_full_search_motion_estimation(Macroblock *, MotionVector *, MEBlock *, distblk, int);
distblk __wrap_full_search_motion_estimation(Macroblock *currMB, MotionVector *pred_mv, MEBlock *mv_block, distblk min_mcost, int lambda_factor)
{
  InputParameters *p_Inp = currMB->p_Inp;
  Slice *currSlice = currMB->p_Slice;
  const int list = mv_block->list, p = partition_of(mv_block);
  const int R = imin(mv_block->searchRange.max_x, mv_block->searchRange.max_y) >> 2;
  MotionVector *mv = &mv_block->mv[list];
  jmhip_me_job job;
  jmhip_me_result res;
  int rc, slot;
  if (!me_common_ok(currMB, mv_block) || !G.part_fs || !p_Inp->rdopt || mv_block->computePredFPel != computeSAD || p < 0 ||
      R < 1 || R > G.R || lambda_factor < 0 || mv_block->pos_x != currMB->pix_x + 4 * mv_block->block_x ||
      mv_block->pos_y != currMB->opix_y + 4 * mv_block->block_y) {
    G.n_passed++;
    { distblk r_; G.in_real_me++; r_ = __real_full_search_motion_estimation(currMB, pred_mv, mv_block, min_mcost, lambda_factor); G.in_real_me--; return r_; }
  }
  slot = slot_of_reference(currSlice->listX[list + currMB->list_offset][(int)mv_block->ref_idx]);
  memset(&job, 0, sizeof job);
  job.mb_x = (int16_t)currMB->pix_x; job.mb_y = (int16_t)currMB->opix_y;
  job.center_x = mv->mv_x; job.center_y = mv->mv_y;
  job.search_range = (int16_t)R; job.max_mvd = 0; job.lambda = lambda_factor;
  job.part_mask = 1ull << p;
  job.pred[p][0] = pred_mv->mv_x; job.pred[p][1] = pred_mv->mv_y;
  if ((rc = jmhip_me_fullsearch(G.ctx, slot, &job, 1, &res))) adapter_die("jmhip_me_fullsearch", rc);
  G.n_fs++;
  if ((distblk)res.best[p].cost < min_mcost) {            /* strict '<' against the cost carried in, me_fullsearch.c:89 */
    mv->mv_x = res.best[p].mv_x; mv->mv_y = res.best[p].mv_y;
    return (distblk)res.best[p].cost;
  }
  return min_mcost;
}

/* ------------------------------------------------------------------ K4: sub-pel refinement (Macroblock.SubPelME) */
extern distblk __real_sub_pel_motion_estimation(Macroblock *, MotionVector *, MEBlock *, distblk, int *);
distblk __wrap_sub_pel_motion_estimation(Macroblock *currMB, MotionVector *pred, MEBlock *mv_block, distblk min_mcost, int *lambda)
{
  VideoParameters *p_Vid = currMB->p_Vid;
  InputParameters *p_Inp = currMB->p_Inp;
  Slice *currSlice = currMB->p_Slice;
  const int list = mv_block->list;
  const int mh = p_Inp->MEErrorMetric[H_PEL], mq = p_Inp->MEErrorMetric[Q_PEL];
  MotionVector *mv = &mv_block->mv[list];
  jmhip_subpel_job job;
  jmhip_me_best best;
  int rc, slot;
  if (!me_common_ok(currMB, mv_block) || !G.part_subpel || !p_Inp->rdopt || mv_block->search_pos2 != 9 || mv_block->search_pos4 != 9 ||
      !((mh == ERROR_SAD && mv_block->computePredHPel == computeSAD) || (mh == ERROR_SATD && mv_block->computePredHPel == computeSATD)) ||
      !((mq == ERROR_SAD && mv_block->computePredQPel == computeSAD) || (mq == ERROR_SATD && mv_block->computePredQPel == computeSATD)) ||
      lambda[H_PEL] < 0 || lambda[Q_PEL] < 0) {
    G.n_passed++;
    { distblk r_; G.in_real_me++; r_ = __real_sub_pel_motion_estimation(currMB, pred, mv_block, min_mcost, lambda); G.in_real_me--; return r_; }
  }
  slot = slot_of_reference(currSlice->listX[list + currMB->list_offset][(int)mv_block->ref_idx]);
  memset(&job, 0, sizeof job);
  job.pos_x = mv_block->pos_x; job.pos_y = mv_block->pos_y; job.bsx = mv_block->blocksize_x; job.bsy = mv_block->blocksize_y;
  job.pred_x = pred->mv_x; job.pred_y = pred->mv_y; job.mv_x = mv->mv_x; job.mv_y = mv->mv_y;
  job.lambda_h = lambda[H_PEL]; job.lambda_q = lambda[Q_PEL];
  job.metric_h = (int8_t)mh; job.metric_q = (int8_t)mq;
  job.start_hp = (int8_t)p_Vid->start_me_refinement_hp; job.start_qp = (int8_t)p_Vid->start_me_refinement_qp;
  job.test8x8 = (int8_t)(mv_block->test8x8 != 0);
  job.min_mcost = min_mcost > 0x7fffffff ? 0x7fffffff : (int32_t)min_mcost;
  if ((rc = jmhip_me_subpel(G.ctx, slot, &job, 1, &best))) adapter_die("jmhip_me_subpel", rc);
  G.n_subpel++;
  mv->mv_x = best.mv_x; mv->mv_y = best.mv_y;
  return (distblk)best.cost;
}

/* ------------------------------------------------------------------ K1/K2: fast full search tables (Macroblock.p_SetupFastFullPelSearch) */
extern void __real_setup_fast_full_search(Macroblock *, MEBlock *, int);
void __wrap_setup_fast_full_search(Macroblock *currMB, MEBlock *mv_block, int list)
{
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currSlice->p_Vid;
  InputParameters *p_Inp = currSlice->p_Inp;
  MEFullFast *ff = p_Vid->p_ffast_me;
  const int ref = mv_block->ref_idx;
  const int R = ff->max_search_range[list][ref], max_pos = (2 * R + 1) * (2 * R + 1), range_q = R << 2;
  const int wp = ((p_Vid->active_pps->weighted_pred_flag && (currSlice->slice_type == P_SLICE || currSlice->slice_type == SP_SLICE)) ||
                  (p_Vid->active_pps->weighted_bipred_idc && currSlice->slice_type == B_SLICE)) && p_Inp->UseWeightedReferenceME;
  PixelPos block[4];
  MotionVector pmv, c;
  jmhip_me_job job;
  int rc, slot, t, k, pos;
  if (!me_common_ok(currMB, mv_block) || !G.part_ffs || wp || p_Inp->MEErrorMetric[F_PEL] != ERROR_SAD || R < 1 || R > G.R) {
    G.n_passed++;
    G.in_real_me++; __real_setup_fast_full_search(currMB, mv_block, list); G.in_real_me--;
    return;
  }
  /* search centre = the 16x16 predictor rounded to full-pel and kept inside the level's MV range (me_fullfast.c:307-327) */
  get_neighbors(currMB, block, 0, 0, 16);
  currMB->GetMVPredictor(currMB, block, &pmv, (short)ref, p_Vid->enc_picture->mv_info, list, 0, 0, 16, 16);
  c.mv_x = (short)(((pmv.mv_x + 2) >> 2) * 4);
  c.mv_y = (short)(((pmv.mv_y + 2) >> 2) * 4);
  if (!p_Inp->rdopt) { c.mv_x = (short)iClip3(-range_q, range_q, c.mv_x); c.mv_y = (short)iClip3(-range_q, range_q, c.mv_y); }
  c.mv_x = (short)iClip3(p_Vid->MaxHmvR[4] + range_q, p_Vid->MaxHmvR[5] - range_q, c.mv_x);
  c.mv_y = (short)iClip3(p_Vid->MaxVmvR[4] + range_q, p_Vid->MaxVmvR[5] - range_q, c.mv_y);
  ff->search_center[list][ref] = c;
  ff->search_center_padded[list][ref] = pad_MVs(c, mv_block);
  if (!p_Inp->rdopt) {                                     /* spiral index of the (0,0) vector, me_fullfast.c:352-365 */
    const int rx = -c.mv_x, ry = -c.mv_y;
    for (pos = 0; pos < max_pos; pos++)
      if (rx == p_Vid->spiral_qpel_search[pos].mv_x && ry == p_Vid->spiral_qpel_search[pos].mv_y) { ff->pos_00[list][ref] = pos; break; }
  }
  slot = slot_of_reference(currSlice->listX[list + currMB->list_offset][ref]);
  memset(&job, 0, sizeof job);
  job.mb_x = (int16_t)currMB->pix_x; job.mb_y = (int16_t)currMB->opix_y;
  job.center_x = c.mv_x; job.center_y = c.mv_y; job.search_range = (int16_t)R;
  if ((rc = jmhip_me_sad_tables(G.ctx, slot, &job, 1, G.tables))) adapter_die("jmhip_me_sad_tables", rc);
  for (t = 1; t < 8; t++)                                  /* BlockSAD[list][ref][blocktype][4x4 raster index][spiral position], distpel */
    for (k = 0; k < 16; k++) {
      const uint16_t *src = G.tables + ((size_t)(t - 1) * 16 + k) * max_pos;
      distpel *dst = ff->BlockSAD[list][ref][t][k];
      for (pos = 0; pos < max_pos; pos++) dst[pos] = src[pos];
    }
  ff->search_setup_done[list][ref] = 1;
  G.n_ffs++;
}

/* ------------------------------------------------------------------ candidate distortions: MEBlock.computePred{F,H,Q}Pel
 * computeSAD (lencod/src/me_distortion.c:349) / computeSATD (:745), reached through p_Dpb->pf_computeSAD / pf_computeSATD (lencod.c:355-357)
 * by the searches that stay on the host: EPZS's predictor and pattern walk (me_epzs*.c) evaluates its candidates one by one.
 * JM's early exit returns min_mcost itself as soon as a partial sum exceeds min_mcost >> 5 (dist_scale_f, mv_search.h:20), and the
 * last check covers the whole block: the result is min_mcost when the full distortion exceeds that bound, else distortion << 5. */
/* EPZS asks for its candidates one by one (a predictor list, then pattern rounds around the running best: me_epzs_int.c:212-247 / :289-330; patterns
 * me_epzs_common.c:48-73, steps of 2, 4 and 8 quarter-pel units).  A distortion is a pure function of (reference, block, candidate, metric), so
 * the adapter answers from a cache and fills it a batch at a time: on a miss the device evaluates, in ONE jmhip_me_eval call, the candidate asked for,
 * the 5 x 5 grid of step 4 around it plus the four step-2 neighbours (every point a pattern round can ask for next; for the SATD metric of the sub-pel
 * stages the 5 x 5 grid of step 1 instead), and -- when the candidate is an entry of the slice's predictor list -- the entries that follow it.  JM's control flow, its strict-'<' scans and its early-exit return values are
 * untouched; only the number of PCIe round trips changes. */
#define EC_CAP 1024
#define EC_CTX 8
static struct ec_ctx {
  StorablePicture *ref; int pos_x, pos_y, bsx, bsy, metric, t8; long pic; unsigned tick;
  int n; int16_t x[EC_CAP], y[EC_CAP]; int32_t d[EC_CAP];
} EC_[EC_CTX], *ECp;                                          /* a few (reference, block, metric) contexts: JM comes back to a block for the next reference / stage */
#define EC (*ECp)
static unsigned ec_tick;
static int ec_find(int x, int y) { int k; for (k = EC.n - 1; k >= 0; k--) if (EC.x[k] == x && EC.y[k] == y) return k; return -1; }
static int32_t cand_cache_get(StorablePicture *ref1, MEBlock *mv_block, int bsx, int bsy, int metric, int rx, int ry)
{
  static const int8_t cross[4][2] = {{2, 0}, {-2, 0}, {0, 2}, {0, -2}};
  jmhip_cand c[96];
  int32_t dist[96];
  int n = 0, k, i, j, rc;
  const int t8 = mv_block->test8x8 != 0;
  {
    struct ec_ctx *lru = &EC_[0];
    ECp = NULL;
    for (k = 0; k < EC_CTX; k++) {
      struct ec_ctx *e = &EC_[k];
      if (e->ref == ref1 && e->pos_x == mv_block->pos_x && e->pos_y == mv_block->pos_y && e->bsx == bsx && e->bsy == bsy && e->metric == metric && e->t8 == t8 && e->pic == G.n_cur) { ECp = e; break; }
      if (e->tick < lru->tick) lru = e;
    }
    if (!ECp) {
      ECp = lru;
      EC.ref = ref1; EC.pos_x = mv_block->pos_x; EC.pos_y = mv_block->pos_y; EC.bsx = bsx; EC.bsy = bsy; EC.metric = metric; EC.t8 = t8; EC.pic = G.n_cur; EC.n = 0;
    }
    EC.tick = ++ec_tick;
  }
  if ((k = ec_find(rx, ry)) >= 0) { G.n_eval_hits++; return EC.d[k]; }
  if (EC.n + 96 > EC_CAP) EC.n = 0;
#define EC_ADD(X, Y) do { const int x_ = (X), y_ = (Y); int q_, dup_ = 0; if (x_ >= -32768 && x_ <= 32767 && y_ >= -32768 && y_ <= 32767 && n < 96 && ec_find(x_, y_) < 0) { \
    for (q_ = 0; q_ < n; q_++) if (c[q_].cand_x == x_ && c[q_].cand_y == y_) { dup_ = 1; break; } \
    if (!dup_) { memset(&c[n], 0, sizeof c[n]); c[n].pos_x = mv_block->pos_x; c[n].pos_y = mv_block->pos_y; c[n].bsx = (int16_t)bsx; c[n].bsy = (int16_t)bsy; \
      c[n].cand_x = (int16_t)x_; c[n].cand_y = (int16_t)y_; c[n].metric = (int16_t)metric; c[n].test8x8 = (int16_t)t8; n++; } } } while (0)
  EC_ADD(rx, ry);
  if (G.part_evalbatch) {
    Slice *sl = mv_block->p_Vid->currentSlice;
    if (metric == JMHIP_METRIC_SAD) {                       /* the integer / EPZS stage: pattern steps of 4 and 8 (and 2) */
      for (j = -8; j <= 8; j += 4) for (i = -8; i <= 8; i += 4) EC_ADD(rx + i, ry + j);
      for (k = 0; k < 4; k++) EC_ADD(rx + cross[k][0], ry + cross[k][1]);
    } else {                                                 /* the sub-pel stages: steps of 1 and 2 quarter-pel units */
      for (j = -2; j <= 2; j++) for (i = -2; i <= 2; i++) EC_ADD(rx + i, ry + j);
    }
    if (sl && sl->p_EPZS && sl->p_EPZS->predictor && sl->p_EPZS->predictor->point) {        /* the predictors JM is about to ask for */
      const EPZSStructure *pr = sl->p_EPZS->predictor;
      for (k = 0; k < pr->searchPoints && k < 64; k++)
        if (pr->point[k].motion.mv_x == rx && pr->point[k].motion.mv_y == ry) {
          for (i = k + 1; i < pr->searchPoints && i < k + 40; i++) EC_ADD(pr->point[i].motion.mv_x, pr->point[i].motion.mv_y);
          break;
        }
    }
  }
#undef EC_ADD
  if ((rc = jmhip_me_eval(G.ctx, slot_of_reference(ref1), c, n, dist))) adapter_die("jmhip_me_eval", rc);
  G.n_eval_calls++; G.n_eval_cands += n;
  for (k = 0; k < n; k++) { EC.x[EC.n] = c[k].cand_x; EC.y[EC.n] = c[k].cand_y; EC.d[EC.n] = dist[k]; EC.n++; }
  return dist[0];
}

static distblk eval_candidate(StorablePicture *ref1, MEBlock *mv_block, distblk min_mcost, MotionVector *cand, int metric,
                              distblk (*real)(StorablePicture *, MEBlock *, distblk, MotionVector *))
{
  VideoParameters *p_Vid = mv_block->p_Vid;
  int32_t dist;
  const int bsx = mv_block->blocksize_x, bsy = mv_block->blocksize_y;
  const int rx = cand->mv_x - (mv_block->pos_x << 2), ry = cand->mv_y - (mv_block->pos_y << 2);
  if (G.in_real_me || !adapter_on(p_Vid) || !G.part_eval || G.n_cur == 0 || p_Vid->structure != FRAME || mv_block->ChromaMEEnable ||
      !ref1 || ref1->size_x != G.W || ref1->size_y != G.H || (bsx != 4 && bsx != 8 && bsx != 16) || (bsy != 4 && bsy != 8 && bsy != 16) ||
      rx < -32768 || rx > 32767 || ry < -32768 || ry > 32767 || (metric == JMHIP_METRIC_SATD && mv_block->test8x8 && (bsx < 8 || bsy < 8))) {
    G.n_passed++;
    return real(ref1, mv_block, min_mcost, cand);
  }
  dist = cand_cache_get(ref1, mv_block, bsx, bsy, metric, rx, ry);
  G.n_eval++;
  return ((distblk)dist >> 5) > (min_mcost >> 5) ? min_mcost : (distblk)dist;
}
extern distblk __real_computeSAD(StorablePicture *, MEBlock *, distblk, MotionVector *);
distblk __wrap_computeSAD(StorablePicture *ref1, MEBlock *mv_block, distblk min_mcost, MotionVector *cand)
{
  return eval_candidate(ref1, mv_block, min_mcost, cand, JMHIP_METRIC_SAD, __real_computeSAD);
}
extern distblk __real_computeSATD(StorablePicture *, MEBlock *, distblk, MotionVector *);
distblk __wrap_computeSATD(StorablePicture *ref1, MEBlock *mv_block, distblk min_mcost, MotionVector *cand)
{
  return eval_candidate(ref1, mv_block, min_mcost, cand, JMHIP_METRIC_SATD, __real_computeSATD);
}

/* ------------------------------------------------------------------ weighted / bi-predictive candidate distortions
 * compute{SAD,SATD,SSE}WP (me_distortion.c:434 / :833 / :1261; MEBlock.computePred{F,H,Q}Pel when MEBlock.apply_weights), computeSSE (:1190) and
 * computeBiPred{SAD,SATD,SSE}1 / 2 (:525 / :943 / :1353 un-weighted, :624 / :1038 / :1438 weighted; MEBlock.computeBiPred{F,H,Q}Pel), reached
 * through p_Dpb->pf_compute* (lencod.c:355-366) by JM's bi-predictive searches and by every search of a weighted slice.  The weights are read
 * where the reference's functions read them: MEBlock.weight_luma / offset_luma, MEBlock.weight1 / weight2 / offsetBi, Slice.wp_luma_round and
 * Slice.luma_log_weight_denom.  Same early-exit rule as eval_candidate. */
static int pred_candidate_ok(StorablePicture *ref, MEBlock *mv_block, int metric)
{
  VideoParameters *p_Vid = mv_block->p_Vid;
  const int bsx = mv_block->blocksize_x, bsy = mv_block->blocksize_y;
  return !G.in_real_me && adapter_on(p_Vid) && G.part_evalp && G.n_cur != 0 && p_Vid->structure == FRAME && !mv_block->ChromaMEEnable &&
         ref && ref->size_x == G.W && ref->size_y == G.H && (bsx == 4 || bsx == 8 || bsx == 16) && (bsy == 4 || bsy == 8 || bsy == 16) &&
         !(metric == JMHIP_METRIC_SATD && mv_block->test8x8 && (bsx < 8 || bsy < 8));
}
static int rel_mv(MEBlock *mv_block, MotionVector *cand, int16_t *rx, int16_t *ry)
{
  const int x = cand->mv_x - (mv_block->pos_x << 2), y = cand->mv_y - (mv_block->pos_y << 2);
  if (x < -32768 || x > 32767 || y < -32768 || y > 32767) return 0;
  *rx = (int16_t)x; *ry = (int16_t)y;
  return 1;
}
static distblk eval_pred_candidate(StorablePicture *ref1, StorablePicture *ref2, MEBlock *mv_block, distblk min_mcost, MotionVector *cand1,
                                   MotionVector *cand2, int metric, int pred, int *handled)
{
  Slice *currSlice = mv_block->p_Slice;
  jmhip_pred_cand c;
  int32_t dist;
  int rc;
  const int two = pred == JMHIP_PRED_AVG || pred == JMHIP_PRED_BI_WP;
  *handled = 0;
  memset(&c, 0, sizeof c);
  if (!pred_candidate_ok(ref1, mv_block, metric) || (two && !pred_candidate_ok(ref2, mv_block, metric)) ||
      !rel_mv(mv_block, cand1, &c.cand_x[0], &c.cand_y[0]) || (two && !rel_mv(mv_block, cand2, &c.cand_x[1], &c.cand_y[1])) ||
      currSlice->luma_log_weight_denom < 0 || currSlice->luma_log_weight_denom > 7) {
    G.n_passed++;
    return 0;
  }
  c.pos_x = mv_block->pos_x; c.pos_y = mv_block->pos_y; c.bsx = mv_block->blocksize_x; c.bsy = mv_block->blocksize_y;
  c.metric = (int8_t)metric; c.test8x8 = (int8_t)(mv_block->test8x8 != 0); c.pred = (int8_t)pred;
  c.slot[0] = (int8_t)slot_of_reference(ref1);
  if (two) c.slot[1] = (int8_t)slot_of_reference(ref2);
  if (pred == JMHIP_PRED_BI_WP) {
    c.weight[0] = mv_block->weight1; c.weight[1] = mv_block->weight2; c.offset = mv_block->offsetBi;
    c.round = (int16_t)(2 * currSlice->wp_luma_round); c.shift = (int8_t)(currSlice->luma_log_weight_denom + 1);
  } else if (pred == JMHIP_PRED_UNI_WP) {
    c.weight[0] = mv_block->weight_luma; c.offset = mv_block->offset_luma;
    c.round = (int16_t)currSlice->wp_luma_round; c.shift = (int8_t)currSlice->luma_log_weight_denom;
  }
  if ((rc = jmhip_me_eval_pred(G.ctx, &c, 1, &dist))) adapter_die("jmhip_me_eval_pred", rc);
  G.n_evalp++;
  *handled = 1;
  return ((distblk)dist >> 5) > (min_mcost >> 5) ? min_mcost : (distblk)dist;
}
#define WRAP_UNI(fn, metric, pred) \
  extern distblk __real_##fn(StorablePicture *, MEBlock *, distblk, MotionVector *); \
  distblk __wrap_##fn(StorablePicture *ref1, MEBlock *mv_block, distblk min_mcost, MotionVector *cand) \
  { int done; distblk d = eval_pred_candidate(ref1, NULL, mv_block, min_mcost, cand, NULL, metric, pred, &done); \
    return done ? d : __real_##fn(ref1, mv_block, min_mcost, cand); }
#define WRAP_BI(fn, metric, pred) \
  extern distblk __real_##fn(StorablePicture *, StorablePicture *, MEBlock *, distblk, MotionVector *, MotionVector *); \
  distblk __wrap_##fn(StorablePicture *ref1, StorablePicture *ref2, MEBlock *mv_block, distblk min_mcost, MotionVector *cand1, MotionVector *cand2) \
  { int done; distblk d = eval_pred_candidate(ref1, ref2, mv_block, min_mcost, cand1, cand2, metric, pred, &done); \
    return done ? d : __real_##fn(ref1, ref2, mv_block, min_mcost, cand1, cand2); }
WRAP_UNI(computeSADWP, JMHIP_METRIC_SAD, JMHIP_PRED_UNI_WP) WRAP_UNI(computeSATDWP, JMHIP_METRIC_SATD, JMHIP_PRED_UNI_WP)
WRAP_UNI(computeSSEWP, JMHIP_METRIC_SSE, JMHIP_PRED_UNI_WP) WRAP_UNI(computeSSE, JMHIP_METRIC_SSE, JMHIP_PRED_UNI)
WRAP_BI(computeBiPredSAD1, JMHIP_METRIC_SAD, JMHIP_PRED_AVG) WRAP_BI(computeBiPredSATD1, JMHIP_METRIC_SATD, JMHIP_PRED_AVG)
WRAP_BI(computeBiPredSSE1, JMHIP_METRIC_SSE, JMHIP_PRED_AVG)
WRAP_BI(computeBiPredSAD2, JMHIP_METRIC_SAD, JMHIP_PRED_BI_WP) WRAP_BI(computeBiPredSATD2, JMHIP_METRIC_SATD, JMHIP_PRED_BI_WP)
WRAP_BI(computeBiPredSSE2, JMHIP_METRIC_SSE, JMHIP_PRED_BI_WP)

/* ------------------------------------------------------------------ luma intra prediction
 * get_intrapred_4x4 (lencod/src/intra4x4.c:521) and the Intra16x16 mode search find_sad_16x16_JM (intra16x16.c:463; Slice.find_sad_16x16,
 * rdopt.c:301).  JM keeps gathering the predictor samples (set_intrapred_4x4 / set_intrapred_16x16: neighbour availability, constrained
 * intra prediction); the predictions, the mode cost and the mode choice come from the device. */
extern void __real_get_intrapred_4x4(Macroblock *, ColorPlane, int, int, int, int, int);
void __wrap_get_intrapred_4x4(Macroblock *currMB, ColorPlane pl, int mode, int img_x, int img_y, int left, int up)
{
  jmhip_ip4_blk b;
  uint8_t out[16];
  int i, j, rc;
  if (!adapter_on(currMB->p_Vid) || !G.part_ip4 || pl != PLANE_Y || mode < 0 || mode > 8) {
    G.n_passed++;
    __real_get_intrapred_4x4(currMB, pl, mode, img_x, img_y, left, up);
    return;
  }
  for (i = 0; i < 13; i++) b.edge[i] = (uint8_t)currMB->intra4x4_pred[pl][i];
  b.mode = (uint8_t)mode; b.left = (uint8_t)(left != 0); b.up = (uint8_t)(up != 0);
  if ((rc = jmhip_intrapred4x4(G.ctx, &b, 1, out))) adapter_die("jmhip_intrapred4x4", rc);
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) currMB->p_Slice->mpr_4x4[pl][mode][j][i] = out[4 * j + i];
  G.n_ip4++;
}
/* get_intrapred_8x8 (lencod/src/intra8x8.c:716, called from rd_intra_jm.c:291 / rd_intra_jm_low.c:226): set_intrapred_8x8 (neighbours,
 * LowPassForIntra8x8Pred) stays with JM; the mode's prediction comes from the device */
extern void __real_get_intrapred_8x8(Macroblock *, ColorPlane, int, int, int);
void __wrap_get_intrapred_8x8(Macroblock *currMB, ColorPlane pl, int mode, int left, int up)
{
  jmhip_ip8_blk b;
  uint8_t out[64];
  int i, j, rc;
  if (!adapter_on(currMB->p_Vid) || !G.part_ip8 || pl != PLANE_Y || mode < 0 || mode > 8) {
    G.n_passed++;
    __real_get_intrapred_8x8(currMB, pl, mode, left, up);
    return;
  }
  for (i = 0; i < 25; i++) b.edge[i] = (uint8_t)currMB->intra8x8_pred[pl][i];
  b.mode = (uint8_t)mode; b.left = (uint8_t)(left != 0); b.up = (uint8_t)(up != 0);
  if ((rc = jmhip_intrapred8x8(G.ctx, &b, 1, out))) adapter_die("jmhip_intrapred8x8", rc);
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) currMB->p_Slice->mpr_8x8[pl][mode][j][i] = out[8 * j + i];
  G.n_ip8++;
}
extern distblk __real_find_sad_16x16_JM(Macroblock *);
distblk __wrap_find_sad_16x16_JM(Macroblock *currMB)
{
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currMB->p_Vid;
  InputParameters *p_Inp = currMB->p_Inp;
  jmhip_i16_mb m;
  static jmhip_i16_out out;
  uint8_t orig[256];
  int left, up, all, k, j, i, rc;
  if (!adapter_on(p_Vid) || !G.part_i16 || currSlice->P444_joined || currSlice->set_intrapred_16x16 != set_intrapred_16x16) {
    G.n_passed++;
    return __real_find_sad_16x16_JM(currMB);
  }
  memset(&m, 0, sizeof m);
  currSlice->set_intrapred_16x16(currMB, PLANE_Y, &left, &up, &all);
  for (k = 0; k < 4; k++) {                                                   /* the modes JM's loop evaluates (intra16x16.c:483-494) */
    int off = 0;
    if (p_Inp->IntraDisableInterOnly == 0 || (currSlice->slice_type != I_SLICE && currSlice->slice_type != SI_SLICE))
      off = (p_Inp->Intra16x16ParDisable && (k == VERT_PRED_16 || k == HOR_PRED_16)) || (p_Inp->Intra16x16PlaneDisable && k == PLANE_16);
    if (!off && !((k == VERT_PRED_16 && !up) || (k == HOR_PRED_16 && !left) || (k == PLANE_16 && (!left || !up || !all)))) m.mode_mask |= (uint8_t)(1 << k);
  }
  for (i = 0; i < 33; i++) m.edge[i] = (uint8_t)currMB->intra16x16_pred[0][i];
  m.left = (uint8_t)(left != 0); m.up = (uint8_t)(up != 0);
  m.metric = (uint8_t)(p_Inp->ModeDecisionMetric == ERROR_SAD ? JMHIP_METRIC_SAD : (p_Inp->ModeDecisionMetric == ERROR_SSE ? JMHIP_METRIC_SSE : JMHIP_METRIC_SATD));
  for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) orig[j * 16 + i] = (uint8_t)p_Vid->pCurImg[currMB->opix_y + j][currMB->pix_x + i];
  if ((rc = jmhip_intra16_search(G.ctx, &m, orig, 1, &out))) adapter_die("jmhip_intra16_search", rc);
  for (k = 0; k < 4; k++)
    if ((m.mode_mask >> k) & 1)
      for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) currSlice->mpr_16x16[0][k][j][i] = out.pred[k][j * 16 + i];
  currMB->i16mode = (char)out.mode;
  G.n_i16++;
  return (distblk)out.cost;
}

/* ------------------------------------------------------------------ chroma intra prediction
 * intra_chroma_prediction (lencod/src/intra_chroma.c:530; Slice.intra_chroma_prediction, slice.c:1135): JM's neighbour logic (getNeighbour,
 * constrained intra prediction) stays here, the four predictions of both planes come from the device, and with RDOptimization = 0 JM's own
 * rdo_low_intra_chroma_decision runs on them as before (:774-777). */
extern void __real_intra_chroma_prediction(Macroblock *, int *, int *, int *);
void __wrap_intra_chroma_prediction(Macroblock *currMB, int *mb_up, int *mb_left, int *mb_up_left)
{
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currMB->p_Vid;
  InputParameters *p_Inp = currMB->p_Inp;
  PixelPos a, c, d;
  jmhip_ic_mb m;
  static uint8_t out[1024];
  int up, left[2], ul, uv, k, j, i, rc;
  const int ch = p_Vid->mb_cr_size_y;
  if (!adapter_on(p_Vid) || !G.part_ic || (p_Vid->yuv_format != YUV420 && p_Vid->yuv_format != YUV422) || p_Vid->mb_cr_size_x != 8 || currMB->mb_field) {
    G.n_passed++;
    __real_intra_chroma_prediction(currMB, mb_up, mb_left, mb_up_left);
    return;
  }
  p_Vid->getNeighbour(currMB, -1, -1, p_Vid->mb_size[IS_CHROMA], &d);
  p_Vid->getNeighbour(currMB, -1,  0, p_Vid->mb_size[IS_CHROMA], &a);
  p_Vid->getNeighbour(currMB,  0, -1, p_Vid->mb_size[IS_CHROMA], &c);
  up = c.available; ul = d.available; left[0] = left[1] = a.available;
  if (p_Inp->UseConstrainedIntraPred) {
    up = c.available ? p_Vid->intra_block[c.mb_addr] : 0;
    left[0] = left[1] = a.available ? p_Vid->intra_block[a.mb_addr] : 0;
    ul = d.available ? p_Vid->intra_block[d.mb_addr] : 0;
  }
  if (mb_up) *mb_up = up;
  if (mb_left) *mb_left = left[0];
  if (mb_up_left) *mb_up_left = ul;
  memset(&m, 0, sizeof m);
  m.up_avail = (uint8_t)(up != 0); m.left_avail = (uint8_t)(left[0] != 0); m.upleft_avail = (uint8_t)(ul != 0);
  for (uv = 0; uv < 2; uv++) {
    imgpel **img = p_Vid->enc_picture->imgUV[uv];
    if (up) for (i = 0; i < 8; i++) m.up[uv][i] = (uint8_t)img[c.pos_y][c.pos_x + i];
    if (left[0]) for (j = 0; j < ch; j++) m.left[uv][j] = (uint8_t)img[a.pos_y + j][a.pos_x];
    if (ul) m.corner[uv] = (uint8_t)img[d.pos_y][d.pos_x];
  }
  if ((rc = jmhip_intra_chroma(G.ctx, &m, 1, out))) adapter_die("jmhip_intra_chroma", rc);
  for (uv = 0; uv < 2; uv++)
    for (k = 0; k < 4; k++) {
      if (!(k == DC_PRED_8 || (k == VERT_PRED_8 && up) || (k == HOR_PRED_8 && left[0]) || (k == PLANE_8 && up && left[0] && ul))) continue;
      for (j = 0; j < ch; j++) for (i = 0; i < 8; i++) currSlice->mpr_16x16[uv + 1][k][j][i] = out[k * 256 + uv * 128 + j * 8 + i];
    }
  G.n_ic++;
  if (!p_Inp->rdopt) currSlice->rdo_low_intra_chroma_decision(currMB, up, left, ul);
}

/* ------------------------------------------------------------------ motion-compensated prediction
 * luma_prediction (lencod/src/mc_prediction.c:144; bound to p_Dpb->pf_luma_prediction in lencod.c:367) and chroma_prediction_4x4
 * (:568), frame macroblocks, un-weighted and weighted (weighted_mc_prediction / weighted_bi_prediction :38-73 with the parameters the two
 * functions hand over, :203-228 and :615-640); field / MBAFF and ChromaMCBuffer = 0 go to JM. */
static MotionVector *****mc_vectors(Macroblock *currMB, int p_dir, int m0, int m1, int r0, int r1, short bipred_me)
{
  Slice *sl = currMB->p_Slice;
  if (bipred_me && r0 == 0 && r1 == 0 && p_dir == 2 && is_bipred_enabled(currMB->p_Vid, m0) && is_bipred_enabled(currMB->p_Vid, m1))
    return sl->bipred_mv[bipred_me - 1];
  return sl->all_mv;
}
static int mc_common_ok(Macroblock *currMB, int p_dir)
{
  Slice *sl = currMB->p_Slice;
  if (!adapter_on(currMB->p_Vid) || p_dir < 0 || p_dir > 2 || currMB->list_offset != 0) return 0;
  if (sl->luma_log_weight_denom < 0 || sl->luma_log_weight_denom > 7 || sl->chroma_log_weight_denom < 0 || sl->chroma_log_weight_denom > 7) return 0;
  return 1;
}
/* apply_weights of luma_prediction / chroma_prediction_4x4 and, when set, the parameters they pass on; comp 0 = luma, 1 / 2 = U / V */
static int mc_weights(Slice *sl, int p_dir, int r0, int r1, int comp, jmhip_mc_weights *w)
{
  const int round_ = comp ? sl->wp_chroma_round : sl->wp_luma_round, denom = comp ? sl->chroma_log_weight_denom : sl->luma_log_weight_denom;
  memset(w, 0, sizeof *w);
  if (!(sl->weighted_prediction == 1 || (sl->weighted_prediction == 2 && p_dir == 2))) return 0;
  if (p_dir == 2) {
    w->weight[0] = sl->wbp_weight[0][r0][r1][comp]; w->weight[1] = sl->wbp_weight[1][r0][r1][comp];
    w->offset = (int16_t)((sl->wp_offset[0][r0][comp] + sl->wp_offset[1][r1][comp] + 1) >> 1);
    w->round = (int16_t)(round_ << 1); w->shift = (int8_t)(denom + 1);
  } else {
    const int r = p_dir ? r1 : r0;
    w->weight[p_dir] = sl->wp_weight[p_dir][r][comp]; w->offset = sl->wp_offset[p_dir][r][comp];
    w->round = (int16_t)round_; w->shift = (int8_t)denom;
  }
  return 1;
}
static int slot_with_chroma(StorablePicture *s)
{
  int k = slot_of_reference(s);
  if (!G.slot_chroma[k]) {
    int rc = jmhip_set_reference_chroma(G.ctx, k, s->imgUV[0][0], s->imgUV[1][0], (int)(s->imgUV[0][1] - s->imgUV[0][0]));
    if (rc) adapter_die("jmhip_set_reference_chroma", rc);
    G.slot_chroma[k] = 1;
  }
  return k;
}
extern void __real_luma_prediction(Macroblock *, int, int, int, int, int, int *, char *, short);
void __wrap_luma_prediction(Macroblock *currMB, int block_x, int block_y, int bsx, int bsy, int p_dir, int list_mode[2], char *ref_idx, short bipred_me)
{
  Slice *sl = currMB->p_Slice;
  jmhip_mc_luma_blk b;
  jmhip_mc_weights w;
  uint8_t out[256];
  int l, j, i, rc, ok = mc_common_ok(currMB, p_dir) && G.part_mcl && (bsx == 4 || bsx == 8 || bsx == 16) && (bsy == 4 || bsy == 8 || bsy == 16);
  memset(&b, 0, sizeof b);
  for (l = 0; ok && l < 2; l++)
    if (p_dir == l || p_dir == 2) {
      StorablePicture *pic = sl->listX[l][(short)ref_idx[l]];
      if (!pic || pic->size_x != G.W || pic->size_y != G.H) ok = 0;
    }
  if (!ok) { G.n_passed++; __real_luma_prediction(currMB, block_x, block_y, bsx, bsy, p_dir, list_mode, ref_idx, bipred_me); return; }
  {
    MotionVector *****mva = mc_vectors(currMB, p_dir, list_mode[0], list_mode[1], ref_idx[0], ref_idx[1], bipred_me);
    b.x = (int16_t)(currMB->pix_x + block_x); b.y = (int16_t)(currMB->opix_y + block_y);
    b.w = (uint8_t)bsx; b.h = (uint8_t)bsy; b.dir = (uint8_t)p_dir;
    for (l = 0; l < 2; l++)
      if (p_dir == l || p_dir == 2) {
        MotionVector *mv = &mva[l][(short)ref_idx[l]][list_mode[l]][block_y >> 2][block_x >> 2];
        b.slot[l] = (int8_t)slot_of_reference(sl->listX[l][(short)ref_idx[l]]);
        b.mv[l][0] = mv->mv_x; b.mv[l][1] = mv->mv_y;
      }
  }
  if (p_dir == 2 && G.slot_pic[b.slot[0]] != sl->listX[0][(short)ref_idx[0]]) b.slot[0] = (int8_t)slot_of_reference(sl->listX[0][(short)ref_idx[0]]);  /* list 1 may have evicted it */
  if ((rc = jmhip_mc_luma_wp(G.ctx, &b, mc_weights(sl, p_dir, (short)ref_idx[0], (short)ref_idx[1], 0, &w) ? &w : NULL, 1, out))) adapter_die("jmhip_mc_luma_wp", rc);
  for (j = 0; j < bsy; j++)
    for (i = 0; i < bsx; i++) sl->mb_pred[0][block_y + j][block_x + i] = out[j * bsx + i];
  G.n_mcl++;
}
extern void __real_chroma_prediction_4x4(Macroblock *, int, int, int, int, int, int, short, short, short);
void __wrap_chroma_prediction_4x4(Macroblock *currMB, int uv, int block_x, int block_y, int p_dir, int l0_mode, int l1_mode,
                                  short l0_ref_idx, short l1_ref_idx, short bipred_me)
{
  VideoParameters *p_Vid = currMB->p_Vid;
  Slice *sl = currMB->p_Slice;
  jmhip_mc_chroma_blk b;
  jmhip_mc_weights w;
  uint8_t out[16];
  int mode[2] = {l0_mode, l1_mode}, ref[2] = {l0_ref_idx, l1_ref_idx};
  int l, j, rc, ok = mc_common_ok(currMB, p_dir) && G.part_mcc && p_Vid->p_Inp->ChromaMCBuffer && (uv == 0 || uv == 1);
  memset(&b, 0, sizeof b);
  for (l = 0; ok && l < 2; l++)
    if (p_dir == l || p_dir == 2) {
      StorablePicture *pic = sl->listX[l][ref[l]];
      if (!pic || pic->size_x != G.W || pic->size_y != G.H || pic->chroma_vector_adjustment != 0) ok = 0;
    }
  if (!ok) { G.n_passed++; __real_chroma_prediction_4x4(currMB, uv, block_x, block_y, p_dir, l0_mode, l1_mode, l0_ref_idx, l1_ref_idx, bipred_me); return; }
  {
    MotionVector *****mva = mc_vectors(currMB, p_dir, l0_mode, l1_mode, l0_ref_idx, l1_ref_idx, bipred_me);
    const int rsx = 4 - p_Vid->chroma_shift_x, rsy = 4 - p_Vid->chroma_shift_y;     /* chroma sample -> luma 4x4 block, as JM does it */
    b.x = (int16_t)(currMB->pix_c_x + block_x); b.y = (int16_t)(currMB->opix_c_y + block_y);
    b.dir = (uint8_t)p_dir; b.plane = (uint8_t)uv;
    for (l = 0; l < 2; l++)
      if (p_dir == l || p_dir == 2) {
        MotionVector **mv = mva[l][ref[l]][mode[l]];
        b.slot[l] = (int8_t)slot_with_chroma(sl->listX[l][ref[l]]);
        for (j = 0; j < 4; j++) {
          const MotionVector *a = &mv[(block_y + j) >> rsy][block_x >> rsx], *c = &mv[(block_y + j) >> rsy][(block_x + 2) >> rsx];
          b.mv[l][j][0][0] = a->mv_x; b.mv[l][j][0][1] = a->mv_y; b.mv[l][j][1][0] = c->mv_x; b.mv[l][j][1][1] = c->mv_y;
        }
      }
  }
  if (p_dir == 2 && (G.slot_pic[b.slot[0]] != sl->listX[0][ref[0]] || !G.slot_chroma[b.slot[0]])) b.slot[0] = (int8_t)slot_with_chroma(sl->listX[0][ref[0]]);
  if ((rc = jmhip_mc_chroma_wp(G.ctx, &b, mc_weights(sl, p_dir, l0_ref_idx, l1_ref_idx, uv + 1, &w) ? &w : NULL, 1, out))) adapter_die("jmhip_mc_chroma_wp", rc);
  for (j = 0; j < 4; j++) {
    imgpel *row = &sl->mb_pred[uv + 1][block_y + j][block_x];
    row[0] = out[j * 4]; row[1] = out[j * 4 + 1]; row[2] = out[j * 4 + 2]; row[3] = out[j * 4 + 3];
  }
  G.n_mcc++;
}

/* ------------------------------------------------------------------ the source picture: pad_borders (lcommon/src/input.c:880; image.c:1244)
 * JM has just read the picture (read_one_frame: buf2img into the top-left output.width x output.height of each plane); the device takes
 * those samples as the file laid them out, builds the coded-size planes (k_load_frame: the copy + both borders) and keeps them as the
 * current picture; the padded planes come back into JM's arrays. */
extern void __real_pad_borders(FrameFormat, int, int, int, int, imgpel **[3]);
void __wrap_pad_borders(FrameFormat output, int img_size_x, int img_size_y, int img_size_x_cr, int img_size_y_cr, imgpel **pImage[3])
{
  VideoParameters *p_Vid = p_Enc ? p_Enc->p_Vid : NULL;
  const int sw = output.width[0], sh = output.height[0], fmt = (int)output.yuv_format;
  const int scw = fmt ? output.width[1] : 0, sch = fmt ? output.height[1] : 0;
  uint8_t *raw;
  uint16_t *planes;
  int k, x, y, rc;
  size_t n = 0;
  if (!p_Vid || !adapter_on(p_Vid) || !G.part_load || pipe_config_ok(p_Vid) || fmt != G.fmt || img_size_x != G.W || img_size_y != G.H || sw > G.W || sh > G.H ||
      G.W - sw >= 16 || G.H - sh >= 16 || output.bit_depth[0] != 8 || (fmt && (output.bit_depth[1] != 8 || scw != sw / 2 || (sw & 1))) ||
      (fmt == 1 && (sch != sh / 2 || (sh & 1))) || (fmt == 2 && sch != sh) || (fmt && (img_size_x_cr != G.W / 2 || img_size_y_cr != (fmt == 1 ? G.H / 2 : G.H)))) {
    if (!pipe_active()) G.n_passed++;  /* with the macroblock pipeline JM pads on the host by design: the padded planes go up once per picture */
    __real_pad_borders(output, img_size_x, img_size_y, img_size_x_cr, img_size_y_cr, pImage);
    return;
  }
  T_pad -= now_s();
  raw = (uint8_t *)malloc((size_t)sw * sh + (size_t)2 * scw * sch);
  planes = (uint16_t *)malloc(((size_t)G.W * G.H + (size_t)2 * img_size_x_cr * img_size_y_cr + 1) * sizeof(uint16_t));
  if (!raw || !planes) { fprintf(stderr, "jmhip adapter: out of memory\n"); exit(70); }
  for (k = 0; k < (fmt ? 3 : 1); k++) {
    const int w = k ? scw : sw, h = k ? sch : sh;
    for (y = 0; y < h; y++) for (x = 0; x < w; x++) raw[n++] = (uint8_t)pImage[k][y][x];
  }
  if ((rc = jmhip_set_current_frame(G.ctx, raw, sw, sh))) adapter_die("jmhip_set_current_frame", rc);
  {
    uint16_t *py = planes, *pu = fmt ? planes + (size_t)G.W * G.H : NULL, *pv = fmt ? pu + (size_t)img_size_x_cr * img_size_y_cr : NULL;
    if ((rc = jmhip_get_current_planes(G.ctx, py, pu, pv))) adapter_die("jmhip_get_current_planes", rc);
    for (y = 0; y < G.H; y++) for (x = 0; x < G.W; x++) pImage[0][y][x] = py[(size_t)y * G.W + x];
    for (k = 1; k < (fmt ? 3 : 1); k++) {
      const uint16_t *pc = k == 1 ? pu : pv;
      for (y = 0; y < img_size_y_cr; y++) for (x = 0; x < img_size_x_cr; x++) pImage[k][y][x] = pc[(size_t)y * img_size_x_cr + x];
    }
  }
  free(raw); free(planes);
  T_pad += now_s();
  G.n_load++;
}

/* ------------------------------------------------------------------ K9/K10: deblocking */
extern void __real_DeblockFrame(VideoParameters *, imgpel **, imgpel ***);
void __wrap_DeblockFrame(VideoParameters *p_Vid, imgpel **imgY, imgpel ***imgUV)
{
  StorablePicture *ids[64];
  int nids = 0, rc, x, y, l, k;
  unsigned i;
  { const double t0_ = now_s(); if (adapter_on(p_Vid) && pipe_deblock(p_Vid, imgY, imgUV)) { T_deblock += now_s() - t0_; if (TL_n) { TL_db[TL_n - 1] += now_s() - t0_; TL_db_at[TL_n - 1] = t0_; } return; } }
  if (!adapter_on(p_Vid) || !G.part_deblock || p_Vid->structure != FRAME || p_Vid->mb_aff_frame_flag ||
      (int)p_Vid->PicSizeInMbs != (G.W / 16) * (G.H / 16)) {
    G.n_passed++;
    __real_DeblockFrame(p_Vid, imgY, imgUV);
    return;
  }
  for (i = 0; i < p_Vid->PicSizeInMbs; i++) {               /* what DeblockMb / GetStrength* read from Macroblock (lencod/inc/global.h) */
    Macroblock *m = &p_Vid->mb_data[i];
    jmhip_db_mb *d = &G.dbmb[i];
    d->mb_type = m->mb_type; d->slice_type = (int16_t)m->p_Slice->slice_type; d->qp = (int16_t)m->qp;
    d->qpc[0] = (int16_t)m->qpc[0]; d->qpc[1] = (int16_t)m->qpc[1]; d->cbp = (int16_t)m->cbp;
    d->cbp_blk = (uint32_t)(m->cbp_blk & 0xFFFF); d->slice_nr = (int16_t)m->slice_nr;
    d->df_disable_idc = (int16_t)m->DFDisableIdc; d->df_alpha_c0 = (int16_t)m->DFAlphaC0Offset; d->df_beta = (int16_t)m->DFBetaOffset;
    d->transform8x8 = (int16_t)m->luma_transform_size_8x8_flag;
  }
  for (y = 0; y < G.H / 4; y++)
    for (x = 0; x < G.W / 4; x++) {
      PicMotionParams *mp = &p_Vid->enc_picture->mv_info[y][x];
      jmhip_db_motion *d = &G.dbmo[(size_t)y * (G.W / 4) + x];
      for (l = 0; l < 2; l++) {
        int id = -1;
        if (mp->ref_idx[l] != -1) {                          /* identity of ref_pic[l]: what compare_mvs' callers test (loop_filter_normal.c) */
          for (k = 0; k < nids; k++) if (ids[k] == mp->ref_pic[l]) break;
          if (k == nids && nids < 64) ids[nids++] = mp->ref_pic[l];
          id = k;
        }
        d->mv[l][0] = mp->mv[l].mv_x; d->mv[l][1] = mp->mv[l].mv_y; d->ref_id[l] = id;
      }
    }
  rc = jmhip_deblock_frame(G.ctx, imgY[0], (int)(imgY[1] - imgY[0]),
                           G.fmt ? imgUV[0][0] : NULL, G.fmt ? imgUV[1][0] : NULL, G.fmt ? (int)(imgUV[0][1] - imgUV[0][0]) : 0,
                           G.dbmb, G.dbmo, p_Vid->active_sps->direct_8x8_inference_flag);
  if (rc) adapter_die("jmhip_deblock_frame", rc);
  G.n_deblock++;
}

/* ------------------------------------------------------------------ K7/K8: luma residual transform + quantisation + reconstruction
 * ("dct_4x4()/quant_4x4()" of the north star).  One block per call, as JM's RDO loop asks for them. */
static void fill_tq_common(Macroblock *currMB, int *cavlc, int *around)
{
  *cavlc = currMB->p_Slice->symbol_mode == CAVLC;
  *around = currMB->p_Vid->AdaptiveRounding != 0;
}

extern int __real_residual_transform_quant_luma_4x4(Macroblock *, ColorPlane, int, int, int *, int);
int __wrap_residual_transform_quant_luma_4x4(Macroblock *currMB, ColorPlane pl, int block_x, int block_y, int *coeff_cost, int intra)
{
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currSlice->p_Vid;
  const int pos_x = block_x >> BLOCK_SHIFT, pos_y = block_y >> BLOCK_SHIFT;
  const int b8 = 2 * (pos_y >> 1) + (pos_x >> 1) + (pl << 2), b4 = 2 * (pos_y & 1) + (pos_x & 1);
  imgpel **img_enc = p_Vid->enc_picture->p_curr_img, **mb_pred = currSlice->mb_pred[pl];
  int **mb_ores = currSlice->mb_ores[pl];
  jmhip_tq_params prm;
  jmhip_tq_out out;
  uint8_t orig[16], pred[16];
  int j, i, k, rc, cavlc, around, qp;
  LevelQuantParams **q;
  if (!adapter_on(p_Vid) || !G.part_tq4 || pl != PLANE_Y || currMB->is_field_mode || currSlice->disthres != 0 ||
      (currSlice->quant_4x4 != quant_4x4_normal && currSlice->quant_4x4 != quant_4x4_around)) {
    G.n_passed++;
    return __real_residual_transform_quant_luma_4x4(currMB, pl, block_x, block_y, coeff_cost, intra);
  }
  fill_tq_common(currMB, &cavlc, &around);
  qp = currMB->qp_scaled[pl];
  q = p_Vid->p_Quant->q_params_4x4[pl][intra][qp];
  memset(&prm, 0, sizeof prm);
  for (j = 0; j < 4; j++)
    for (i = 0; i < 4; i++) {
      prm.q[j * 4 + i].OffsetComp = q[j][i].OffsetComp; prm.q[j * 4 + i].ScaleComp = q[j][i].ScaleComp; prm.q[j * 4 + i].InvScaleComp = q[j][i].InvScaleComp;
      pred[j * 4 + i] = (uint8_t)mb_pred[block_y + j][block_x + i];
      orig[j * 4 + i] = (uint8_t)(mb_pred[block_y + j][block_x + i] + mb_ores[block_y + j][block_x + i]);   /* the source sample */
    }
  prm.qp_per = p_Vid->p_Quant->qp_per_matrix[qp]; prm.cavlc = cavlc; prm.adaptive_rounding = currSlice->quant_4x4 == quant_4x4_around;
  prm.adapt_rnd_weight = p_Vid->AdaptRndWeight; prm.max_pel = p_Vid->max_imgpel_value;
  if ((rc = jmhip_tq_luma4x4(G.ctx, &prm, orig, pred, 1, &out))) adapter_die("jmhip_tq_luma4x4", rc);
  G.n_tq4++;
  currMB->subblock_x = ((b8 & 1) == 0) ? (((b4 & 1) == 0) ? 0 : 4) : (((b4 & 1) == 0) ? 8 : 12);     /* block.c:693-694 */
  currMB->subblock_y = (b8 < 2) ? ((b4 < 2) ? 0 : 4) : ((b4 < 2) ? 8 : 12);
  for (k = 0; k < out.ncoef; k++) { currSlice->cofAC[b8][b4][0][k] = out.level[k]; currSlice->cofAC[b8][b4][1][k] = out.run[k]; }
  currSlice->cofAC[b8][b4][0][out.ncoef] = 0;
  *coeff_cost += out.coeff_cost;
  if (out.any_residual && prm.adaptive_rounding) {
    int **fadj = &p_Vid->ARCofAdj4x4[pl][currMB->ar_mode][block_y];
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) fadj[j][block_x + i] = out.fadjust[j * 4 + i];
  }
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) img_enc[currMB->pix_y + block_y + j][currMB->pix_x + block_x + i] = out.rec[j * 4 + i];
  return out.nonzero;
}

/* residual_transform_quant_luma_4x4 is stored into its slot by select_transform() in the SAME translation unit that defines it
 * (block.c:2364-2432), so no link-time reference exists to wrap; the slot itself is rebound right after JM fills it. */
/* residual_transform_quant_chroma_4x4 (block.c:954): one plane of the macroblock per call */
extern int residual_transform_quant_chroma_4x4(Macroblock *, int, int);
static int jmhip_rtq_chroma(Macroblock *currMB, int uv, int cr_cbp)
{
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currSlice->p_Vid;
  const int yuv = p_Vid->yuv_format, H = p_Vid->mb_cr_size_y, intra = is_intra(currMB);
  const int cur_qp = currMB->qpc[uv] + currSlice->bitdepth_chroma_qp_scale, qp_dc = yuv == YUV422 ? cur_qp + 3 : cur_qp;
  const int uv_scale = uv * (p_Vid->num_blk8x8_uv >> 1), nblk = H / 2;
  imgpel **mb_pred = currSlice->mb_pred[uv + 1];
  int **mb_ores = currSlice->mb_ores[uv + 1];
  jmhip_tqc_params prm;
  jmhip_tqc_mb mb;
  jmhip_tqc_out out;
  uint8_t orig[128], pred[128];
  LevelQuantParams **qa, *qd;
  int j, i, k, c, rc, around;
  if (!adapter_on(p_Vid) || !G.part_tqc || (yuv != YUV420 && yuv != YUV422) || currMB->is_field_mode || currSlice->disthres != 0 ||
      p_Vid->mb_cr_size_x != 8 || (currSlice->quant_ac4x4cr != quant_ac4x4_normal && currSlice->quant_ac4x4cr != quant_ac4x4_around)) {
    G.n_passed++;
    return residual_transform_quant_chroma_4x4(currMB, uv, cr_cbp);
  }
  around = currSlice->quant_ac4x4cr == quant_ac4x4_around;
  qa = p_Vid->p_Quant->q_params_4x4[uv + 1][intra][cur_qp];
  qd = &p_Vid->p_Quant->q_params_4x4[uv + 1][intra][qp_dc][0][0];
  memset(&prm, 0, sizeof prm); memset(orig, 0, sizeof orig); memset(pred, 0, sizeof pred);
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) {
    prm.q_ac[j * 4 + i].OffsetComp = qa[j][i].OffsetComp; prm.q_ac[j * 4 + i].ScaleComp = qa[j][i].ScaleComp; prm.q_ac[j * 4 + i].InvScaleComp = qa[j][i].InvScaleComp;
  }
  prm.q_dc.OffsetComp = qd->OffsetComp; prm.q_dc.ScaleComp = qd->ScaleComp; prm.q_dc.InvScaleComp = qd->InvScaleComp;
  prm.qp_per_ac = p_Vid->p_Quant->qp_per_matrix[cur_qp]; prm.qp_per_dc = p_Vid->p_Quant->qp_per_matrix[qp_dc];
  prm.yuv_format = yuv; prm.cavlc = currSlice->symbol_mode == CAVLC; prm.adaptive_rounding = around;
  prm.adapt_rnd_weight = p_Vid->AdaptRndWeight; prm.max_pel = p_Vid->max_pel_value_comp[uv + 1];
  for (j = 0; j < H; j++) for (i = 0; i < 8; i++) { pred[j * 8 + i] = (uint8_t)mb_pred[j][i]; orig[j * 8 + i] = (uint8_t)(mb_pred[j][i] + mb_ores[j][i]); }
  mb.cbp_blk = currMB->cbp_blk; mb.cr_cbp = cr_cbp; mb.uv = uv;
  if ((rc = jmhip_tq_chroma(G.ctx, &prm, &mb, orig, pred, 1, &out))) adapter_die("jmhip_tq_chroma", rc);
  G.n_tqc++;
  p_Vid->is_v_block = uv;                                  /* block.c:1008 */
  currMB->cbp_blk = mb.cbp_blk;
  for (k = 0; k < 9; k++) { currSlice->cofDC[uv + 1][0][k] = out.dc_level[k]; currSlice->cofDC[uv + 1][1][k] = out.dc_run[k]; }
  for (k = 0; k < nblk; k++) {
    int *lev = currSlice->cofAC[4 + (k >> 2) + uv_scale][k & 3][0], *run = currSlice->cofAC[4 + (k >> 2) + uv_scale][k & 3][1];
    for (c = 0; c < 16; c++) { lev[c] = out.ac_level[k][c]; run[c] = out.ac_run[k][c]; }
  }
  /* the last block's coeff_count context position, as JM leaves it (block.c:1108-1109) */
  currMB->subblock_y = (short)(4 * ((nblk - 1) >> 1) & 0xf); currMB->subblock_x = (short)(4 * ((nblk - 1) & 1));
  if (around) {
    int **fadj = (currMB->mb_type == P8x8 && currMB->luma_transform_size_8x8_flag) ? p_Vid->ARCofAdj4x4[uv + 1][4] : p_Vid->ARCofAdj4x4[uv + 1][currMB->ar_mode];
    for (j = 0; j < H; j++) for (i = 0; i < 8; i++) if ((j & 3) || (i & 3)) fadj[j][i] = out.fadjust[j * 8 + i];     /* AC positions only */
  }
  for (j = 0; j < H; j++) for (i = 0; i < 8; i++) p_Vid->enc_picture->imgUV[uv][currMB->pix_c_y + j][currMB->pix_c_x + i] = out.rec[j * 8 + i];
  return mb.cr_cbp;
}

extern void __real_select_transform(Macroblock *);
/* residual_transform_quant_luma_16x16 (block.c:208): the Intra16x16 luma of one macroblock */
extern int residual_transform_quant_luma_16x16(Macroblock *, ColorPlane);
static int jmhip_rtq_luma_16x16(Macroblock *currMB, ColorPlane pl)
{
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currSlice->p_Vid;
  jmhip_tq_params prm;
  static jmhip_tq16_out out;
  uint8_t orig[256], pred[256];
  int j, i, k, b, rc, cavlc, around, qp;
  LevelQuantParams **q;
  if (!adapter_on(p_Vid) || !G.part_tq16 || pl != PLANE_Y || p_Vid->yuv_format == YUV444 || currMB->is_field_mode ||
      currSlice->slice_type == SP_SLICE || currSlice->slice_type == SI_SLICE || currSlice->quant_dc4x4 != quant_dc4x4_normal ||
      (currSlice->quant_ac4x4 != quant_ac4x4_normal && currSlice->quant_ac4x4 != quant_ac4x4_around)) {
    G.n_passed++;
    return residual_transform_quant_luma_16x16(currMB, pl);
  }
  fill_tq_common(currMB, &cavlc, &around);
  qp = currMB->qp_scaled[pl];
  q = p_Vid->p_Quant->q_params_4x4[pl][1][qp];
  memset(&prm, 0, sizeof prm);
  for (j = 0; j < 4; j++)
    for (i = 0; i < 4; i++) { prm.q[j * 4 + i].OffsetComp = q[j][i].OffsetComp; prm.q[j * 4 + i].ScaleComp = q[j][i].ScaleComp; prm.q[j * 4 + i].InvScaleComp = q[j][i].InvScaleComp; }
  prm.qp_per = p_Vid->p_Quant->qp_per_matrix[qp]; prm.cavlc = cavlc; prm.adaptive_rounding = currSlice->quant_ac4x4 == quant_ac4x4_around;
  prm.adapt_rnd_weight = p_Vid->AdaptRndWeight; prm.max_pel = p_Vid->max_imgpel_value;
  for (j = 0; j < 16; j++)
    for (i = 0; i < 16; i++) {
      orig[j * 16 + i] = (uint8_t)p_Vid->pCurImg[currMB->opix_y + j][currMB->pix_x + i];
      pred[j * 16 + i] = (uint8_t)currSlice->mpr_16x16[pl][currMB->i16mode][j][i];
    }
  if ((rc = jmhip_tq_luma16x16(G.ctx, &prm, orig, pred, 1, &out))) adapter_die("jmhip_tq_luma16x16", rc);
  G.n_tq16++;
  currMB->subblock_x = 12; currMB->subblock_y = 12;                          /* what the loop of block.c:310-336 leaves behind */
  for (k = 0; k < 17; k++) { currSlice->cofDC[pl][0][k] = out.dc_level[k]; currSlice->cofDC[pl][1][k] = out.dc_run[k]; }
  for (b = 0; b < 16; b++) {
    int *lev = currSlice->cofAC[(pl << 2) + (b >> 2)][b & 3][0], *run = currSlice->cofAC[(pl << 2) + (b >> 2)][b & 3][1];
    for (k = 0; k < out.ac_ncoef[b]; k++) { lev[k] = out.ac_level[b][k]; run[k] = out.ac_run[b][k]; }
    lev[out.ac_ncoef[b]] = 0;
  }
  if (prm.adaptive_rounding) {                                                /* rows 0..3 only, the DC positions untouched: as JM's call does */
    int **fadj = p_Vid->ARCofAdj4x4[pl][I16MB];
    for (j = 0; j < 4; j++) for (i = 0; i < 16; i++) if (j || (i & 3)) fadj[j][i] = out.fadjust[j][i];
  }
  for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) p_Vid->enc_picture->p_curr_img[currMB->pix_y + j][currMB->pix_x + i] = out.rec[j * 16 + i];
  return out.ac_coef;
}

void __wrap_select_transform(Macroblock *currMB)
{
  __real_select_transform(currMB);
  if (currMB->residual_transform_quant_luma_16x16 == residual_transform_quant_luma_16x16) currMB->residual_transform_quant_luma_16x16 = jmhip_rtq_luma_16x16;
  if (currMB->residual_transform_quant_luma_4x4 == __real_residual_transform_quant_luma_4x4)
    currMB->residual_transform_quant_luma_4x4 = __wrap_residual_transform_quant_luma_4x4;
  if (currMB->residual_transform_quant_chroma_4x4[0] == residual_transform_quant_chroma_4x4) currMB->residual_transform_quant_chroma_4x4[0] = jmhip_rtq_chroma;
  if (currMB->residual_transform_quant_chroma_4x4[1] == residual_transform_quant_chroma_4x4) currMB->residual_transform_quant_chroma_4x4[1] = jmhip_rtq_chroma;
}

static int tq8_common(Macroblock *currMB, ColorPlane pl, int b8, int *coeff_cost, int intra, int cavlc_variant,
                      int (*real)(Macroblock *, ColorPlane, int, int *, int))
{
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currSlice->p_Vid;
  const int block_x = 8 * (b8 & 1), block_y = 8 * (b8 >> 1), pl_off = b8 + (pl << 2);
  imgpel **img_enc = p_Vid->enc_picture->p_curr_img, **mb_pred = currSlice->mb_pred[pl];
  int **mb_ores = currSlice->mb_ores[pl];
  jmhip_tq8_params prm;
  jmhip_tq8_out out;
  uint8_t orig[64], pred[64];
  int j, i, k, l, rc, qp, around;
  LevelQuantParams **q;
  if (cavlc_variant) around = currSlice->quant_8x8cavlc == quant_8x8cavlc_around;
  else around = currSlice->quant_8x8 == quant_8x8_around;
  if (!adapter_on(p_Vid) || !G.part_tq8 || pl != PLANE_Y || currMB->is_field_mode || currSlice->disthres != 0 ||
      (cavlc_variant ? (currSlice->quant_8x8cavlc != quant_8x8cavlc_normal && !around) : (currSlice->quant_8x8 != quant_8x8_normal && !around))) {
    G.n_passed++;
    return real(currMB, pl, b8, coeff_cost, intra);
  }
  qp = currMB->qp_scaled[pl];
  q = p_Vid->p_Quant->q_params_8x8[pl][intra][qp];
  memset(&prm, 0, sizeof prm);
  for (j = 0; j < 8; j++)
    for (i = 0; i < 8; i++) {
      prm.q[j * 8 + i].OffsetComp = q[j][i].OffsetComp; prm.q[j * 8 + i].ScaleComp = q[j][i].ScaleComp; prm.q[j * 8 + i].InvScaleComp = q[j][i].InvScaleComp;
      pred[j * 8 + i] = (uint8_t)mb_pred[block_y + j][block_x + i];
      orig[j * 8 + i] = (uint8_t)(mb_pred[block_y + j][block_x + i] + mb_ores[block_y + j][block_x + i]);
    }
  prm.qp_per = p_Vid->p_Quant->qp_per_matrix[qp]; prm.cavlc = cavlc_variant; prm.adaptive_rounding = around;
  prm.adapt_rnd_weight = p_Vid->AdaptRndWeight; prm.max_pel = p_Vid->max_imgpel_value;
  if ((rc = jmhip_tq_luma8x8(G.ctx, &prm, orig, pred, 1, &out))) adapter_die("jmhip_tq_luma8x8", rc);
  G.n_tq8++;
  if (cavlc_variant) {                                     /* four lists, cofAC[pl_off][k] (transform8x8.c:643) */
    for (l = 0; l < 4; l++) {
      for (k = 0; k < out.ncoef[l]; k++) { currSlice->cofAC[pl_off][l][0][k] = out.level[17 * l + k]; currSlice->cofAC[pl_off][l][1][k] = out.run[17 * l + k]; }
      currSlice->cofAC[pl_off][l][0][out.ncoef[l]] = 0;
    }
  } else {
    for (k = 0; k < out.ncoef[0]; k++) { currSlice->cofAC[pl_off][0][0][k] = out.level[k]; currSlice->cofAC[pl_off][0][1][k] = out.run[k]; }
    currSlice->cofAC[pl_off][0][0][out.ncoef[0]] = 0;
  }
  *coeff_cost += out.coeff_cost;
  if (around && (cavlc_variant || out.any_residual)) {
    int **fadj = &p_Vid->ARCofAdj8x8[pl][currMB->ar_mode][block_y];
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) fadj[j][block_x + i] = out.fadjust[j * 8 + i];
  }
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) img_enc[currMB->pix_y + block_y + j][currMB->pix_x + block_x + i] = out.rec[j * 8 + i];
  return out.nonzero;
}
extern int __real_residual_transform_quant_luma_8x8(Macroblock *, ColorPlane, int, int *, int);
int __wrap_residual_transform_quant_luma_8x8(Macroblock *m, ColorPlane pl, int b8, int *cc, int intra)
{
  return tq8_common(m, pl, b8, cc, intra, 0, __real_residual_transform_quant_luma_8x8);
}
extern int __real_residual_transform_quant_luma_8x8_cavlc(Macroblock *, ColorPlane, int, int *, int);
int __wrap_residual_transform_quant_luma_8x8_cavlc(Macroblock *m, ColorPlane pl, int b8, int *cc, int intra)
{
  return tq8_common(m, pl, b8, cc, intra, 1, __real_residual_transform_quant_luma_8x8_cavlc);
}

/* ------------------------------------------------------------------ the RDO-off macroblock pipeline
 * encode_one_macroblock_low (lencod/src/md_low.c:104; Slice.encode_one_macroblock, bound in rdopt.c:242-260 when RDOptimization = 0).
 * JM calls it macroblock after macroblock from encode_one_slice (slice.c:512).  At a slice's first macroblock the whole slice is encoded
 * on the MI355X (jmhip_encode_slice: motion search, mode decision, transform / quantisation, reconstruction of every macroblock in
 * wavefront order); each call then only unpacks its macroblock's record into the structures JM's own write_macroblock
 * (macroblock.c:2810) reads, so the entropy coder and the bitstream writer stay JM's.  The reconstruction stays on the device:
 * DeblockFrame runs there (jmhip_deblock_picture_dev) and hands the picture back once, getSubImagesLuma / getSubImagesChroma become
 * jmhip_reference_from_recon (no sub-pel plane ever crosses PCIe; the host does no motion compensation in this mode).
 * Part name: mbpipe.  Eligibility is decided once per sequence from the configuration (pipe_config_ok); a sequence is either
 * entirely on this path or not at all. */
#include "mode_decision.h"
#include "macroblock.h"
#include "md_common.h"

static struct {
  int checked, ok;
  StorablePicture *pic;               /* the picture the records / the device reconstruction belong to */
  int mbs;                            /* its macroblocks served so far */
  int deblocked;
  int slice_last;                     /* last macroblock the running launch covers (one slice, or every slice of the picture) */
  int dealt, share_mbs, band_rows;    /* the picture's slices are dealt to G.nctx contexts: macroblocks / macroblock rows per context */
  long n_slices, n_mbs, n_refs;
  double t_dev, t_fill, t_wait;
} P;

static int pipe_active(void) { return G.ctx && !G.off && P.ok; }
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int pipe_config_ok(VideoParameters *p_Vid)
{
  if (!P.checked) {
    InputParameters *p = p_Vid->p_Inp;
    const char *why = NULL;
    const int R = imax(p->search_range[0], p->search_range[1]);
    const int ox = (2 * R + 3) & ~3;
    P.checked = 1;
    if (p_Vid->active_sps && p->PicInterlace == FRAME_CODING && p->MbInterlace == FRAME_CODING) update_mv_limits(p_Vid, FALSE);   /* the level's vector limits (conformance.c:604): JM sets them per picture
                                                                    (image.c:1635), later than the first question about this sequence; frame pictures always get these values */
    if (!G.part_mbpipe) why = "part mbpipe not selected";
    else if (p->rdopt != 0) why = "RDOptimization != 0";
    else if (p_Vid->yuv_format != YUV420 && p_Vid->yuv_format != YUV422) why = "neither 4:2:0 nor 4:2:2";
    else if (p->SearchMode[0] != FULL_SEARCH && p->SearchMode[0] != FAST_FULL_SEARCH && p->SearchMode[0] != EPZS) why = "SearchMode other than -1 (full search), 0 (fast full search) and 3 (EPZS)";
    else if (p->SearchMode[0] == EPZS && (!p->EPZSSubPelGrid || p->EPZSSubPelME != 1 || p->HMEEnable)) why = "EPZS without EPZSSubPelGrid = 1 / EPZSSubPelME = 1, or with HME";
    else if (p->SearchMode[0] == FAST_FULL_SEARCH && (p_Vid->MaxVmvR[5] - 4 * R < 4 * R || p_Vid->MaxVmvR[4] + 4 * R > -4 * R || p_Vid->MaxHmvR[5] - 4 * R < 4 * R || p_Vid->MaxHmvR[4] + 4 * R > -4 * R))
      why = "fast full search with a level whose vector limit cuts into the search range (the search centre can leave the sample grid: me_fullfast.c:325-326, and JM then reads a stale pos_00)";
    else if (p_Vid->MaxVmvR[4] > -4 * R || p_Vid->MaxVmvR[5] < 4 * R || p_Vid->MaxHmvR[4] > -4 * R || p_Vid->MaxHmvR[5] < 4 * R)
      why = "vector limits (UseMVLimits / the level) narrower than the search range";      /* jmhip_encode_slice refuses such a slice */
    else if (p->DisableMEPrediction) why = "DisableMEPrediction";
    else if (p->SkipDeBlockNonRef || p->DisposableP) why = "SkipDeBlockNonRef / DisposableP (pictures that are not deblocked stay on the device)";
    else if (p->RDPictureDeblocking || p->RDPictureDecision) why = "RDPictureDecision / RDPictureDeblocking";
    else if (p->redundant_pic_flag) why = "UseRedundantPicture";
    else if (p->Transform8x8Mode != 0 && p->Transform8x8Mode != 1) why = "Transform8x8Mode 2 (8x8 transform only)";
    else if (p->Transform8x8Mode == 1 && (p->ScalingMatrixPresentFlag || p->RDOQ_CP_MV || !p->InterSearch[0][0][4])) why = "Transform8x8Mode with scaling matrices / RDOQ_CP_MV / without the 8x8 partition";
    else if (p->AdaptiveRounding != 0) why = "AdaptiveRounding";
    else if (p->WeightedPrediction || p->WeightedBiprediction) why = "weighted prediction";
    else if (p->RCEnable) why = "rate control";
    else if (p->slice_mode != NO_SLICES && p->slice_mode != FIXED_MB) why = "SliceMode > 1";
    else if (p->num_slice_groups_minus1 != 0) why = "FMO";
    else if (p->UseConstrainedIntraPred) why = "UseConstrainedIntraPred";
    else if (p->RestrictRef || p->UseRDOQuant || p->ChromaMEEnable) why = "RestrictRefFrames / UseRDOQuant / ChromaMEEnable";
    else if (p->MEErrorMetric[F_PEL] != ERROR_SAD || p->MEErrorMetric[H_PEL] != ERROR_SATD || p->MEErrorMetric[Q_PEL] != ERROR_SATD || p->ModeDecisionMetric != ERROR_SATD) why = "distortion metrics other than SAD / SATD / SATD / SATD";
    else if (p->disthres != 0) why = "DisableThresholding";
    else if (p->Intra4x4ParDisable || p->Intra4x4DiagDisable || p->Intra4x4DirDisable || p->Intra16x16ParDisable || p->Intra16x16PlaneDisable || p->ChromaIntraDisable) why = "intra mode restrictions";
    else if (!p->FastCrIntraDecision) why = "FastCrIntraDecision = 0";
    else if (p->SkipIntraInInterSlices || p->SelectiveIntraEnable || p->RandomIntraMBRefresh || p->intra_upd || p->CtxAdptLagrangeMult) why = "intra refresh / selective intra / CtxAdptLagrangeMult";
    else if (p->sp_periodicity != 0) why = "SP pictures";
    /* B pictures (mbpipe_b.inc): non-reference B pictures between the P pictures, spatial direct mode, the lists as init_lists leaves them */
    else if (p->NumberBFrames != 0 && (p->BRefPictures || p->LowDelay || p->ExplicitSeqCoding || p->EnableOpenGOP))
      why = "B pictures with BReferencePictures, LowDelay, ExplicitSeqCoding or EnableOpenGOP";
    else if (p->NumberBFrames != 0 && !p->direct_spatial_mv_pred_flag && !p->directInferenceFlag) why = "B pictures with DirectModeType 0 (temporal) and DirectInferenceFlag 0";
    else if (p->NumberBFrames != 0 && p->SearchMode[0] == EPZS) why = "B pictures with EPZS";
    else if (p->NumberBFrames != 0 && p->BiPredMotionEstimation && (p->BiPredSearch[3] || p->BiPredMESearchRange[0] < 1 || p->BiPredMESearchRange[0] > 16 || p->BiPredMESubPel > 2 || p->BiPredMERefinements > 15))
      why = "BiPredMotionEstimation with BiPredSearch8x8, or a search range / sub-pel level / refinement count outside the device's";
    else if (p->full_search != 2) why = "RestrictSearchRange != 2";
    else if (p->num_of_views != 1 || p->separate_colour_plane_flag) why = "MVC / separate colour planes";
    else if (!p->InterSearch[0][0][0]) why = "PSliceSkip = 0";
    else if (p->HierarchicalCoding || p->PicInterlace != FRAME_CODING || p->MbInterlace != FRAME_CODING) why = "hierarchical / interlaced coding";
    else if (R < 1 || R > 32) why = "SearchRange outside 1..32";
    else if (p_Vid->max_num_references > JMHIP_MB_MAX_REF) why = "more than 16 references";
    else if (p->SearchMode[0] != EPZS && (size_t)p_Vid->max_num_references * (16 + 4 * R) * (2 * ox + 20) + 20 * (2 * ox + 20) + 50 * 1024 > 160 * 1024) why = "references x search window beyond the LDS";
    else if (p_Vid->bitdepth_chroma_qp_scale != 0) why = "chroma QP scale";
    P.ok = why == NULL;
    if (!P.ok && G.part_mbpipe && p->rdopt == 0)
      fprintf(stderr, "jmhip adapter: macroblock pipeline not used (%s): JM's own encode_one_macroblock_low runs\n", why);
  }
  return P.ok;
}

/* (level, run) list of JM from levels at their scan positions */
static void list_from_dense(const int16_t *dense, int first, int n, int *level, int *run)
{
  int k, c = 0, r = 0;
  for (k = first; k < n; k++) {
    if (dense[k]) { level[c] = dense[k]; run[c] = r; c++; r = 0; }
    else r++;
  }
  level[c] = 0; run[c] = 0;
}

/* ---- pictures in flight: see the note at struct F */
extern int __real_read_one_frame(VideoParameters *, VideoDataFile *, int, int, FrameFormat *, FrameFormat *, imgpel **[3]);
int __wrap_read_one_frame(VideoParameters *p_Vid, VideoDataFile *input_file, int FrameNoInFile, int HeaderSize, FrameFormat *source, FrameFormat *output, imgpel **pImage[3])
{
  /* where JM finds its frames (ReadFrameConcatenated lcommon/src/io_raw.c:103: one file, frame n at HeaderSize + frame bytes x (n + start_frame)): the pictures launched
   * ahead of time are read from the same descriptor with pread; what they read is checked against JM's own planes when JM gets there */
  InputParameters *p_Inp = p_Vid->p_Inp;
  const int fmt = (int)source->yuv_format;
  F.have_file = input_file == &p_Inp->input_file1 && input_file->is_concatenated && !input_file->is_interleaved && input_file->vdtype == VIDEO_YUV && input_file->f_num >= 0 &&
                source->pic_unit_size_shift3 == 1 && source->bit_depth[0] == 8 && output->bit_depth[0] == 8 && (fmt == YUV420 || fmt == YUV422) &&
                source->width[0] == output->width[0] && source->height[0] == output->height[0] && source->color_model != CM_RGB && p_Inp->ProcessInput == 0;
  if (F.have_file) {
    F.fd = input_file->f_num; F.header = HeaderSize; F.start_frame = p_Inp->start_frame; F.cur_frame_no = FrameNoInFile; F.frame_step = 1 + p_Inp->frame_skip;
    F.src_w = source->width[0]; F.src_h = source->height[0];
    F.frame_bytes = (long)source->size_cmp[0] + 2 * (long)source->size_cmp[1];
  }
  return __real_read_one_frame(p_Vid, input_file, FrameNoInFile, HeaderSize, source, output, pImage);
}

static void flight_begin(VideoParameters *p_Vid)
{
  int rc, k;
  if (F.on || !F.depth || P.n_slices) return;               /* decided at the sequence's first picture */
  F.total = p_Vid->p_Inp->no_frames;
  init_prof("flight_begin: before jmhip_seq_open");
  if ((rc = jmhip_seq_open(G.ctx, F.depth, 0))) adapter_die("jmhip_seq_open", rc);
  init_prof("flight_begin: jmhip_seq_open done");
  /* B pictures: several times a P picture's time per macroblock, and no launch waits for them inside the device -- twice a P picture's workgroups (1080p, I P B P B ..., eight
     in flight: 24 ms per picture against 28 with equal shares; profiles/r05_b_in_flight.txt) */
  if (F.nb && (rc = jmhip_seq_b_workgroups(G.ctx, 64))) adapter_die("jmhip_seq_b_workgroups", rc);
  for (k = 0; k < F.depth; k++) F.sub[k].valid = 0;
  F.on = 1;
}

/* do JM's source planes hold exactly the frame that was sent ahead of time? */
static int flight_source_equal(VideoParameters *p_Vid, const uint8_t *raw)
{
  const int cw = F.src_w / 2, chh = p_Vid->yuv_format == YUV422 ? F.src_h : F.src_h / 2;
  int y, x, uv;
  if (!raw) return 0;
  for (y = 0; y < F.src_h; y++) {
    const imgpel *row = p_Vid->pCurImg[y];
    const uint8_t *q = raw + (size_t)y * F.src_w;
    int bad = 0;
    for (x = 0; x < F.src_w; x++) bad |= row[x] ^ q[x];
    if (bad) return 0;
  }
  for (uv = 0; uv < 2; uv++)
    for (y = 0; y < chh; y++) {
      const imgpel *row = p_Vid->pImgOrg[1 + uv][y];
      const uint8_t *q = raw + (size_t)F.src_w * F.src_h + (size_t)uv * cw * chh + (size_t)y * cw;
      int bad = 0;
      for (x = 0; x < cw; x++) bad |= row[x] ^ q[x];
      if (bad) return 0;
    }
  return 1;
}

/* place in display order of the n-th coded picture of the regular order I P B.. P B.. (NumberBFrames nb) */
static long fl_disp(long n, int nb)
{
  if (n == 0 || nb == 0) return n;
  { const long g = (n - 1) / (nb + 1), i = (n - 1) % (nb + 1); return i == 0 ? (g + 1) * (nb + 1) : g * (nb + 1) + i; }
}
/* a reference picture enters the sliding window (most recent first, `window` of them) */
static void fl_store(int *slot, long *disp, int *n, int window, int s, long d)
{
  int a;
  if (window > JMHIP_MB_MAX_REF) window = JMHIP_MB_MAX_REF;
  for (a = imin(*n, window - 1); a > 0; a--) { slot[a] = slot[a - 1]; disp[a] = disp[a - 1]; }
  slot[0] = s; disp[0] = d;
  if (*n < window) (*n)++;
}

/* picture F.pic with the parameters JM really has: take the launch made ahead of time if it was given exactly these, else launch now; then launch ahead */
static void flight_launch(VideoParameters *p_Vid, const jmhip_slice_params *prm)
{
  InputParameters *p_Inp = p_Vid->p_Inp;
  const long k = F.pic;
  const int e = F.cur_entry, d8 = p_Vid->active_sps->direct_8x8_inference_flag;
  int rc, j, hit;
  hit = F.sub[e].valid && F.sub[e].pic == k && F.sub[e].slot == F.cur_slot && !memcmp(&F.sub[e].prm, prm, sizeof *prm) && flight_source_equal(p_Vid, F.sub[e].raw);
  if (hit) F.n_hit++;
  else {
    if (F.sub[e].valid && F.sub[e].pic == k && getenv("JMHIP_ADAPTER_DEBUG")) {          /* why a launch made ahead of time is not used */
      size_t o = 0;
      while (o < sizeof *prm && ((const unsigned char *)&F.sub[e].prm)[o] == ((const unsigned char *)prm)[o]) o++;
      fprintf(stderr, "jmhip adapter: picture %ld (slice type %d) launched ahead of time is void: slot %d / %d, parameters differ from byte %zu of %zu (num_ref %d / %d, num_ref1 %d / %d, "
                      "ref_slot %d %d %d %d / %d %d %d %d), source %s\n", k, prm->slice_type, F.sub[e].slot, F.cur_slot, o, sizeof *prm, F.sub[e].prm.num_ref, prm->num_ref, F.sub[e].prm.num_ref1, prm->num_ref1,
              F.sub[e].prm.ref_slot[0], F.sub[e].prm.ref_slot[1], F.sub[e].prm.ref_slot[2], F.sub[e].prm.ref_slot[3], prm->ref_slot[0], prm->ref_slot[1], prm->ref_slot[2], prm->ref_slot[3],
              flight_source_equal(p_Vid, F.sub[e].raw) ? "equal" : "differs");
    }
    for (j = 0; j < F.depth; j++)                           /* whatever is in flight was built on a picture that is not this one -- unless this one is a B picture: nothing is built on those */
      if (F.sub[j].valid && (F.sub[j].pic == k || (F.sub[j].pic > k && prm->slice_type != B_SLICE))) { (void)jmhip_seq_wait(G.ctx, j); F.sub[j].valid = 0; F.n_void++; }
    rc = jmhip_seq_set_planes(G.ctx, e, p_Vid->pCurImg[0], (int)(p_Vid->pCurImg[1] - p_Vid->pCurImg[0]),
                              p_Vid->pImgOrg[1][0], p_Vid->pImgOrg[2][0], (int)(p_Vid->pImgOrg[1][1] - p_Vid->pImgOrg[1][0]));
    if (rc) adapter_die("jmhip_seq_set_planes", rc);
    G.n_cur++;
    if (TL_n <= 2) init_prof("flight_launch: jmhip_seq_set_planes done");
    if ((rc = jmhip_seq_encode(G.ctx, e, prm, F.cur_slot, d8, 1, NULL))) adapter_die("jmhip_seq_encode", rc);
    if (TL_n <= 2) init_prof("flight_launch: jmhip_seq_encode done");
    F.sub[e].valid = 1; F.sub[e].pic = k; F.sub[e].slot = F.cur_slot; F.sub[e].prm = *prm;
  }
  F.tmpl[prm->slice_type] = *prm; F.have_tmpl[prm->slice_type] = 1;
  F.poc_step = k ? prm->poc_cur - F.poc_last : 2 * (1 + p_Inp->frame_skip);
  F.poc_last = prm->poc_cur; F.slot_poc[F.cur_slot] = prm->poc_cur;
  /* ---- where the sequence stands: the picture's place in display order (from the frame JM read for it), the stored reference pictures as the sliding window now holds them */
  if (k == 0) { F.poc0 = p_Vid->enc_picture->poc; F.first_frame_no = F.cur_frame_no; F.regular = 1; F.window = imax(1, (int)p_Vid->active_sps->num_ref_frames); F.st_n = 0; F.refcount = F.bcount = 0; }
  {
    const long disp = F.frame_step > 0 ? (F.cur_frame_no - F.first_frame_no) / F.frame_step : k;
    if (disp != fl_disp(k, F.nb)) F.regular = 0;            /* (the sequence's tail: a last group of fewer pictures) -- nothing more is launched ahead of time */
    if (prm->slice_type == B_SLICE) F.bcount++;
    else {
      if (p_Vid->currentPicture && p_Vid->currentPicture->idr_flag) F.st_n = 0;
      fl_store(F.st_slot, F.st_disp, &F.st_n, F.window, F.cur_slot, disp);
      F.refcount++;
    }
  }
  /* ---- the next pictures: the frame from the file, the parameters of the last picture of the type the picture will have (IntraPeriod / IDRPeriod as get_idr_flag /
   * set_slice_type apply them without B pictures; with B pictures the regular order I P B.. P B.., the lists as init_lists_p_slice / init_lists_b_slice (list.c) build them in
   * a sliding window; a wrong guess is found out above and costs the launch, never a byte) */
  {
    int sim_n = F.st_n, sim_slot[JMHIP_MB_MAX_REF + 1];
    long sim_disp[JMHIP_MB_MAX_REF + 1], sim_ref = F.refcount, sim_b = F.bcount;
    memcpy(sim_slot, F.st_slot, sizeof sim_slot); memcpy(sim_disp, F.st_disp, sizeof sim_disp);
    for (j = 1; j < F.depth && F.have_file && F.regular; j++) {
      const long n = k + j, d = fl_disp(n, F.nb);
      const int en = (int)(n % F.depth);
      const int isb = F.nb && (n - 1) % (F.nb + 1) != 0;
      const int intra = !F.nb && ((p_Inp->intra_period > 0 && n % p_Inp->intra_period == 0) || (p_Inp->idr_period > 0 && n % p_Inp->idr_period == 0));
      const int type = isb ? B_SLICE : intra ? I_SLICE : P_SLICE;
      const int sl = isb ? F.nring + (int)(sim_b % F.nring_b) : (int)(sim_ref % F.nring);
      jmhip_slice_params q;
      int r, nref, ok = 1;
      if (n >= F.total || d >= F.total) break;
      if (!F.nb && p_Inp->idr_period > 0 && n % p_Inp->idr_period == 0) sim_n = 0;
      if (!(F.sub[en].valid && F.sub[en].pic == n)) {         /* (else: launched already) */
        if (!F.have_tmpl[type]) ok = 0;
        else {
          q = F.tmpl[type];
          if (type == P_SLICE) {
            /* list 0 of a P picture in a sliding window: the reference pictures before it, most recent first */
            const int cap = p_Inp->P_List0_refs[0] ? imin(p_Inp->P_List0_refs[0], p_Vid->max_num_references) : p_Vid->max_num_references;
            nref = imin(cap, sim_n);
            if (nref < 1) ok = 0;
            q.num_ref = nref;
            for (r = 0; r < JMHIP_MB_MAX_REF; r++) { q.ref_slot[r] = r < nref ? sim_slot[r] : 0; q.ref_id[r] = q.ref_slot[r]; }
          } else if (type == B_SLICE) {
            /* list 0: the stored pictures before this one in display order, nearest first, then those after it, nearest first; list 1 the other way round; cut to
             * B_List0_refs / B_List1_refs */
            int l0[JMHIP_MB_MAX_REF + 1], l1[JMHIP_MB_MAX_REF + 1], n0 = 0, n1 = 0, a, pass;
            const int cap0 = p_Inp->B_List0_refs[0] ? imin(p_Inp->B_List0_refs[0], p_Vid->max_num_references) : p_Vid->max_num_references;
            const int cap1 = p_Inp->B_List1_refs[0] ? imin(p_Inp->B_List1_refs[0], p_Vid->max_num_references) : p_Vid->max_num_references;
            for (pass = 0; pass < 2; pass++) {                /* pass 0: before (descending), pass 1: after (ascending); the stored pictures are few: selection by scanning */
              long last = pass == 0 ? d : d;
              for (;;) {
                int best = -1;
                for (a = 0; a < sim_n; a++) {
                  if (pass == 0 ? (sim_disp[a] < last && (best < 0 || sim_disp[a] > sim_disp[best])) : (sim_disp[a] > last && (best < 0 || sim_disp[a] < sim_disp[best]))) best = a;
                }
                if (best < 0) break;
                last = sim_disp[best];
                if (pass == 0) l0[n0++] = sim_slot[best]; else l1[n1++] = sim_slot[best];
              }
            }
            {                                                 /* l0 = before + after, l1 = after + before */
              int b0 = n0, a1 = n1;
              for (a = 0; a < a1; a++) l0[n0++] = l1[a];
              for (a = 0; a < b0; a++) l1[n1++] = l0[a];
            }
            if (n1 > 1 && n0 == n1) {                         /* the two lists equal (no picture on one side): list 1's first two entries change places (list.c) */
              int same = 1;
              for (a = 0; a < n0; a++) same &= l0[a] == l1[a];
              if (same) { const int t = l1[0]; l1[0] = l1[1]; l1[1] = t; }
            }
            n0 = imin(n0, cap0); n1 = imin(n1, cap1);
            if (n0 < 1 || n1 < 1 || n0 + n1 > JMHIP_MB_MAX_REF) ok = 0;
            q.num_ref = n0; q.num_ref1 = n1;
            for (r = 0; r < JMHIP_MB_MAX_REF; r++) { q.ref_slot[r] = r < n0 ? l0[r] : r < n0 + n1 ? l1[r - n0] : 0; q.ref_id[r] = q.ref_slot[r]; }
            if (q.b_switches & 32) {                          /* temporal direct: picture order counts in step with the display order (2 (1 + FrameSkip) per picture: set_poc) */
              const int unit = 2 * (1 + p_Inp->frame_skip);
              q.poc_cur = F.poc0 + (int)d * unit;
              for (r = 0; r < JMHIP_MB_MAX_REF; r++) {
                int a2, pd = 0;
                for (a2 = 0; a2 < sim_n; a2++) if (r < n0 + n1 && sim_slot[a2] == q.ref_slot[r]) pd = F.poc0 + (int)sim_disp[a2] * unit;
                q.poc_ref[r] = r < n0 + n1 ? pd : 0;
              }
            }
          }
        }
        if (ok && q.search_mode == 3) {                       /* EPZS scales its predictors by picture distances: the picture's own count continues the last step, its references' are the ring's */
          q.poc_cur = F.poc_last + j * F.poc_step;
          for (r = 0; r < JMHIP_MB_MAX_REF; r++) q.poc_ref[r] = (type == P_SLICE && r < q.num_ref) ? F.slot_poc[q.ref_slot[r]] : 0;
          F.slot_poc[sl] = q.poc_cur;
        }
        if (ok) {
          if (!F.sub[en].raw && !(F.sub[en].raw = (uint8_t *)malloc((size_t)F.frame_bytes))) { fprintf(stderr, "jmhip adapter: out of memory\n"); exit(70); }
          if (pread(F.fd, F.sub[en].raw, (size_t)F.frame_bytes, (off_t)(F.header + F.frame_bytes * (F.first_frame_no + d * (long)F.frame_step + F.start_frame))) != (ssize_t)F.frame_bytes) break;
          if ((rc = jmhip_seq_set_frame(G.ctx, en, F.sub[en].raw, F.src_w, F.src_h))) adapter_die("jmhip_seq_set_frame", rc);
          if ((rc = jmhip_seq_encode(G.ctx, en, &q, sl, d8, 1, NULL))) adapter_die("jmhip_seq_encode (ahead of time)", rc);
          F.sub[en].valid = 1; F.sub[en].pic = n; F.sub[en].slot = sl; F.sub[en].prm = q;
          F.n_ahead++;
        } else if (!isb) break;                               /* a reference picture that cannot be launched: the pictures behind it need it */
      }
      if (isb) sim_b++;
      else { fl_store(sim_slot, sim_disp, &sim_n, F.window, sl, d); sim_ref++; }
    }
  }
  F.pic++;
}

static void pipe_run_slice(Macroblock *currMB)
{
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currMB->p_Vid;
  InputParameters *p_Inp = currMB->p_Inp;
  QuantParameters *p_Quant = p_Vid->p_Quant;
  jmhip_slice_params prm;
  RD_PARAMS enc_mb;
  const int first = currMB->mbAddrX, left = (int)p_Vid->PicSizeInMbs - first;
  int r, rc, intra, uv, j, i, m;
  double t0 = now_s();
  if (P.pic == p_Vid->enc_picture && first > 0 && first <= P.slice_last) { P.n_slices++; return; }      /* launched together with the picture's first slice */
  if (P.pic != p_Vid->enc_picture || first == 0) {          /* a new picture: its source planes go up once */
    P.pic = p_Vid->enc_picture; P.mbs = 0; P.deblocked = 0;
    flight_begin(p_Vid);
    if (F.on) {                                             /* the picture's place in the ring of slots; its source goes up in flight_launch (or went up ahead of time) */
      int s;
      F.cur_entry = (int)(F.pic % F.depth);
      F.cur_slot = currSlice->slice_type == B_SLICE ? F.nring + (int)(F.bcount % F.nring_b) : (int)(F.refcount % F.nring);
      for (s = 0; s < G.nslots; s++) if (G.slot_pic[s] == p_Vid->enc_picture) G.slot_pic[s] = NULL;
      G.slot_pic[F.cur_slot] = p_Vid->enc_picture; G.slot_tick[F.cur_slot] = ++G.tick; G.slot_chroma[F.cur_slot] = 2;
    } else {
      rc = jmhip_set_current_planes(G.ctx, p_Vid->pCurImg[0], (int)(p_Vid->pCurImg[1] - p_Vid->pCurImg[0]),
                                    p_Vid->pImgOrg[1][0], p_Vid->pImgOrg[2][0], (int)(p_Vid->pImgOrg[1][1] - p_Vid->pImgOrg[1][0]));
      if (rc) adapter_die("jmhip_set_current_planes", rc);
      G.n_cur++;
    }
  }
  memset(&prm, 0, sizeof prm);
  init_enc_mb_params(currMB, &enc_mb, currSlice->slice_type == I_SLICE);
  prm.slice_type = currSlice->slice_type;
  prm.first_mb = first;
  prm.num_mb = p_Inp->slice_mode == FIXED_MB ? imin(p_Inp->slice_argument, left) : left;
  prm.slice_nr = currMB->slice_nr;
  prm.qp = currMB->qp; prm.qpc = currMB->qpc[0];
  prm.qpc_cr_delta = currMB->qpc[1] - currMB->qpc[0];        /* CbQPOffset != CrQPOffset (High profiles) */
  prm.search_range = p_Vid->searchRange.max_x >> 2;
  prm.num_ref = currSlice->slice_type == I_SLICE ? 0 : currSlice->listXsize[LIST_0];
  for (r = 0; r < prm.num_ref; r++) {
    prm.ref_slot[r] = F.on ? slot_find(currSlice->listX[LIST_0][r]) : slot_with_chroma(currSlice->listX[LIST_0][r]);
    if (prm.ref_slot[r] < 0) { fprintf(stderr, "jmhip adapter: pictures in flight: reference %d of picture %ld is not a picture the device holds\n", r, F.pic); exit(70); }
    prm.ref_id[r] = prm.ref_slot[r];
  }
  if (currSlice->slice_type == B_SLICE) {                     /* list 1 behind list 0; the switches of the B slices' decision (include/jmhip.h: b_switches) */
    prm.num_ref1 = currSlice->listXsize[LIST_1];
    for (r = 0; r < prm.num_ref1; r++) {
      prm.ref_slot[prm.num_ref + r] = F.on ? slot_find(currSlice->listX[LIST_1][r]) : slot_with_chroma(currSlice->listX[LIST_1][r]);
      if (prm.ref_slot[prm.num_ref + r] < 0) { fprintf(stderr, "jmhip adapter: pictures in flight: list-1 reference %d of picture %ld is not a picture the device holds\n", r, F.pic); exit(70); }
      prm.ref_id[prm.num_ref + r] = prm.ref_slot[prm.num_ref + r];
    }
    prm.b_switches = (p_Vid->active_sps->direct_8x8_inference_flag ? 1 : 0) | (currSlice->direct_spatial_mv_pred_flag ? 0 : 32);
    if (!currSlice->direct_spatial_mv_pred_flag) {            /* temporal direct: the picture distances of compute_colocated (mbuffer.c:3122) */
      prm.poc_cur = p_Vid->enc_picture->poc;
      for (r = 0; r < prm.num_ref; r++) prm.poc_ref[r] = currSlice->listX[LIST_0][r]->poc;
      for (r = 0; r < prm.num_ref1; r++) prm.poc_ref[prm.num_ref + r] = currSlice->listX[LIST_1][r]->poc;
    }
    if (p_Inp->BiPredMotionEstimation)
      prm.b_switches |= 2 | (p_Inp->BiPredSearch[0] ? 4 : 0) | (p_Inp->BiPredSearch[1] ? 8 : 0) | (p_Inp->BiPredSearch[2] ? 16 : 0) | ((p_Inp->BiPredMERefinements & 15) << 8) |
                        ((p_Inp->BiPredMESearchRange[0] & 255) << 16) | ((p_Inp->BiPredMESubPel & 3) << 24);
  }
  for (m = 0; m < 3; m++) prm.lambda_mf[m] = enc_mb.lambda_mf[m];
  prm.lambda_mdfp = enc_mb.lambda_mdfp;
  prm.max_mvd = p_Vid->max_mvd;
  prm.mv_limit[0] = p_Vid->MaxHmvR[4]; prm.mv_limit[1] = p_Vid->MaxHmvR[5]; prm.mv_limit[2] = p_Vid->MaxVmvR[4]; prm.mv_limit[3] = p_Vid->MaxVmvR[5];
  for (m = 0; m < 8; m++) prm.inter_valid[m] = enc_mb.valid[m];
  prm.intra4_valid = enc_mb.valid[I4MB]; prm.intra16_valid = enc_mb.valid[I16MB];
  prm.subpel = !p_Inp->DisableSubpelME[0];
  prm.start_qp = p_Vid->start_me_refinement_qp;
  if (p_Vid->start_me_refinement_hp != 0) { fprintf(stderr, "jmhip adapter: macroblock pipeline: start_me_refinement_hp != 0\n"); exit(70); }
  for (r = 0; r < JMHIP_MB_MAX_REF; r++) prm.refbits[r] = p_Vid->refbits[r];   /* at least 31 entries (mv_search.c:323-345); the sub-macroblock types' cost reads [0..3] */
  for (intra = 0; intra < 2; intra++)
    for (j = 0; j < 4; j++)
      for (i = 0; i < 4; i++) {
        const LevelQuantParams *q = &p_Quant->q_params_4x4[0][intra][currMB->qp_scaled[0]][j][i];
        prm.q_luma[intra][j * 4 + i].OffsetComp = q->OffsetComp; prm.q_luma[intra][j * 4 + i].ScaleComp = q->ScaleComp; prm.q_luma[intra][j * 4 + i].InvScaleComp = q->InvScaleComp;
        for (uv = 0; uv < 2; uv++) {
          q = &p_Quant->q_params_4x4[uv + 1][intra][currMB->qpc[uv] + currSlice->bitdepth_chroma_qp_scale][j][i];
          prm.q_chroma[uv][intra][j * 4 + i].OffsetComp = q->OffsetComp; prm.q_chroma[uv][intra][j * 4 + i].ScaleComp = q->ScaleComp; prm.q_chroma[uv][intra][j * 4 + i].InvScaleComp = q->InvScaleComp;
        }
      }
  if (p_Vid->yuv_format == YUV422)                          /* the 2x4 chroma DC transform is quantised with the parameters of qpc + 3 (block.c:1059-1063) */
    for (intra = 0; intra < 2; intra++)
      for (uv = 0; uv < 2; uv++) {
        const LevelQuantParams *q = &p_Quant->q_params_4x4[uv + 1][intra][currMB->qpc[uv] + 3 + currSlice->bitdepth_chroma_qp_scale][0][0];
        prm.q_chroma_dc[uv][intra].OffsetComp = q->OffsetComp; prm.q_chroma_dc[uv][intra].ScaleComp = q->ScaleComp; prm.q_chroma_dc[uv][intra].InvScaleComp = q->InvScaleComp;
      }
  prm.symbol_mode = currSlice->symbol_mode == CABAC;        /* the entropy coder stays JM's; the quantiser clamps levels for CAVLC only */
  if (p_Inp->Transform8x8Mode == 1) {                       /* High profile: the 8x8 transform beside the 4x4 one, Intra8x8 */
    prm.transform8x8 = 1;
    prm.intra8_valid = enc_mb.valid[I8MB];
    for (intra = 0; intra < 2; intra++)
      for (j = 0; j < 8; j++)
        for (i = 0; i < 8; i++) {
          const LevelQuantParams *q = &p_Quant->q_params_8x8[0][intra][currMB->qp_scaled[0]][j][i];
          prm.q_luma8[intra][j * 8 + i].OffsetComp = q->OffsetComp; prm.q_luma8[intra][j * 8 + i].ScaleComp = q->ScaleComp; prm.q_luma8[intra][j * 8 + i].InvScaleComp = q->InvScaleComp;
        }
  }
  if (p_Inp->SearchMode[0] == FAST_FULL_SEARCH) prm.search_mode = 1;
  if (p_Inp->SearchMode[0] == EPZS) {                       /* EPZSStructInit / EPZSSliceInit read these (me_epzs_common.c:423, :620) */
    prm.search_mode = 3;
    prm.epzs_pattern = p_Inp->EPZSPattern; prm.epzs_dual = p_Inp->EPZSDual; prm.epzs_fixed = p_Inp->EPZSFixed; prm.epzs_aggressive = p_Inp->EPZSAggressiveWindow;
    prm.epzs_temporal = p_Inp->EPZSTemporal[0]; prm.epzs_spatial_mem = p_Inp->EPZSSpatialMem; prm.epzs_blocktype = p_Inp->EPZSBlockType;
    prm.epzs_min_scale = p_Inp->EPZSMinThresScale[0]; prm.epzs_med_scale = p_Inp->EPZSMedThresScale[0]; prm.epzs_max_scale = p_Inp->EPZSMaxThresScale[0];
    prm.epzs_sub_scale = p_Inp->EPZSSubPelThresScale[0];
    prm.poc_cur = p_Vid->enc_picture->poc;
    for (r = 0; r < prm.num_ref; r++) prm.poc_ref[r] = currSlice->listX[LIST_0][r]->poc;
  }
  prm.df_disable_idc = currMB->DFDisableIdc; prm.df_alpha_c0 = currMB->DFAlphaC0Offset; prm.df_beta = currMB->DFBetaOffset;
  if (currMB->qp_scaled[0] != currMB->qp) { fprintf(stderr, "jmhip adapter: macroblock pipeline: luma QP scale\n"); exit(70); }
  /* SliceMode 1: every slice of the picture has the same parameters (no rate control here), so all of them are launched with the first one and
   * their wavefronts run side by side on the device; JM still codes them one after the other */
  if (p_Inp->slice_mode == FIXED_MB && first == 0 && prm.num_mb < left) prm.num_slices = (left + prm.num_mb - 1) / prm.num_mb;
  P.dealt = 0;
  if (G.nctx > 1 && first == 0 && prm.num_slices > 1 && prm.num_slices % G.nctx == 0 && prm.num_mb % (int)p_Vid->PicWidthInMbs == 0) {
    /* the slices dealt to the devices in order, num_slices / nctx each: every device gets the source picture and launches its share; the records come back from the
     * device that owns the macroblock, the bands are exchanged before DeblockFrame (pipe_deblock), every device keeps the whole reference */
    const int spr = prm.num_slices / G.nctx;
    int c;
    for (r = 0; r < prm.num_ref; r++)
      if (slot_find(currSlice->listX[LIST_0][r]) != prm.ref_slot[r] || G.slot_chroma[prm.ref_slot[r]] != 2) { fprintf(stderr, "jmhip adapter: slices dealt to %d devices: reference %d is not a picture every device holds\n", G.nctx, r); exit(70); }
    P.dealt = 1; P.share_mbs = spr * prm.num_mb; P.band_rows = P.share_mbs / (int)p_Vid->PicWidthInMbs;
    for (c = 0; c < G.nctx; c++) {
      jmhip_slice_params q = prm;
      q.first_mb = c * P.share_mbs; q.num_slices = spr; q.slice_nr = prm.slice_nr + c * spr;
      if (spr == 1) q.num_mb = imin(prm.num_mb, (int)p_Vid->PicSizeInMbs - q.first_mb);      /* the picture's last slice may be shorter */
      if (c && (rc = jmhip_set_current_planes(G.ctxs[c], p_Vid->pCurImg[0], (int)(p_Vid->pCurImg[1] - p_Vid->pCurImg[0]),
                                              p_Vid->pImgOrg[1][0], p_Vid->pImgOrg[2][0], (int)(p_Vid->pImgOrg[1][1] - p_Vid->pImgOrg[1][0])))) adapter_die("jmhip_set_current_planes", rc);
      if ((rc = jmhip_encode_slice_begin(G.ctxs[c], &q))) { fprintf(stderr, "jmhip adapter: device %d: %s\n", c, jmhip_last_error(G.ctxs[c])); adapter_die("jmhip_encode_slice_begin", rc); }
    }
  }
  else if (G.nctx > 1) { fprintf(stderr, "jmhip adapter: JMHIP_DEVICES names %d contexts but this picture's %d slice(s) of %d macroblocks cannot be dealt to them (whole rows, a multiple of the contexts)\n", G.nctx, prm.num_slices > 1 ? prm.num_slices : 1, prm.num_mb); exit(70); }
  else if (F.on) flight_launch(p_Vid, &prm);
  else if ((rc = jmhip_encode_slice_begin(G.ctx, &prm))) adapter_die("jmhip_encode_slice_begin", rc);
  P.slice_last = prm.num_slices > 1 ? (int)p_Vid->PicSizeInMbs - 1 : first + prm.num_mb - 1;
  P.n_slices++; P.n_refs += prm.num_ref;
  P.t_dev += now_s() - t0;
  if (TL_n && first == 0) TL_begun[TL_n - 1] = now_s();
  if (TL_n <= 2 && first == 0) init_prof("the picture's launch is queued");
}

/* one macroblock's record into what write_macroblock (macroblock.c:2810), the MV predictor of later macroblocks and JM's statistics read */
static void mb_from_record(Macroblock *currMB, const jmhip_mb_record *r)
{
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currMB->p_Vid;
  PicMotionParams **motion = p_Vid->enc_picture->mv_info;
  const int mbt = r->mb_type, intra = mbt >= I4MB;
  int k, j, i, uv;
  currMB->mb_type = (short)mbt;
  currMB->best_mode = (short)((mbt == 0 && currSlice->slice_type != B_SLICE) ? 1 : mbt);
  currMB->ar_mode = currMB->best_mode;
  currMB->cbp = r->cbp; currMB->cbp_blk = (int64)r->cbp_blk;
  currMB->luma_transform_size_8x8_flag = (byte)(r->transform8x8 != 0);
  currMB->i16mode = r->i16mode;
  currMB->i16offset = mbt == I16MB ? I16Offset(r->cbp, r->i16mode) : 0;
  currMB->c_ipred_mode = intra ? r->c_ipred_mode : DC_PRED_8;
  currMB->min_rdcost = (distblk)r->min_rdcost;
  for (k = 0; k < 4; k++) {
    currMB->b8x8[k].mode = r->b8mode[k]; currMB->b8x8[k].pdir = (char)(intra ? -1 : 0);
    currMB->b8x8[k].ref[LIST_0] = r->b8ref[k]; currMB->b8x8[k].ref[LIST_1] = -1; currMB->b8x8[k].bipred = 0;
    if (currSlice->slice_type == B_SLICE) { currMB->b8x8[k].pdir = r->b8pdir[k]; currMB->b8x8[k].ref[LIST_1] = r->b8ref1[k]; currMB->b8x8[k].bipred = r->b8bipred[k]; }
  }
  memcpy(currMB->intra_pred_modes, r->ipred_syntax, 16);
  if (mbt == I8MB) memcpy(currMB->intra_pred_modes8x8, r->ipred_syntax, 16);       /* writeIntra8x8Modes (macroblock.c:1817) reads these, at [4 * b8] */
  for (j = 0; j < 4; j++)
    for (i = 0; i < 4; i++) {
      PicMotionParams *mp = &motion[currMB->block_y + j][currMB->block_x + i];
      const int ref = intra ? -1 : r->b8ref[(j >> 1) * 2 + (i >> 1)];
      p_Vid->ipredmode[currMB->block_y + j][currMB->block_x + i] = r->ipredmode[j * 4 + i];
      mp->mv[LIST_0].mv_x = r->mv[j * 4 + i][0]; mp->mv[LIST_0].mv_y = r->mv[j * 4 + i][1];
      mp->ref_idx[LIST_0] = (char)ref; mp->ref_pic[LIST_0] = ref < 0 ? NULL : currSlice->listX[LIST_0][ref];
      mp->mv[LIST_1].mv_x = mp->mv[LIST_1].mv_y = 0; mp->ref_idx[LIST_1] = -1; mp->ref_pic[LIST_1] = NULL;
      if (currSlice->slice_type == B_SLICE) {
        const int ref1 = intra ? -1 : r->b8ref1[(j >> 1) * 2 + (i >> 1)];
        mp->mv[LIST_1].mv_x = r->mv1[j * 4 + i][0]; mp->mv[LIST_1].mv_y = r->mv1[j * 4 + i][1];
        mp->ref_idx[LIST_1] = (char)ref1; mp->ref_pic[LIST_1] = ref1 < 0 ? NULL : currSlice->listX[LIST_1][ref1];
      }
    }
  if (r->transform8x8 && mbt != I4MB && mbt != I16MB) {
    /* an 8x8 transform block: the record holds its 64 levels in zig-zag order; JM's writers read one 64-entry list (CABAC, residual_transform_quant_luma_8x8
     * transform8x8.c:522) or four lists of every fourth position (CAVLC, residual_transform_quant_luma_8x8_cavlc :604) */
    for (k = 0; k < 4; k++) {
      const int16_t *z = &r->luma[4 * k][0];
      if (currSlice->symbol_mode == CABAC) {
        list_from_dense(z, 0, 64, currSlice->cofAC[k][0][0], currSlice->cofAC[k][0][1]);
        for (j = 1; j < 4; j++) { currSlice->cofAC[k][j][0][0] = 0; currSlice->cofAC[k][j][1][0] = 0; }
      } else
        for (j = 0; j < 4; j++) {
          int16_t d16[16];
          for (i = 0; i < 16; i++) d16[i] = z[4 * i + j];
          list_from_dense(d16, 0, 16, currSlice->cofAC[k][j][0], currSlice->cofAC[k][j][1]);
        }
    }
  } else
  for (k = 0; k < 16; k++) list_from_dense(r->luma[k], mbt == I16MB ? 1 : 0, 16, currSlice->cofAC[k >> 2][k & 3][0], currSlice->cofAC[k >> 2][k & 3][1]);
  list_from_dense(r->luma_dc, 0, mbt == I16MB ? 16 : 0, currSlice->cofDC[0][0], currSlice->cofDC[0][1]);
  for (uv = 0; uv < 2; uv++) {
    if (p_Vid->yuv_format == YUV422) {                       /* eight DC levels in SCAN_YUV422 order; the plane's eight blocks live in cofAC[4 + 2 uv] and [5 + 2 uv] (block.c:1096-1105) */
      list_from_dense(r->chroma_dc[uv], 0, 8, currSlice->cofDC[uv + 1][0], currSlice->cofDC[uv + 1][1]);
      for (k = 0; k < 8; k++) list_from_dense(r->chroma_ac[uv][k], 1, 16, currSlice->cofAC[4 + 2 * uv + (k >> 2)][k & 3][0], currSlice->cofAC[4 + 2 * uv + (k >> 2)][k & 3][1]);
      continue;
    }
    list_from_dense(r->chroma_dc[uv], 0, 4, currSlice->cofDC[uv + 1][0], currSlice->cofDC[uv + 1][1]);
    for (k = 0; k < 4; k++) list_from_dense(r->chroma_ac[uv][k], 1, 16, currSlice->cofAC[4 + uv][k][0], currSlice->cofAC[4 + uv][k][1]);
  }
}

extern void __real_encode_one_macroblock_low(Macroblock *);
void __wrap_encode_one_macroblock_low(Macroblock *currMB)
{
  Slice *currSlice = currMB->p_Slice;
  VideoParameters *p_Vid = currMB->p_Vid;
  double t0;
  if (!adapter_on(p_Vid) || !pipe_config_ok(p_Vid) || (currSlice->slice_type != P_SLICE && currSlice->slice_type != I_SLICE && currSlice->slice_type != B_SLICE) ||
      (currSlice->slice_type == B_SLICE && (p_Vid->nal_reference_idc != 0 || currSlice->listXsize[LIST_0] + currSlice->listXsize[LIST_1] > JMHIP_MB_MAX_REF || p_Vid->active_pps->weighted_bipred_idc != 0)) ||
      currSlice->mb_aff_frame_flag || p_Vid->structure != FRAME) {
    if (P.n_slices) { fprintf(stderr, "jmhip adapter: macroblock pipeline: a slice outside its scope after %ld slices on the device\n", P.n_slices); exit(70); }
    G.n_passed++;
    __real_encode_one_macroblock_low(currMB);
    return;
  }
  if (currMB->mbAddrX == currSlice->start_mb_nr) pipe_run_slice(currMB);
  {
    const jmhip_mb_record *rec;
    int rc;
    t0 = now_s();
    if (F.on) { if ((rc = jmhip_seq_record(G.ctx, F.cur_entry, currMB->mbAddrX, &rec))) adapter_die("jmhip_seq_record", rc); }
    else if (P.dealt) { if ((rc = jmhip_slice_record(G.ctxs[currMB->mbAddrX / P.share_mbs], currMB->mbAddrX, &rec))) adapter_die("jmhip_slice_record (dealt)", rc); }
    else if ((rc = jmhip_slice_record(G.ctx, currMB->mbAddrX, &rec))) adapter_die("jmhip_slice_record", rc);     /* waits while the device is behind */
    P.t_wait += now_s() - t0;
    if (TL_n) { TL_wait[TL_n - 1] += now_s() - t0; if (currMB->mbAddrX == 0) TL_first[TL_n - 1] = now_s(); if (currMB->mbAddrX == (int)p_Vid->PicSizeInMbs - 1) TL_last[TL_n - 1] = now_s(); }
    t0 = now_s();
    mb_from_record(currMB, rec);
    P.t_fill += now_s() - t0;
    if (P.dealt) {
      const int c = currMB->mbAddrX / P.share_mbs;
      if ((currMB->mbAddrX == P.slice_last || (currMB->mbAddrX + 1) % P.share_mbs == 0) && (rc = jmhip_encode_slice_end(G.ctxs[c]))) adapter_die("jmhip_encode_slice_end (dealt)", rc);
    } else
    if (!F.on && currMB->mbAddrX == P.slice_last && (rc = jmhip_encode_slice_end(G.ctx))) adapter_die("jmhip_encode_slice_end", rc);
    if (TL_n && currMB->mbAddrX == P.slice_last) TL_ended[TL_n - 1] = now_s();
  }
  P.mbs++; P.n_mbs++;
}

/* DeblockFrame / getSubImagesLuma / getSubImagesChroma of a picture the pipeline encoded: 1 = served here */
static int pipe_deblock(VideoParameters *p_Vid, imgpel **imgY, imgpel ***imgUV)
{
  int rc;
  double t0 = now_s();
  if (!P.ok || P.pic != p_Vid->enc_picture || imgY != p_Vid->enc_picture->imgY || P.mbs != (int)p_Vid->PicSizeInMbs) return 0;
  if (F.on) {                                               /* filtered (and interpolated) inside the picture's launch: wait for it, read its errors, fetch the picture */
    if ((rc = jmhip_seq_wait(G.ctx, F.cur_entry))) adapter_die("jmhip_seq_wait", rc);
    if ((rc = jmhip_seq_get_recon(G.ctx, F.cur_slot, imgY[0], (int)(imgY[1] - imgY[0]), imgUV[0][0], imgUV[1][0], (int)(imgUV[0][1] - imgUV[0][0])))) adapter_die("jmhip_seq_get_recon", rc);
    P.deblocked = 1; G.n_deblock++; P.t_dev += now_s() - t0;
    return 1;
  }
  if (P.dealt) {                                            /* every device gets every band, then filters the whole picture (it keeps the whole reference) */
    int c;
    if ((rc = jmhip_allgather_bands(G.ctxs, G.nctx, P.band_rows))) adapter_die("jmhip_allgather_bands", rc);
    for (c = 1; c < G.nctx; c++) if ((rc = jmhip_deblock_picture_dev(G.ctxs[c], p_Vid->active_sps->direct_8x8_inference_flag))) adapter_die("jmhip_deblock_picture_dev (dealt)", rc);
  }
  if ((rc = jmhip_deblock_picture_dev(G.ctx, p_Vid->active_sps->direct_8x8_inference_flag))) adapter_die("jmhip_deblock_picture_dev", rc);
  if ((rc = jmhip_get_recon(G.ctx, imgY[0], (int)(imgY[1] - imgY[0]), imgUV[0][0], imgUV[1][0], (int)(imgUV[0][1] - imgUV[0][0])))) adapter_die("jmhip_get_recon", rc);
  P.deblocked = 1;
  G.n_deblock++;
  P.t_dev += now_s() - t0;
  return 1;
}
static int pipe_reference(StorablePicture *s)
{
  int k, rc;
  double t0 = now_s();
  if (!P.ok || P.pic != s || !P.deblocked) return 0;
  if (!s->used_for_reference) return 1;                      /* a B picture: nobody predicts from it, it takes no slot and gets no planes */
  if (F.on) { G.n_interp++; return 1; }                     /* the planes are in the picture's slot already (flight_launch registered it) */
  k = slot_take(s);
  if (G.nctx > 1) { int c; for (c = 1; c < G.nctx; c++) if ((rc = jmhip_reference_from_recon(G.ctxs[c], k))) adapter_die("jmhip_reference_from_recon (dealt)", rc); }
  if ((rc = jmhip_reference_from_recon(G.ctx, k))) adapter_die("jmhip_reference_from_recon", rc);
  G.slot_chroma[k] = 2;                                     /* on the device, and nobody on the host needs the sub-images */
  G.n_interp++;
  P.t_dev += now_s() - t0;
  return 1;
}
static void pipe_report(void)
{
  if (G.nctx > 1) fprintf(stderr, "jmhip adapter: %d contexts (JMHIP_DEVICES): the slices of a picture dealt to them, the bands exchanged with jmhip_allgather_bands\n", G.nctx);
  if (F.on) fprintf(stderr, "jmhip adapter: pictures in flight: %ld pictures, %ld launched ahead of time (up to %d in flight), %ld of them served as launched, %ld voided\n", F.pic, F.n_ahead, F.depth, F.n_hit, F.n_void);
  if (P.n_slices)
    fprintf(stderr, "jmhip adapter: macroblock pipeline: %ld slices, %ld macroblocks encoded on the MI355X (encode_one_macroblock_low never ran on the host); "
                    "device calls %.3f s, waiting for records %.3f s, unpacking them %.3f s; wall time inside encode_one_slice %.3f s, pad_borders %.3f s, DeblockFrame %.3f s, getSubImagesLuma %.3f s\n",
            P.n_slices, P.n_mbs, P.t_dev, P.t_wait, P.t_fill, T_slice, T_pad, T_deblock, T_interp);
  if (getenv("JMHIP_ADAPTER_TIMELINE")) {
    int i;
    for (i = 0; i < TL_n; i++)
      fprintf(stderr, "jmhip adapter: picture %d: slices %.1f ms (after its start: launched %.1f, first record %.1f, last record %.1f, launch closed %.1f ms; waiting for records %.1f ms), DeblockFrame %.1f ms, getSubImagesLuma %.1f ms, everything else up to the next picture's first slice %.1f ms (slice end -> DeblockFrame %.1f, DeblockFrame -> getSubImagesLuma %.1f, getSubImagesLuma -> next slice %.1f)\n", i,
              1e3 * (TL_out[i] - TL_in[i]), 1e3 * (TL_begun[i] - TL_in[i]), 1e3 * (TL_first[i] - TL_in[i]), 1e3 * (TL_last[i] - TL_in[i]), 1e3 * (TL_ended[i] - TL_in[i]), 1e3 * TL_wait[i], 1e3 * TL_db[i], 1e3 * TL_ip[i], i + 1 < TL_n ? 1e3 * (TL_in[i + 1] - TL_out[i] - TL_db[i] - TL_ip[i]) : 0.0,
              1e3 * (TL_db_at[i] - TL_out[i]), 1e3 * (TL_ip_at[i] - TL_db_at[i] - TL_db[i]), i + 1 < TL_n ? 1e3 * (TL_in[i + 1] - TL_ip_at[i] - TL_ip[i]) : 0.0);
  }
}

/* ------------------------------------------------------------------ FmoGetLastCodedMBOfSliceGroup (lencod/src/fmo.c:676)
 * end_macroblock (macroblock.c:512) asks for it once per macroblock and JM answers with a loop over the whole macroblock-to-slice-group map:
 * PicSizeInMbs^2 steps per picture (66 million at 1080p, a fifth of the host time left once the macroblock decisions are off the CPU).
 * Without FMO there is one slice group and the answer is the picture's last macroblock; with FMO JM's own loop runs. */
extern int __real_FmoGetLastCodedMBOfSliceGroup(VideoParameters *, int);
int __wrap_FmoGetLastCodedMBOfSliceGroup(VideoParameters *p_Vid, int SliceGroupID)
{
  if (p_Vid->p_Inp->num_slice_groups_minus1 == 0 && SliceGroupID == 0 && p_Vid->PicSizeInMbs > 0 && !(G.init_done && G.off))
    return (int)p_Vid->PicSizeInMbs - 1;
  return __real_FmoGetLastCodedMBOfSliceGroup(p_Vid, SliceGroupID);
}
