// mbpipe.hip -- the RDO-off macroblock pipeline of a slice on the MI355X (gfx950): SURVEY.md 8f row 1.
//
// What it computes, macroblock for macroblock, is what the reference's encode_one_macroblock_low (lencod/src/md_low.c:104-687) leaves behind
// for write_macroblock and DeblockFrame; the entry point and the functions it stands for are listed in include/jmhip.h (jmhip_encode_slice).
//
// Design (not the reference's: JM walks macroblocks in raster order on one thread).
//   * One persistent launch per slice.  A workgroup draws a ticket, the ticket names a macroblock in wavefront order x + 2y (every macroblock
//     a macroblock depends on -- left, up-left, up, up-right -- has a smaller ticket), the workgroup waits for those neighbours' done flags
//     and then owns the macroblock from motion search to reconstruction.  What neighbours read from each other (the bottom row / right column
//     of the reconstruction, of mv_info and of ipredmode) travels in a 136-byte edge record written with write-through (sc1) stores and read
//     with sc1 loads; the flag follows the drained stores (per-XCD L2s are not coherent, MI355X guide: inter-workgroup visibility).
//   * Inside a macroblock the searches form independent chains: the sub-modes of an 8x8 block only read vectors of their own sub-blocks and of
//     finished 8x8 blocks, 16x8 / 8x16 only their own first partition.  Eight waves: waves 0-3 take the sub-modes 4x4, 4x8, 8x4, 8x8 of the
//     P8x8 chain (meeting after every 8x8 block), waves 4-6 the 16x16 (with the skip vector's cost), 16x8 and 8x16 searches, wave 7 the intra
//     side (Intra4x4 chain, Intra16x16 search, chroma intra decision).  Then one wave decides and codes the winner.
//   * A search = one wave.  The integer windows of all references sit in LDS once per macroblock: with RDOptimization = 0 JM clamps the
//     search centre to +-SearchRange (mv_search.c:945-954), so every candidate of every partition lies within +-2 SearchRange of the macroblock.
//     lane = window column, the lane slides down its column keeping one running SAD per candidate in flight (each window row is loaded and
//     byte-aligned once and meets every row of the block), the block's samples are scalar operands.  Integer SAD (v_sad_u8), 64-bit keys
//     (cost, spiral index): JM's "first in spiral order wins" is the unsigned minimum.
//   * Sub-pel refinement: (candidate, 4x4 sub-block) items over the lanes, Hadamard SATD from the 16 quarter-pel planes in HBM/L2 with JM's
//     per-sub-block origin clamp (UMVLine4X), JM's strict-'<' scan replayed on the nine sums.
// No MFMA: the path is byte / integer add, sub, shift, abs, min.
#include <mutex>
#include <thread>
#include <unistd.h>
#include <time.h>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include "jmhip_internal.h"
#include "me_common.h"
#include "deblock_common.h"

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

// The eight kernel instances are compiled as separate translation units (jm_amd/build.py: -DMBPIPE_PART=0..3, 5..8, a kernel each; part 4 = the host side, which
// only declares them): one hipcc process per kernel instead of one for all -- minutes of build time, nothing else.  Without the macro: one unit.
#ifndef MBPIPE_PART
#define MBPIPE_PART -1
#endif
#define MBPIPE_HOST (MBPIPE_PART == -1 || MBPIPE_PART == 4)
// JMHIP_MB_PROF's time stamps: a kernel instance compiled with -DMBPIPE_PROF_ON=0 has none of their tests (the headline instance k_mb_pipe: ~55 instructions of every search; its
// twin k_mb_pipe_prof, part 12, keeps them and is what a context made under JMHIP_MB_PROF launches)
#ifndef MBPIPE_PROF_ON
#define MBPIPE_PROF_ON 1
#endif
#define APROF (MBPIPE_PROF_ON ? A.prof : (unsigned long long *)nullptr)

#define MB_THREADS 512
#define MAXC 0x7fffffff
#define EDGE_WORDS 27                       // u64 per macroblock edge record: 0-1 bottom luma row, 2-3 right luma column, 4/5 bottom U/V rows,
                                            // 6/7 right U/V columns, 8 ipredmode (bytes 0-3 bottom row, 4-7 right column), 9-12 / 13-16 mv_info of the
                                            // bottom row / right column {packed vector, reference index}; 4:2:2: 17/18 rows 8..15 of the right U/V columns;
                                            // B slices: 19-22 / 23-26 the same mv_info of LIST_1
#define EDGE_MV_END 17                      // words 9 .. 16 are the tagged vectors
#define EDGE_MV1 19                         // ... and 19 .. 26 those of list 1
#define SPIN_LIMIT (1u << 26)
#define JMHIP_SEQ_MAX_FLIGHT 8              // references of a picture that may still be in the making when its launch starts (the most recent ones)

struct __attribute__((packed)) U32un { u32 v; };
__device__ __forceinline__ u32 ldu32(const u8 *p) { return ((const U32un *)p)->v; }
// Samples of a reference picture (the sixteen quarter-pel planes, the chroma planes).  With pictures in flight side by side (jmhip_seq_*) another workgroup of another
// launch may have written them microseconds ago, write-through: they are read with sc1 loads (served by the L2 but never by this compute unit's L1, and coherent with
// the other XCDs' write-through stores -- MI355X guide, inter-workgroup visibility), the same form the edge records use.  Any byte alignment, like ldu32.
#ifdef JMHIP_REF_PLAIN                                           // A/B aid: plain loads (valid only when every reference is complete before the launch)
__device__ __forceinline__ u32 ldref32(const u8 *p) { return ((const U32un *)p)->v; }
__device__ __forceinline__ u32 ldref8(const u8 *p) { return *p; }
#else
__device__ __forceinline__ u32 ldref32(const u8 *p) { return __hip_atomic_load((const u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ldref8(const u8 *p) { return (u32)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
__device__ __forceinline__ int clampi3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int mvx(int p) { return (int)(int16_t)(p & 0xffff); }
__device__ __forceinline__ int mvy(int p) { return p >> 16; }
__device__ __forceinline__ int mvpack(int x, int y) { return (x & 0xffff) | (y << 16); }
__device__ __forceinline__ int median3(int a, int b, int c) { return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b)); }

static_assert(sizeof(jmhip_mb_record) == 1296 && sizeof(jmhip_slice_params) == 3200, "record sizes of include/jmhip.h");

// What one PICTURE of a launch is made of.  A launch of one picture (jmhip_encode_slice*, jmhip_seq_encode) carries it in its arguments (PipeArgs::v); a launch of several
// consecutive pictures (jmhip_seq_batch) reads picture k's from PipeArgs::pics[k] when a workgroup draws a ticket of that picture.  Either way the workgroup works on its
// copy in LDS (Shared::V, PV below).
struct PicView {
  const u8 *cur_y, *cur_u, *cur_v;           // source planes
  const u8 *ref_y[JMHIP_MB_MAX_REF];         // 16 quarter-pel planes of each reference (plane (0,0) first)
  const u8 *ref_u[JMHIP_MB_MAX_REF];         // the references' chroma: U, then V cw x ch bytes on (jmhip_ctx::d_refc)
  u8 *rec_y, *rec_u, *rec_v;                 // reconstruction; with `fused` the planes of the slot the picture goes to (filtered in place)
  u8 *out_planes;                            // fused: the sixteen quarter-pel planes of that slot
  u64 *edge;
  unsigned *done;
  jmhip_mb_record *records;
  jmhip_mb_record *hrecords;                 // streaming to the host (jmhip_encode_slice_begin): pinned, device-visible copies of the records
  unsigned *hflags;                          // ... and per macroblock the epoch of the launch whose record is complete there; null otherwise
  jmhip_db_mb *dbmb;
  jmhip_db_motion *dbmo;
  u32 *post;                                 // fused: per macroblock of the slot: post_tag once the macroblock is filtered and its share of the planes is written
  const u32 *ref_post[JMHIP_SEQ_MAX_FLIGHT]; // the post flags of reference r's slot, or null (complete before the launch); references beyond are always complete
  const u32 *war_post;                       // a launch of several pictures: the post flags of the picture that must be done before this one may write its slot, or null
  const u32 *prev_post;                      // ... and of the picture before this one in the launch: pictures complete in launch order (this one's last macroblock waits for that one's), so
                                             // "picture j is done" (war_post) says the same of every picture before j
  int *ez_state;                             // EPZS: per macroblock its four columns of p_EPZS->distortion [7][4] and p_EPZS->p_motion [num_ref][7][4][4] when it was done
  u64 *mot_out;                              // EPZS: per 4x4 block of the picture {packed vector, poc of the picture referred to}: later pictures' temporal predictors
  const u64 *mot_ref[2];                     // EPZS in flight: the motion kept with the slots of references 0 / 1, read where k_epzs_coloc's whole-picture pass would have put the
                                             // co-located vectors (the reference may still be in the making); null: no temporal predictors
  u8 *colz_out;                              // P / I pictures: per 4x4 block bit 0 where a B picture's spatial direct mode finds "the co-located block does not move" in this picture, bit 1: the block is inter
  u64 *colm_out;                             // ... and its vector | the picture id of its reference << 32 (a B picture's temporal direct mode: Get_Direct_MV_Temporal mv_direct.c:40)
                                             // (get_colocated_info mv_direct.c:428: ref_idx[LIST_0] == 0 and a vector within +-1), else 0
  const u8 *colz;                            // B pictures: that map of listX[LIST_1][0]
  const u64 *colm;                           // ... and its motion
  u32 ref_tag[JMHIP_SEQ_MAX_FLIGHT];
  u32 post_tag, war_tag, prev_tag;
  int key0;                                  // a launch of several pictures: the queue key of this picture's macroblock (0, 0) -- macroblock (x, y) has key0 + x + 2 y, tickets go by key
  int poc_off;                               // ... EPZS: this picture's order counts are the launch's poc_cur / poc_ref[] + poc_off (jmhip_seq_picture::poc_offset)
  unsigned short ref_dkey[JMHIP_SEQ_MAX_FLIGHT];   // ... and how far behind the picture that reference r is (key0 - ITS key0; at most 8 x the keys of a picture), where ref_post[r] is set
                                             // (EPZS: is the macroblock a search waits for handed out?)
  int pad2_;
  unsigned epoch;
  int ref_id[JMHIP_MB_MAX_REF];              // identity of reference r for the loop filter's comparison (jmhip_slice_params::ref_id)
};
static_assert(sizeof(PicView) % 8 == 0, "copied as dwords, pointers first");

struct PipeArgs {
  jmhip_slice_params p;
  int W, H, wmb, hmb, cw, ch;
  unsigned role_perm;                        // role of hardware wave w = nibble w: which chain a wave runs (0-3 the P8x8 sub-modes, 4 16x16 + Intra16x16 + chroma decision, 5 16x8,
                                             // 6 8x16, 7 the Intra4x4 chain).  Waves w and w + 4 share a SIMD: the pairing decides who competes with the 4x4 chain for issue slots
  int help;                                  // P pictures of the full searches with several references: waves that are done with their own role take references of the sub-mode waves' passes
  int c422;                                  // 4:2:2: chroma planes cw x H, 8 x 16 samples per macroblock (ch = H); else 4:2:0
  int total_mb;                              // tickets of the launch: macroblocks of num_slices slices of p.num_mb, cut at the end of the picture -- or of every picture of a batch
  int nbands, band_start[9];                 // 8 (or 1): the ticket order is cut into bands of macroblock rows, tickets band_start[b] .. band_start[b + 1] - 1 belong to band b, and
                                             // workgroup g draws from band g % 8 first -- block g is observed to run on XCD g % 8, so an XCD's L2 then only ever sees its band's
                                             // rows of the references' planes (speed only: any workgroup may take any ticket)
  int cur_pitch, ref_pitch, rec_pitch;       // luma pitches (bytes); chroma planes are cw wide
  long plane_stride;
  int win_h, win_p, win_ox;                  // LDS window of a reference: (16 + 4R) rows of win_p bytes; the macroblock's column 0 sits at byte win_ox
                                             // (2R rounded up to a multiple of 4, so picture dwords stay aligned)
  unsigned *sync;
  const int *order;                          // ticket -> macroblock address (a batch: picture << 16 | address)
  int prof_mode;                             // JMHIP_MB_PROF value: 1 = a 4x4 search's parts in stamps 18..22, 2 = an Intra4x4 block's parts there
  // EPZS (search_mode 3)
  const int *col;                            // [H / 4][W / 4] co-located vectors scaled to this picture (k_epzs_coloc), or null (EPZSTemporal = 0)
  int ez_words;                              // ints per macroblock of PicView::ez_state: 28 + 112 num_ref
  unsigned long long *prof;                  // profiling aid (JMHIP_MB_PROF=1): 24 time stamps (100 MHz) per macroblock, or null
  // Pictures in flight side by side (jmhip_seq_encode, mbpipe_post.inc): the macroblock's share of the loop filter and of the quarter-pel planes follows its coding
  // inside the launch, and the references may still be in the making
  int fused;
  int direct8x8;                             // active_sps->direct_8x8_inference_flag (DeblockMb's skip rule for B_Skip: never met in P / I pictures)
  int reach_x, reach_y;                      // macroblock (X, r) reads reference samples that macroblocks up to (X + reach_x, r + reach_y) of the reference produce
  unsigned char win_of[JMHIP_MB_MAX_REF];    // B slices: which LDS window holds reference r of the combined list (list 0, then list 1: the same picture may be in both)
  int nwin;                                  // ... and how many windows there are; window w is loaded from reference win_src[w]
  unsigned char win_src[JMHIP_MB_MAX_REF];
  int npics;                                 // > 0: a launch of several consecutive pictures (jmhip_seq_batch): pics[0 .. npics), tickets ordered by wavefront index + lag x picture
  const PicView *pics;
                                             // ... behind the tickets (256-byte aligned): key_end[k] = tickets with keys <= k (EPZS: ez_ensure_ref; the arguments' 4 KB are full)
  PicView v;                                 // the picture of a one-picture launch
};
static_assert(sizeof(PipeArgs) <= 4096, "kernel arguments: 4 KB");

// scratch of the post stage (mbpipe_post.inc)
struct PostShared {
  jmhip_db_motion own[16], mo[2][4];         // the macroblock's 4x4 blocks; the left neighbour's right column, the upper neighbour's bottom row
  jmhip_db_mb q, nb[2];                      // the macroblock, its left and its upper neighbour
  u8 str[2][4][4];                           // [dir][edge][segment] boundary strengths
  __attribute__((aligned(8))) u8 y[20 * LP]; // luma rows -4 .. 15, columns -4 .. 15 at [(r + 4) * LP + c + 4]
  __attribute__((aligned(8))) u8 c[2][18 * CP];   // chroma rows -2 .. 15, columns -4 .. 7 at [(r + 2) * CP + c + 4]
  __attribute__((aligned(8))) u8 src[21][24];     // interpolation: source rows y0 - 2 .. y0 + 18, columns x0 - 4 .. x0 + 19
  int16_t h[21][16];                         // unclipped horizontal six-tap sums
  u8 v[16][20];                              // vertical half-pel samples (17 columns used)
  __attribute__((aligned(16))) u8 outp[16][16][16];   // the block's sixteen planes [plane][row][column]: gathered here so that they leave as 16-byte write-through stores
};

// per-workgroup state in LDS
struct Shared {
  alignas(16) u32 cur_y[64];                 // the source macroblock, 16 rows of 4 dwords
  u32 cur_c[2][32];                          // U, V: 8 (4:2:0) or 16 (4:2:2) rows of 2 dwords
  u64 nb[4][EDGE_WORDS];                     // edge records of A (left), B (up), C (up-right), D (up-left)
  u32 ptab[8][16];                           // mv_predictor's neighbour table: per (block type, 4x4 position of the block) four codes A, B, C, D (once per launch)
  int avail[4];
  int ticket, addr, err;
  int vpic;                                  // a launch of several pictures: the picture V holds (-1: none yet)
  int allmv[JMHIP_MB_MAX_REF][8][16];        // currSlice->all_mv[LIST_0][ref][mode][4x4 raster], packed
  int mcost[4][JMHIP_MB_MAX_REF][4];         // p_Vid->motion_cost[mode][LIST_0][ref][block] of the modes 1..3 (a sub-mode sums its costs up as the references go by)
  int mvi[8][16][2];                         // per wave: the macroblock's mv_info as that wave's chain sees it {packed mv, ref_idx}
  int p8_cost[4][4], p8_bref[4][4];          // [block][mode - 4]
  int p8_cnt;
  int p8_mode[4], p8_ref[4], p8_total;
  int m_cost[4], m_bref[4][4];               // modes 1..3: cost, best reference per 8x8 block
  int skip_mv;
  // intra side
  int i4_cost, i4_cbp;
  int8_t i4_ipm[16], i4_syn[16];
  int16_t i4_lev[16][16];
  u8 i4_rec[256];
  u8 i4p[2][16], i4val[2][44], i4tab[144];   // i4p / i4val: per wave of the Intra4x4 chain (two in I slices)
  int i4prog[2], i4_t1, i4_c1;               // I slices: blocks each of the two Intra4x4 waves has finished (running counts over the launch), the second wave's cost and cbp
  jmhip_qparam q_luma[2][16], q_chroma[2][2][16];   // the slice's quantiser tables (out of the kernel arguments once per workgroup: LDS reads can be batched)
  int i16_cost, i16_mode;
  u8 e16[36];
  u8 ec[2][28];                              // chroma predictor samples: [0] corner, [1..8] up, [9..16] left (4:2:2: [9..24])
  int c_ipred;
  u8 icpred[2][4][128];                      // [plane][mode][row * 8 + column]
  // final coding
  u8 pred[256], rec[256];
  u8 predc[2][128], recc[2][128];
  int16_t dcbuf[16];
  int red[8][80];
  int ref_ok[3][JMHIP_SEQ_MAX_FLIGHT];       // EPZS in flight, per search chain and reference: the macroblock (column | row << 16) whose post flag the chain has seen (-1: none) --
                                             // everything left of and above it is filtered and interpolated (mbpipe_post.inc)
  int nbflag;                                // the neighbours' samples are in nb (running count over the launch)
  int stagecnt;                              // waves that have staged their share of the current macroblock (running count over the launch: mbpipe_kernel.inc)
  int pflag[4];                              // waves 0-3: how many 8x8 blocks of the P8x8 chain each has finished (running count over the launch)
  // Several references, full searches: the passes of a sub-mode over the references are independent of each other until the costs are compared, so a wave that has finished its
  // own role (4-7) takes references of sub-mode wave (role - 4)'s current 8x8 block (search_phase)
  int prof_addr;                             // JMHIP_MB_PROF (B slices): the macroblock the time stamps belong to
  int hword[4];                              // per sub-mode wave: (phase number << 8) | next reference to take -- taken with a compare-and-swap by the wave itself and by its helper
  int hcomp[4];                              // ... references of the phase whose pass is complete
  int sbt[4][JMHIP_MB_MAX_REF];              // ... the passes' costs, by reference (the sub-mode wave compares them in JM's order once all are in)
  u32 ytab4[4][72];                          // the same for the four usual offsets of the predictor from the search centre (-2 .. 1 quarter-pels), full range: once per launch
  u32 ytab[8][72];                           // per wave: what a search's candidate row contributes to every key: (lambda * bits(vy - py)) << 8 | zero row << 7 | far rank
  int fin_mv[16], fin_ref[16], fin_type, fin_cbp;
  // High profile (Transform8x8Mode 1)
  jmhip_qparam q8[2][64];                    // the slice's 8x8 quantiser tables
  int t8buf[2][64];                          // scratch of tq8_wave: [0] the inter side (the tr8x8 pass, the final stage), [1] the Intra8x8 chain -- they run side by side
  int i8_cost, i8_cbp;                       // Intra8x8
  int8_t i8_ipm[16], i8_syn[4];
  int16_t i8_lev[16][16];
  u8 i8_rec[256], i8p[28], i8f[28];
  int m_t8[4];                               // transform_decision of modes 1..3
  int p8t_ref[4], p8t_total, p8t_cnt, p8t_cbp, p8t_nonz;   // the tr8x8 pass of P8x8: references, cost, blocks decided, cbp8x8, cnt_nonz_8x8
  u64 p8t_cbp_blk;
  int16_t t8_lev[16][16];
  u8 t8_rec[256], t8_pred[256];
  int fin_t8;                                // the macroblock's final luma_transform_size_8x8_flag (loop-filter side information)
  int fl_cbp, fc_cr;                         // what the luma wave and the chroma wave of the final stage found
  u64 fl_cbp_blk, fc_bits;
  u64 pre_cbp_blk, pre_bits;                 // the 16x16 mode coded AHEAD by the 16x8 / 8x16 waves once they are done (mbpipe_kernel.inc: PRE): what the final stage's two waves would find
  int pre_cbp, pre_cr;
  int m1flag, pre_pad_;                      // ... and: the 16x16 search is done, its vectors stand (running count over the launch)
  u32 fin_cbp_blk;
  jmhip_mb_record out;
  PostShared post;
  PicView V;                                 // the picture the workgroup's current macroblock belongs to
  PicView Vp;                                // ... and the one its PREVIOUS macroblock belonged to: that macroblock's post stage runs beside this one's staging (mbpipe_kernel.inc)
  PicView Vn;                                // ... and the one its NEXT macroblock belongs to, when that ticket was drawn ahead (mbpipe_kernel.inc: draw_ahead)
  int nticket, npacked;                      // the ticket drawn ahead and its entry of the order table
  int pad_n_[2];
  // ---- from here on: B slices only.  The other kernels' dynamic region (the references' windows / the EPZS tables) starts HERE (SHARED_COMMON): with two four-wave EPZS
  // workgroups per compute unit every KB counts twice
  // B slices (mbpipe_b.inc).  References are numbered through both lists: list 0's first, then list 1's (S.allmv, the windows, PicView::ref_y).
  int mvi1[8][16][2];                        // per wave: mv_info of LIST_1 as that wave's chain sees it
  int bpmv[2][2][4][16];                     // currSlice->bipred_mv[set][list][0][mode 1..3][4x4 raster]
  int dmv[2][16], d_ref8[4][2], d_pdir8[4];  // the direct mode: vectors per list and 4x4 block; direct_ref_idx and direct_pdir per 8x8 block (spatial: alike for the whole macroblock; temporal: -1 = the block has none)
  int d_cost4[4], d_cost8[4];                // GetDirectCost8x8 of the four 8x8 blocks (4x4 Hadamards; 8x8 Hadamard)
  int dflag;                                 // the direct vectors and costs are there (running count over the launch)
  int m_info[4][4];                          // modes 1..3: b8x8info->best[mode][block] packed (binfo_pack)
  int p8b_cost[4][5], p8b_info[4][5];        // [block][0 direct, 1..4 modes 4..7]
  int p8_info[4], p8t_info[4];               // the parts the tr4x4 / tr8x8 pass of P8x8 decided
  int fin_mv1[16], fin_ref1[16];             // the macroblock's final LIST_1 vectors and reference indices
  alignas(16) u8 bpred[8][256];              // per wave: a prediction being priced
  __attribute__((aligned(4))) u8 bireg[3][48 * 52];   // waves 4..6: the samples a bi-predictive search's candidates cover (range <= 16: 48 rows of 52 bytes)
};
#define SHARED_COMMON ((offsetof(Shared, mvi1) + 15) & ~(size_t)15)      // what the P / I kernels keep of Shared
#define SHARED_ALL ((sizeof(Shared) + 15) & ~(size_t)15)
extern __shared__ __attribute__((aligned(16))) u8 mb_smem[];
#define PV (((Shared *)mb_smem)->V)
#define PVP (((Shared *)mb_smem)->Vp)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }
// a value every lane of the wave holds alike, moved to a scalar register (loop bounds, addresses and branches on it become scalar)
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// hand-over between two waves of the workgroup through a sequence number in LDS (a wave's LDS operations execute in order)
__device__ __forceinline__ void lds_signal(int *f, int seq, int lane) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); if (lane == 0) *(volatile int *)f = seq; }
// the same for four flags at once (one LDS read per poll: lanes 0..3 read a flag each)
__device__ __forceinline__ void lds_wait_ge4(const int *f, int seq, int lane)
{
  for (;;) {
    const int v = lane < 4 ? *(const volatile int *)(f + lane) : seq;
    if (__ballot(v < seq) == 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void lds_wait_ge(const int *f, int seq) { while (*(const volatile int *)f < seq) __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// Cross-lane steps inside a row of 16 lanes as DPP operands of the ALU (no trip through the LDS crossbar that __shfl_xor takes):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- after n of them every lane holds the result over its 2^n neighbours.
// (every lane of these four patterns has a source lane: `old` is never used, and with old = 0 / bound_ctrl the compiler folds the move into the min / add that follows --
// one instruction per step instead of mov, nop, mov_dpp, op)
template <int CTRL> __device__ __forceinline__ int dpp_(int v)
{
  static_assert(CTRL == 0xB1 || CTRL == 0x4E || CTRL == 0x141 || CTRL == 0x140, "quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror");
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ u32 umin_(u32 a, u32 b) { return a < b ? a : b; }
// minimum over each row of 16 lanes (every lane of the row gets it)
__device__ __forceinline__ u32 row16_min_u32(u32 x)
{
  x = umin_(x, (u32)dpp_<0xB1>((int)x)); x = umin_(x, (u32)dpp_<0x4E>((int)x));
  x = umin_(x, (u32)dpp_<0x141>((int)x)); x = umin_(x, (u32)dpp_<0x140>((int)x));
  return x;
}
// minimum over the wave, in a scalar register
__device__ __forceinline__ u32 wave_min_u32(u32 x)
{
  x = row16_min_u32(x);
  const u32 a = (u32)__builtin_amdgcn_readlane((int)x, 0), b = (u32)__builtin_amdgcn_readlane((int)x, 16);
  const u32 c = (u32)__builtin_amdgcn_readlane((int)x, 32), d = (u32)__builtin_amdgcn_readlane((int)x, 48);
  return umin_(umin_(a, b), umin_(c, d));
}
// sum over aligned groups of n = 1, 2, 4, 8 or 16 lanes (every lane of the group gets it)
__device__ __forceinline__ int group_sum(int v, int n)
{
  if (n > 1) v += dpp_<0xB1>(v);
  if (n > 2) v += dpp_<0x4E>(v);
  if (n > 4) v += dpp_<0x141>(v);
  if (n > 8) v += dpp_<0x140>(v);
  return v;
}

__device__ __forceinline__ u64 ld_sc1(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ------------------------------------------------------------------ neighbours of the current macroblock
// mv_info entry at 4x4 position (x4, y4) relative to the macroblock, x4, y4 in [-1, 4]: where its {packed mv, ref} lies in LDS, and whether it exists.
// Inside the macroblock: this chain's view (two ints); outside: the neighbour's edge record (words 9..12 bottom row, 13..16 right column: the vector
// in the low half, the reference index in byte 4).  Both read as one 64-bit word: mv = low half, ref = sign-extended byte 4.
// avm: bit n = neighbour n (A left, B up, C up-right, D up-left) is inside the picture and the slice.
__device__ __forceinline__ const u64 *mvinfo_ptr(const Shared &S, int view, int avm, int x4, int y4, bool &ok, int list = 0)
{
  const int e0 = list ? EDGE_MV1 : 9;
  if (x4 < 0) {
    if (y4 < 0) { ok = (avm >> 3) & 1; return &S.nb[3][e0 + 3]; }
    ok = y4 < 4 && (avm & 1); return &S.nb[0][e0 + 4 + (y4 & 3)];
  }
  if (x4 < 4) {
    if (y4 < 0) { ok = (avm >> 1) & 1; return &S.nb[1][e0 + x4]; }
    ok = y4 < 4; return (const u64 *)&(list ? S.mvi1 : S.mvi)[view][(y4 & 3) * 4 + x4][0];
  }
  ok = y4 < 0 && ((avm >> 2) & 1); return &S.nb[2][e0 + 0];
}
__device__ __forceinline__ void mvinfo_get(const u64 *p, bool ok, int &mv, int &ref)
{
  const u64 w = *p;
  mv = ok ? rfl((int)(u32)w) : 0; ref = ok ? rfl((int)(int8_t)(w >> 32)) : -1;
}

// Where mv_predictor finds a block's neighbour: 0..15 = inside the macroblock (4x4 raster index in the chain's view), 16 + 8 n + k = word 9 + k of
// neighbour n's edge record, 255 = never available.  (x4, y4) in [-1, 4] relative to the macroblock, as mvinfo_ptr.
__device__ __forceinline__ u32 nb_code(int x4, int y4)
{
  if (x4 < 0) return y4 < 0 ? 16u + 8u * 3u + 3u : (y4 < 4 ? 16u + 4u + (u32)y4 : 255u);
  if (x4 < 4) return y4 < 0 ? 16u + 8u * 1u + (u32)x4 : (y4 < 4 ? (u32)(y4 * 4 + x4) : 255u);
  return y4 < 0 ? 16u + 8u * 2u : 255u;
}
// the table entry of block type bt (1..7) at (mb_x, mb_y) (samples): get_neighbors (mv_search.c:268-307) incl. the cases where C is not yet coded
__device__ __forceinline__ u32 ptab_entry(int bt, int mb_x, int mb_y)
{
  const int bsx = bt == 1 || bt == 2 ? 16 : (bt == 3 || bt == 4 || bt == 5 ? 8 : 4);
  const u32 a = nb_code((mb_x - 1) >> 2, mb_y >> 2), b = nb_code(mb_x >> 2, (mb_y - 1) >> 2), d = nb_code((mb_x - 1) >> 2, (mb_y - 1) >> 2);
  u32 c = nb_code((mb_x + bsx) >> 2, (mb_y - 1) >> 2);
  if (mb_y > 0) {
    if (mb_x < 8) {
      if (mb_y == 8) { if (bsx == 16) c = 255u; }
      else if (mb_x + bsx == 8) c = 255u;
    } else if (mb_x + bsx == 16) c = 255u;
  }
  return a | (b << 8) | (c << 16) | (d << 24);
}

// get_neighbors (mv_search.c:268-307) + GetMotionVectorPredictorNormal (lcommon/src/mv_prediction.c:194-325): one table read, then the four
// candidates with one LDS read (a lane each)
__device__ __forceinline__ int mv_predictor(const Shared &S, int view, int avm, int ref, int mb_x, int mb_y, int bsx, int bsy, int lane, int list = 0)
{
  const int bt = bsx == 16 ? (bsy == 16 ? 1 : 2) : (bsx == 8 ? (bsy == 16 ? 3 : (bsy == 8 ? 4 : 5)) : (bsy == 8 ? 6 : 7));
  const u32 codes = S.ptab[bt][(mb_y >> 2) * 4 + (mb_x >> 2)];
  const u32 c = (codes >> (8 * (lane & 3))) & 255u;
  const bool inside = c < 16u;
  const int n = (int)(c - 16u) >> 3;
  const bool ok = c != 255u && (inside || ((avm >> n) & 1));
  const u64 *p = inside ? (const u64 *)&(list ? S.mvi1 : S.mvi)[view][c][0] : &S.nb[n & 3][(list ? EDGE_MV1 : 9) + ((c - 16u) & 7u)];
  const u64 w = *p;
  const int mvl = ok ? (int)(u32)w : 0, rfl_ = ok ? (int)(int8_t)(w >> 32) : -1;
  const u32 okm = (u32)__ballot(ok) & 15u;
  int mv[3], rf[3];
  bool av[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { mv[k] = __builtin_amdgcn_readlane(mvl, k); rf[k] = __builtin_amdgcn_readlane(rfl_, k); av[k] = (okm >> k) & 1; }
  if (!av[2]) { mv[2] = __builtin_amdgcn_readlane(mvl, 3); rf[2] = __builtin_amdgcn_readlane(rfl_, 3); av[2] = (okm >> 3) & 1; }
  int type = 0;
  if (rf[0] == ref && rf[1] != ref && rf[2] != ref) type = 1;
  else if (rf[0] != ref && rf[1] == ref && rf[2] != ref) type = 2;
  else if (rf[0] != ref && rf[1] != ref && rf[2] == ref) type = 3;
  if (bsx == 8 && bsy == 16) {
    if (mb_x == 0) { if (rf[0] == ref) type = 1; }
    else { if (rf[2] == ref) type = 3; }
  } else if (bsx == 16 && bsy == 8) {
    if (mb_y == 0) { if (rf[1] == ref) type = 2; }
    else { if (rf[0] == ref) type = 1; }
  }
  if (type == 0) {
    if (!(av[1] || av[2])) return mv[0];
    return mvpack(median3(mvx(mv[0]), mvx(mv[1]), mvx(mv[2])), median3(mvy(mv[0]), mvy(mv[1]), mvy(mv[2])));
  }
  return type == 1 ? mv[0] : (type == 2 ? mv[1] : mv[2]);
}

// FindSkipModeMotionVector mv_search.c:1333-1405
__device__ __forceinline__ int skip_vector(const Shared &S, int view, int avm, int lane)
{
  int mA, rA, mB, rB;
  bool a, b;
  const u64 *pA = mvinfo_ptr(S, view, avm, -1, 0, a), *pB = mvinfo_ptr(S, view, avm, 0, -1, b);
  mvinfo_get(pA, a, mA, rA); mvinfo_get(pB, b, mB, rB);
  const bool zl = !a || (rA == 0 && mA == 0), za = !b || (rB == 0 && mB == 0);
  if (za || zl) return 0;
  return mv_predictor(S, view, avm, 0, 0, 0, 16, 16, lane);
}

// ------------------------------------------------------------------ integer search of one block by one wave
template <int BW>
__device__ __forceinline__ void load_row(const u8 *rowp, int sh, u32 (&b)[BW / 4])
{
  const u32 *rp = (const u32 *)rowp;
  u32 a[BW / 4 + 1];
#pragma unroll
  for (int k = 0; k <= BW / 4; k++) a[k] = rp[k];
#pragma unroll
  for (int k = 0; k < BW / 4; k++) b[k] = __builtin_amdgcn_alignbyte(a[k + 1], a[k], sh);
}

// the largest |d| whose mvbits(d) stay within `bits` (-1: none): mvbits = 1, 3, 5, 7, ... for |d| = 0, 1, 2..3, 4..7, ...
__device__ __forceinline__ int bits_reach(int bits)
{
  const int e = (bits - 3) >> 1;
  return bits < 1 ? -1 : (bits < 3 ? 0 : (e >= 14 ? (1 << 20) : (2 << e) - 1));
}

struct FsCost {                              // what turns a SAD into JM's motion cost
  int lambda, cqx, cqy, pqx, pqy, Rs, R;     // lambda_factor[F_PEL], centre and predictor (quarter-pel), search range of this search and of the slice
  int check00;
  int count;                                 // JMHIP_MB_PROF=11: count rows and candidates
};
__device__ __forceinline__ u64 fs_key(const FsCost &c, int sad, int dx, int dy, int bits_x)
{
  const int vx = c.cqx + 4 * dx, vy = c.cqy + 4 * dy;
  int rate = c.lambda * (bits_x + mvbits(vy - c.pqy));
  if (c.check00 && vx == 0 && vy == 0) rate = rate > 16 * c.lambda ? rate - 16 * c.lambda : 0;       // me_fullsearch.c:78-82
  const u32 cost = (u32)((sad << 5) + rate);
  return ((u64)cost << 32) | ((u64)spiral_index(dx, dy) << 16) | (u64)(((dy + 128) << 8) | (dx + 128));
}

// the compiler may not move the sums' updates across this point (they stay in their registers: no instruction is issued)
template <int N> __device__ __forceinline__ void fence_regs(u32 (&a)[N])
{
  static_assert(N == 4 || N == 8 || N == 16, "block heights");
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
  if constexpr (N >= 8) asm volatile("" : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
  if constexpr (N >= 16) {
    asm volatile("" : "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]));
    asm volatile("" : "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
  }
}

// win: the reference's window in LDS (picture row wy0 + y at row y); cur: the block's rows in LDS (S.cur_y + by * 4 + bx / 4);
// (X0, Y0): window position of the candidate (dx, dy) = (-Rs, -Rs).  Returns the wave-wide minimum key.
// NB candidate rows i0 .. i0 + NB - 1 (those beyond i_last do not count) of one column at once: every window row read before the first is used (left
// alone the compiler funnels the reads through two registers, an LDS round trip each), SADs side by side, keys as in the sliding loop of fs_wave.
// Returns min(bkey, the rows' keys).
template <int BW, int BH, int NB>
__device__ __forceinline__ u32 rows_at_once(const u8 *pbase, int WP, int sh, const u32 (&cb)[BH][BW / 4], const u32 *ytab, int Rs, int i0, int i_last, u32 rx7, int k1, int adx, u32 bkey)
{
  constexpr int NR = NB + BH - 1;
  const u8 *p = pbase + i0 * WP;
  u32 bb[NR][BW / 4], ytv[NB], aa[NR][BW / 4 + 1];
#pragma unroll
  for (int r = 0; r < NR; r++)
#pragma unroll
    for (int k = 0; k <= BW / 4; k++) aa[r][k] = ((const u32 *)(p + r * WP))[k];
#pragma unroll
  for (int k = 0; k < NB; k++) ytv[k] = ytab[i0 + k > 2 * Rs ? 2 * Rs : i0 + k];
#pragma unroll
  for (int r = 0; r < NR; r++)
#pragma unroll
    for (int k = 0; k <= BW / 4; k++) asm volatile("" : "+v"(aa[r][k]));
#pragma unroll
  for (int r = 0; r < NR; r++)
#pragma unroll
    for (int k = 0; k < BW / 4; k++) bb[r][k] = __builtin_amdgcn_alignbyte(aa[r][k + 1], aa[r][k], sh);
  u32 vv[NB];
#pragma unroll
  for (int r = 0; r < BH; r++)                                   // (consecutive instructions belong to NB different sums)
#pragma unroll
    for (int q = 0; q < BW / 4; q++)
#pragma unroll
      for (int k = 0; k < NB; k++) vv[k] = __builtin_amdgcn_sad_u8(bb[k + r][q], cb[r][q], (r == 0 && q == 0) ? 0u : vv[k]);
#pragma unroll
  for (int k = 0; k < NB; k++) {
    const u32 v = vv[k];
    const int i = i0 + k, dy = i - Rs, ady = dy < 0 ? -dy : dy;
    const u32 kf = (v << 12) + rx7 + ytv[k];
    const u32 kn = (kf & ~127u) | (u32)(i + k1);
    const u32 key = (u32)ady <= (u32)adx ? kn : kf;
    bkey = (i <= i_last && key < bkey) ? key : bkey;
  }
  return bkey;
}

// the SAD of ONE candidate (the block at p): four sums side by side (a dependent v_sad_u8 issues only every 9.5 cycles), every row's loads issued before the first is used
template <int BW, int BH>
__device__ __forceinline__ u32 sad_block(const u8 *p, int WP, int sh, const u32 (&cb)[BH][BW / 4])
{
  u32 v[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r0 = 0; r0 < BH; r0 += 4) {
    u32 b[4][BW / 4];
#pragma unroll
    for (int r = 0; r < 4; r++) load_row<BW>(p + (r0 + r) * WP, sh, b[r]);
#pragma unroll
    for (int q = 0; q < BW / 4; q++)
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = __builtin_amdgcn_sad_u8(b[r][q], cb[r0 + r][q], v[r]);
  }
  return (v[0] + v[1]) + (v[2] + v[3]);
}

// spec(mv): called (pruned searches only) with the best integer vector of the first rows -- most often the search's result -- so that the caller can
// start fetching what its sub-pel stage will need.
template <int BW, int BH, class Spec>
__device__ __forceinline__ u64 fs_wave(const u8 *win, int WP, const u32 *cur, int X0, int Y0, const FsCost &c, int lane, u32 *ytab_own, const u32 *ytab4, unsigned long long *pf, int i_lo, int i_hi, bool col64, Spec spec)
{
  u32 cb[BH][BW / 4];
#pragma unroll
  for (int r = 0; r < BH; r++)
#pragma unroll
    for (int k = 0; k < BW / 4; k++) cb[r][k] = BW * BH >= 128 ? cur[r * 4 + k] : (u32)__builtin_amdgcn_readfirstlane(cur[r * 4 + k]);
  // candidate rows i_lo .. i_hi of the (2 Rs + 1) x (2 Rs + 1) positions (all of them, or one of two waves' halves)
  const int Rs = c.Rs, ncol = 2 * Rs + 1;
  u64 best = ~0ull;
  if (pf && lane == 0) pf[25] = wall_clock64();
  int dbg_steps = 0, dbg_items = 0;                             // JMHIP_MB_PROF: window rows the sliding lanes read, candidates of step 2
  // JM skips a candidate whose vector cost alone reaches the running minimum (me_fullsearch.c:83) -- result-neutral, and what makes its full search
  // affordable on a CPU.  Here the same bound, for the blocks of at most 64 samples (a larger block's SAD dwarfs any vector cost: the bound
  // excludes nothing; nor for the 16x16 search on reference 0, whose (0,0) vector's cost is not its rate), in three steps.  A candidate whose
  // vector cost EXCEEDS a cost already seen can neither hold the minimum nor tie with it:
  //   0. the five candidate rows nearest the predictor, every column (a lane slides down its column)            -> bound B0
  //   1. the rows within h of the predictor's row that B0 leaves, every column, the same way                     -> bound B1 <= B0
  //   2. the rows beyond, as far as B1 leaves any: only the columns whose bits + the cheapest such row's bits stay within B1 -- few, so
  //      one lane per candidate (plain SADs).
  // h is the one of 2, 3, 7, 15, 31, all that makes steps 1 + 2 cheapest (vector bits double their reach every two bits, hence the values).
  const int t_y = c.pqy - c.cqy, t_x = c.pqx - c.cqx;           // the predictor relative to the search centre, quarter-pel
  const int ip = clampi3(i_lo, i_hi, Rs + ((t_y + 2) >> 2));   // the row nearest the predictor
  // (an 8x8 block's SAD << 5 dwarfs its vectors' cost unless lambda is large: below 400 -- QP 34 or so -- the bound leaves every row, and the rows of step 0 with their
  // lead-in are read twice: one pass over everything is cheaper)
  const bool prune = (BW * BH < 64 || (BW * BH == 64 && c.lambda >= 400)) && !c.check00 && c.lambda > 0;
  constexpr int NBH = BH == 4 ? 3 : 2;                          // step 0: rows within NBH of the predictor's (a block four rows high pays little for two more, and is then mostly done)
  const int a_lo = prune ? max(i_lo, ip - NBH) : i_lo, a_hi = prune ? min(i_hi, ip + NBH) : i_hi;
  const int bx_min = mvbits(t_x - 4 * clampi3(-Rs, Rs, (t_x + 2) >> 2));       // the cheapest column's bits
  int lo1 = i_lo, hi1 = i_hi;                                  // step 1's rows, set after step 0
  bool step1 = false;
  int d_lo2 = 0, d_hi2 = -1, d_dqx2 = -1;                      // what the decision after step 0 found (step 2 uses it as it is when step 1 read nothing)
  int ev_lo = a_lo, ev_hi = a_hi;                              // hull of the rows read by the sliding lanes
  u32 bound = 0xffffffffu;                                     // the best cost so far
  bool fast = false;                                           // the rows of step 0 settled the search (below)
  const float rcp_lambda = __builtin_amdgcn_rcpf((float)(c.lambda > 0 ? c.lambda : 1));
  {                                                            // columns 0..63: a lane slides down its column
    // A lane beyond the last column repeats the last column's candidates (same keys: harmless).  The column keeps ONE 32-bit key per
    // candidate: (cost << 7) | rank, rank = the candidate's place among the column's candidates in JM's spiral order (rows -|dx| .. |dx|
    // downwards first, then -(|dx| + 1), +(|dx| + 1), ...: mv_search.c:405-442), so that the unsigned minimum is JM's strict '<' scan
    // restricted to the column.  cost < 2^25: SAD << 5 < 2^21, rate < 2^22 with the lambda the host accepts.
    const int col = lane < ncol ? lane : ncol - 1;
    const int dx = col - Rs, adx = dx < 0 ? -dx : dx, xl = X0 + col, sh = xl & 3;
    const u8 *pbase = win + (xl & ~3) + Y0 * WP;
    const int vx = c.cqx + 4 * dx;
    const u32 rate_x = (u32)(c.lambda * mvbits(vx - c.pqx));
    const bool zero_x = c.check00 && vx == 0;
    u32 bkey = 0xffffffffu;
    // the row table (one LDS read per row instead of a dozen scalar instructions): (rate of the row) << 7 | far rank.  It depends on the row's
    // distance from the predictor only: with the predictor at its usual -2 .. 1 quarter-pels from the centre the launch's own tables serve.
    const u32 *ytab = ytab4 + (t_y + 2) * 72 + (c.R - Rs);
    if (t_y < -2 || t_y > 1) {
      for (int i = lane; i <= 2 * Rs; i += 64) {
        const int dy = i - Rs, vy = c.cqy + 4 * dy;
        ytab_own[i] = ((u32)(c.lambda * mvbits(vy - c.pqy)) << 7) | ((u32)(2 * (dy < 0 ? -dy : dy) - 1 + (dy > 0 ? 1 : 0)) & 127u);
      }
      wave_sync();
      ytab = ytab_own;
    }
    if (pf && lane == 0) pf[24] = wall_clock64();
    const u32 rx7 = rate_x << 7;
    const int k1 = adx - Rs;                                   // rank of a row within |dx| of the centre row: dy + |dx| = i + k1
    u32 sad00 = 0;
    for (int seg = 0; seg < 3; seg++) {                        // the rows around the predictor, then what the bound leaves above and below them
      // (with lead-ins of seven rows one pass over both sides and the rows between them is cheaper than two)
      const bool both = BH >= 8 && lo1 < a_lo && hi1 > a_hi;
      const int s_lo = seg == 0 ? a_lo : (seg == 1 ? lo1 : a_hi + 1), s_hi = seg == 0 ? a_hi : (seg == 1 ? (both ? hi1 : a_lo - 1) : (both ? a_hi : hi1));
      if (BW * BH <= 64 && seg == 0 && prune) {
        // step 0 as straight-line code: the (at most) five or seven candidate rows and the BH - 1 window rows below them are read at once and
        // summed side by side (independent instructions: a lone wave issues a dependent one only every ~9 cycles); rows past the last candidate
        // lie in the window's slack rows
        bkey = rows_at_once<BW, BH, 2 * NBH + 1>(pbase, WP, sh, cb, ytab, Rs, a_lo, a_hi, rx7, k1, adx, bkey);
        dbg_steps += 2 * NBH + BH;
      } else if (BW * BH <= 32 && prune && s_lo <= s_hi) {
        // step 1 of the smallest blocks the same way, eight candidate rows at a time (twice as fast per row as the sliding loop below, although
        // BH - 1 window rows are read again per chunk)
        ev_lo = min(ev_lo, s_lo); ev_hi = max(ev_hi, s_hi);
        for (int i0 = s_lo; i0 <= s_hi; i0 += 8) {
          bkey = rows_at_once<BW, BH, 8>(pbase, WP, sh, cb, ytab, Rs, i0, s_hi, rx7, k1, adx, bkey);
          dbg_steps += 8 + BH - 1;
        }
      } else if (s_lo <= s_hi) {
        ev_lo = min(ev_lo, s_lo); ev_hi = max(ev_hi, s_hi);
        const int nrows = (s_hi - s_lo + 1) + BH - 1;
        dbg_steps += nrows;
        const u8 *p = pbase + s_lo * WP;
        u32 acc[BH];
#pragma unroll
        for (int k = 0; k < BH; k++) acc[k] = 0;
        // window rows and row-table entries travel PF rows ahead of their use (an LDS read takes a couple of hundred cycles; a row of a small
        // block is summed in far less)
        constexpr int PF = BW * BH <= 64 ? 4 : (BW * BH <= 128 ? 2 : 1);
        u32 a[PF][BW / 4 + 1], yt[PF];
#pragma unroll
        for (int f = 0; f < PF; f++) {
#pragma unroll
          for (int k = 0; k <= BW / 4; k++) a[f][k] = ((const u32 *)(p + f * WP))[k];
          const int i = s_lo + f - (BH - 1);
          yt[f] = ytab[i < 0 ? 0 : (i > 2 * Rs ? 2 * Rs : i)];
        }
        // key = ((SAD << 5) + rate_x + rate_y) << 7 | rank.  With the row's (rate_y << 7 | far rank) from the table that is one shift-add and one add
        // for the rows beyond |dx| (far rank); for the rows within, the low seven bits are replaced by dy + |dx|.  Rows that end no candidate of this segment give ~0.
        auto key_of = [&](u32 fin_, u32 ytf_, int i) {           // i: the candidate row, the same for every lane
          const int dy = i - Rs, ady = dy < 0 ? -dy : dy;
          const u32 kf = (fin_ << 12) + rx7 + ytf_;
          const u32 kn = (kf & ~127u) | (u32)(i + k1);
          const u32 key = (u32)ady <= (u32)adx ? kn : kf;
          if (BW == 16 && BH == 16) sad00 = (c.check00 && c.cqy + 4 * dy == 0 && i >= s_lo) ? fin_ : sad00;     // the (0,0) vector's row, for its bonus below
          return (i >= s_lo && i <= s_hi) ? key : 0xffffffffu;
        };
        constexpr bool IMM = BH >= 16;                           // sixteen-row blocks: a row's key at once (64 / 32 sums lie between two keys: nothing waits), not 48 registers of them
        for (int j0 = 0; j0 < nrows; j0 += BH) {               // rows past the last one (a partial final group) lie in the window's slack rows
          u32 fin[IMM ? 1 : BH], ytf[IMM ? 1 : BH];
#pragma unroll
          for (int s = 0; s < BH; s++) {
            const int j = j0 + s;
            u32 b[BW / 4];
#pragma unroll
            for (int k = 0; k < BW / 4; k++) b[k] = __builtin_amdgcn_alignbyte(a[s % PF][k + 1], a[s % PF][k], sh);
            const u32 yts = yt[s % PF];
            {                                                  // row j + PF into the slot just freed
              const u8 *pn = p + (j + PF) * WP;
#pragma unroll
              for (int k = 0; k <= BW / 4; k++) a[s % PF][k] = ((const u32 *)pn)[k];
              const int in = s_lo + j - (BH - 1) + PF;
              yt[s % PF] = ytab[in < 0 ? 0 : (in > 2 * Rs ? 2 * Rs : in)];
            }
            __builtin_amdgcn_sched_barrier(0);                 // keeps the compiler from sinking the read-ahead to where the data is needed
            // window row j is row r of the candidate that starts at row j - r.  Dword column by dword column, so that consecutive instructions belong to BH different
            // sums (a lone wave issues a dependent v_sad_u8 only every 9.5 cycles: profiles/microbench/issue_rate.hip), and fenced row by row: left alone the compiler
            // keeps sixteen aligned window rows in registers and sums a 16-row block candidate by candidate -- 64 dependent instructions in a row (round 6)
#pragma unroll
            for (int q = 0; q < BW / 4; q++)
#pragma unroll
              for (int r = 0; r < BH; r++) {
                const int k = (s - r + BH) % BH;
                acc[k] = __builtin_amdgcn_sad_u8(b[q], cb[r][q], (q == 0 && r == 0) ? 0u : acc[k]);
              }
            fence_regs<BH>(acc);
            if constexpr (IMM) {
              const u32 key = key_of(acc[(s + 1) % BH], yts, s_lo + j - (BH - 1));
              bkey = key < bkey ? key : bkey;
            } else { fin[s] = acc[(s + 1) % BH]; ytf[s] = yts; }   // the candidate that ends with this row
          }
          if constexpr (!IMM) {
            // the group's keys side by side: BH independent chains (a lone wave issues a dependent instruction only every ~9 cycles), no branch
            u32 gk[BH];
#pragma unroll
            for (int s = 0; s < BH; s++) gk[s] = key_of(fin[s], ytf[s], s_lo + j0 + s - (BH - 1));
#pragma unroll
            for (int w = BH / 2; w >= 1; w >>= 1)
#pragma unroll
              for (int s = 0; s < w; s++) gk[s] = gk[s] < gk[s + w] ? gk[s] : gk[s + w];
            bkey = gk[0] < bkey ? gk[0] : bkey;
          }
        }
      }
      if (seg == 0 && prune) {
        if (pf && lane == 0) pf[27] = wall_clock64();
        const u32 mk = wave_min_u32(bkey);
        bound = mk >> 7;
        {
          const int gl = __ffsll((unsigned long long)__ballot(bkey == mk)) - 1, gdx = (gl < ncol ? gl : ncol - 1) - Rs, gadx = gdx < 0 ? -gdx : gdx;
          const int grank = (int)(mk & 127u), gh = (grank + 1) >> 1, gdy = grank <= 2 * gadx ? grank - gadx : ((grank & 1) ? -gh : gh);
          spec(mvpack(c.cqx + 4 * gdx, c.cqy + 4 * gdy));
        }
        if (pf && lane == 0) pf[28] = wall_clock64();
        const int kq = (int)((float)bound * rcp_lambda) + 1;    // a candidate is worth reading while its vector's bits do not exceed bound / lambda (rounded up: safe)
        const int dq = bits_reach(kq - bx_min);                // ... a row, while |vy - py| <= dq
        const int lo2 = dq < 0 ? i_hi + 1 : max(i_lo, Rs + ((t_y - dq + 3) >> 2)), hi2 = dq < 0 ? i_lo - 1 : min(i_hi, Rs + ((t_y + dq) >> 2));
        // The usual case of the small blocks: no row beyond the ones just read is worth reading, and the 65th column's cheapest candidate costs more than the bound too -- the
        // search is over, its smallest cost IS the bound: steps 1 and 2, the 65th column and the first of the two reductions at the end are skipped (round 6)
        if (lo2 >= a_lo && hi2 <= a_hi) {
          bool c64_out = !(col64 && ncol > 64);
          if (!c64_out) {
            const int dyc = clampi3(-Rs, Rs, (t_y + 2) >> 2);
            c64_out = (u32)c.lambda * (u32)(mvbits(c.cqx + 4 * (64 - Rs) - c.pqx) + mvbits(t_y - 4 * dyc)) > bound;
          }
          if (c64_out) { fast = true; break; }
        }
        // h = 2 (no step 1) when step 2 then fits one pass -- the usual case, decided with a few scalar instructions
        int byo2 = 1 << 20;
        if (a_lo - 1 >= lo2) byo2 = min(byo2, mvbits(4 * (a_lo - 1 - Rs) - t_y));
        if (a_hi + 1 <= hi2) byo2 = min(byo2, mvbits(4 * (a_hi + 1 - Rs) - t_y));
        const int dqx2 = bits_reach(kq - byo2);
        const int nc2 = dqx2 < 0 ? 0 : max(0, min(2 * Rs, Rs + ((t_x + dqx2) >> 2)) - max(0, Rs + ((t_x - dqx2 + 3) >> 2)) + 1);
        d_lo2 = lo2; d_hi2 = hi2; d_dqx2 = dqx2;
        if ((max(0, a_lo - lo2) + max(0, hi2 - a_hi)) * nc2 <= 64) { lo1 = a_lo; hi1 = a_hi; }
        else {
          // the cheapest h, one lane per choice: sliding steps of step 1 (with their BH - 1 rows of lead-in) against passes of step 2
          const int hk = lane == 0 ? 2 : (2 << (lane < 5 ? lane : 5)) - 1;
          const int l1 = max(lo2, ip - hk), h1 = min(hi2, ip + hk);
          int byo = 1 << 20;
          if (l1 - 1 >= lo2) byo = min(byo, mvbits(4 * (l1 - 1 - Rs) - t_y));
          if (h1 + 1 <= hi2) byo = min(byo, mvbits(4 * (h1 + 1 - Rs) - t_y));
          const int dqx = bits_reach(kq - byo);
          const int nc = dqx < 0 ? 0 : max(0, min(2 * Rs, Rs + ((t_x + dqx) >> 2)) - max(0, Rs + ((t_x - dqx + 3) >> 2)) + 1);
          const int passes = ((max(0, l1 - lo2) + max(0, hi2 - h1)) * nc + 63) >> 6;
          const int up = max(0, a_lo - l1), dn = max(0, h1 - a_hi);
          const int steps = (BH >= 8 && up && dn) ? (h1 - l1 + 1) + BH - 1 : (up ? up + BH - 1 : 0) + (dn ? dn + BH - 1 : 0);
          const u32 est = (u32)(steps * (BW * BH / 4 + 12) + passes * (BH * (BW / 2 + 3) + 45));
          const u32 pick = (u32)rfl((int)row16_min_u32(lane < 6 ? (est << 3) | (u32)lane : 0xffffffffu)) & 7u;
          const int h = pick == 0 ? 2 : (2 << pick) - 1;
          lo1 = max(lo2, ip - h); hi1 = min(hi2, ip + h);
          step1 = lo1 < a_lo || hi1 > a_hi;
        }
        if (pf && lane == 0) pf[29] = wall_clock64();
      }
    }
    if (pf && lane == 0) pf[30] = wall_clock64();
    if (BW == 16 && BH == 16 && zero_x) {                       // me_fullsearch.c:78-82: the (0,0) vector of the 16x16 search on reference 0 gets 16 lambda off its rate
      const int i0 = Rs - (c.cqy >> 2);                        // its row, if the window holds it
      if (i0 >= i_lo && i0 <= i_hi) {
        const int dy = i0 - Rs, ady = dy < 0 ? -dy : dy;
        const u32 rate = rate_x + (u32)(c.lambda * mvbits(0 - c.pqy)), t16 = 16u * (u32)c.lambda;
        const u32 rank = ady <= adx ? (u32)(dy + adx) : (u32)(2 * ady - 1 + (dy > 0 ? 1 : 0));
        const u32 key = (((sad00 << 5) + (rate > t16 ? rate - t16 : 0u)) << 7) | rank;
        bkey = key < bkey ? key : bkey;
      }
    }
    if (prune && step1) bound = wave_min_u32(bkey) >> 7;        // B1 (= B0 if step 1 read nothing)
    {
      const int rank = (int)(bkey & 127u);
      const int l = (rank + 1) >> 1;
      const int dy = rank <= 2 * adx ? rank - adx : ((rank & 1) ? -l : l);
      best = ((u64)(bkey >> 7) << 32) | ((u64)spiral_index(dx, dy) << 16) | (u64)(((dy + 128) << 8) | (dx + 128));
    }
  }
  if (pf && lane == 0) pf[7] = wall_clock64();
  for (int col = 64; col < ((col64 && !fast) ? ncol : 64); col++) {           // the columns beyond the wave (SearchRange 32: one): lane = row, plain SADs
    const int dx = col - Rs, xl = X0 + col, sh = xl & 3;
    const int bits_x = mvbits(c.cqx + 4 * dx - c.pqx);
    {                                                          // the same bound: the column's bits + the cheapest row's against the best cost so far
      const int dyc = clampi3(-Rs, Rs, (t_y + 2) >> 2);
      if (bound != 0xffffffffu && (u32)c.lambda * (u32)(bits_x + mvbits(t_y - 4 * dyc)) > bound) continue;
    }
    for (int i0 = ev_lo; i0 <= ev_hi; i0 += 64) {              // the rows the sliding lanes read
      const int i = i0 + lane;
      const bool live = i <= ev_hi;
      const u8 *p = win + (xl & ~3) + (Y0 + (live ? i : ev_hi)) * WP;
      u32 v = sad_block<BW, BH>(p, WP, sh, cb);
      if (live) { const u64 key = fs_key(c, (int)v, dx, i - Rs, bits_x); best = key < best ? key : best; }
    }
  }
  if (prune && !fast) {                                        // step 2: the rows beyond the hull, one lane per candidate
    int r_lo = d_lo2, r_hi = d_hi2, dqx = d_dqx2;
    if (step1) {                                               // a tighter bound since: once more
      const int kq = (int)((float)bound * rcp_lambda) + 1;
      const int dq = bits_reach(kq - bx_min);
      r_lo = dq < 0 ? i_hi + 1 : max(i_lo, Rs + ((t_y - dq + 3) >> 2)); r_hi = dq < 0 ? i_lo - 1 : min(i_hi, Rs + ((t_y + dq) >> 2));
      int byo = 1 << 20;
      if (ev_lo - r_lo > 0) byo = min(byo, mvbits(4 * (ev_lo - 1 - Rs) - t_y));
      if (r_hi - ev_hi > 0) byo = min(byo, mvbits(4 * (ev_hi + 1 - Rs) - t_y));
      dqx = bits_reach(kq - byo);
    }
    const int n_up = max(0, ev_lo - r_lo), n_dn = max(0, r_hi - ev_hi), nr = n_up + n_dn;
    const int cl = max(0, Rs + ((t_x - dqx + 3) >> 2)), ch = min(2 * Rs, Rs + ((t_x + dqx) >> 2));
    const int total = (dqx < 0 || ch < cl) ? 0 : nr * (ch - cl + 1);
    const float inv = __builtin_amdgcn_rcpf((float)(nr > 0 ? nr : 1));
    dbg_items = total;
    for (int base = 0; base < total; base += 64) {
      const int item = base + lane;
      const bool live = item < total;
      const int it = live ? item : 0;
      const int ci = (int)(((float)it + 0.5f) * inv), ri = it - ci * nr;       // exact: it < 2^13, the fraction is at least 1 / (2 nr) off an integer
      const int i = ri < n_up ? r_lo + ri : ev_hi + 1 + (ri - n_up);
      const int col = cl + ci, dx = col - Rs, xl = X0 + col, sh = xl & 3;
      const u8 *p = win + (xl & ~3) + (Y0 + i) * WP;
      u32 v = sad_block<BW, BH>(p, WP, sh, cb);
      if (live) { const u64 key = fs_key(c, (int)v, dx, i - Rs, mvbits(c.cqx + 4 * dx - c.pqx)); best = key < best ? key : best; }
    }
  }
  if (c.count && lane == 0) { ytab_own[70] += (u32)dbg_steps; ytab_own[71] += (u32)dbg_items; }     // JMHIP_MB_PROF=11: per-wave totals (spare entries of the wave's row table)
  if (c.count && lane == 0) ytab_own[68] += (u32)(dbg_steps * 64 + dbg_items) * (u32)(BW * BH / 16);            // ... and the absolute differences issued, in units of 16 (a window row read by the 64 sliding lanes is BH rows of BW samples each)
  if (pf && lane == 0) { pf[23] = wall_clock64(); pf[26] = (unsigned long long)dbg_steps | ((unsigned long long)dbg_items << 16) | ((unsigned long long)(hi1 - lo1 + 1 > 0 ? hi1 - lo1 + 1 : 0) << 32); }
  const u32 hi = fast ? bound : wave_min_u32((u32)(best >> 32));              // the smallest cost, then the earliest spiral index among the candidates that have it
  const u32 lo = wave_min_u32((u32)(best >> 32) == hi ? (u32)best : 0xffffffffu);
  return ((u64)hi << 32) | lo;
}

// ------------------------------------------------------------------ Hadamard SATD of a 4x4 block (HadamardSAD4x4 me_distortion.c:175-258)
__device__ __forceinline__ int hadamard4(const int (&d)[16])
{
  int m[16], s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int s0 = d[4 * i] + d[4 * i + 3], s1 = d[4 * i + 1] + d[4 * i + 2], s2 = d[4 * i + 1] - d[4 * i + 2], s3 = d[4 * i] - d[4 * i + 3];
    m[4 * i] = s0 + s1; m[4 * i + 1] = s0 - s1; m[4 * i + 2] = s2 + s3; m[4 * i + 3] = s3 - s2;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int s0 = m[i] + m[12 + i], s1 = m[4 + i] + m[8 + i], s2 = m[4 + i] - m[8 + i], s3 = m[i] - m[12 + i];
    s += iabs_(s0 + s1) + iabs_(s0 - s1) + iabs_(s2 + s3) + iabs_(s3 - s2);
  }
  return (s + 1) >> 1;
}
// source rows (dwords) against reference rows (dwords)
__device__ __forceinline__ int satd4_rows(const u32 (&o)[4], const u32 (&r)[4])
{
  int d[16];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) d[4 * j + i] = (int)((o[j] >> (8 * i)) & 255) - (int)((r[j] >> (8 * i)) & 255);
  return hadamard4(d);
}
// the four rows of the 4x4 block at absolute quarter-pel position (qx, qy) of reference `ref`, UMVLine4X origin clamp (refbuf.h:22-26)
__device__ __forceinline__ void ref_rows4(const PipeArgs &A, int ref, int qx, int qy, u32 (&r)[4])
{
  const int iy = clampi3(-JMHIP_PAD_Y, A.H + 3, qy >> 2), ix = clampi3(-JMHIP_PAD_X, A.W + 15, qx >> 2);
  // (32-bit offsets: the sixteen planes of a 2160p picture are 140 MB)
  const u32 off = (u32)((qy & 3) * 4 + (qx & 3)) * (u32)A.plane_stride + (u32)(iy + JMHIP_PAD_Y) * (u32)A.ref_pitch + (u32)(ix + JMHIP_PAD_X);
  const u8 *p = PV.ref_y[ref];
#pragma unroll
  for (int j = 0; j < 4; j++) r[j] = ldref32(p + (off + (u32)j * (u32)A.ref_pitch));
}

// spiral positions 0..8 (mv_search.c:405-442): {0,0}, {0,-1}, {0,1}, {-1,-1}, {1,-1}, {-1,0}, {1,0}, {-1,1}, {1,1} as two packed constants (a table in
// memory would put a global load on every sub-pel stage's critical path)
__device__ __forceinline__ int sp9x(int k) { return (int)((0x22215u >> (2 * k)) & 3u) - 1; }
__device__ __forceinline__ int sp9y(int k) { return (int)((0x29421u >> (2 * k)) & 3u) - 1; }

// one stage of sub_pel_motion_estimation (me_fullsearch.c:221-246 / :263-281): the nine SATDs around mv with the given step, into S.red[wave][0..8]
__device__ __forceinline__ void subpel_satds(Shared &S, const PipeArgs &A, int wave, int lane, int ref, int px, int py, int mb_x, int mb_y, int bw4, int bh4, int mv, int step)
{
  const int nb4 = bw4 * bh4, items = 9 * nb4;
  for (int base = 0; base < items; base += 64) {
    const int item = base + lane;
    const bool live = item < items;
    // bw4, bh4 are 1, 2 or 4: shifts and masks, not divisions
    const int lw = bw4 >> 1, lh = bh4 >> 1;                    // log2
    const int it = live ? item : 0, cnd = it >> (lw + lh), b = it & (nb4 - 1), sbx = (b & (bw4 - 1)) * 4, sby = (b >> lw) * 4;
    const int qx = ((px + sbx) << 2) + mvx(mv) + sp9x(cnd) * step, qy = ((py + sby) << 2) + mvy(mv) + sp9y(cnd) * step;
    u32 r[4], o[4];
    ref_rows4(A, ref, qx, qy, r);
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = S.cur_y[(mb_y + sby + j) * 4 + ((mb_x + sbx) >> 2)];
    int v = live ? satd4_rows(o, r) : 0;
    v = group_sum(v, nb4);
    if (live && b == 0) S.red[wave][cnd] = v;
  }
  wave_sync();
}

// (A 4x4 block's sub-pel search reads all 49 SATDs within three quarter-pels of the integer vector at once, a lane each -- block_search -- so that
// the quarter-pel stage does not wait for a second trip to the sub-pel planes.)
// The strict-'<' scan over the nine positions of one stage of sub_pel_motion_estimation (me_fullsearch.c:221-246 / :263-281), nine lanes at once:
// JM skips a position whose vector cost alone reaches the running minimum -- such a position could not have won -- so the winner is the first
// position with the smallest total cost.  first = 1 (the quarter-pel stage when both stages use the same metric): position 0 is not evaluated,
// it stands for the minimum the half-pel stage left (`incumbent`).  SATD sums in S.red[wave][0..8].
__device__ __forceinline__ int scan9(const Shared &S, int wave, int lane, int mv, int step, int lambda, int pqx, int pqy, int bonus0, int incumbent, int first, int &min_out, bool grid = false, int mv0 = 0)
{
  const int pos = lane < 9 ? lane : 0;
  const int qx = mvx(mv) + step * sp9x(pos), qy = mvy(mv) + step * sp9y(pos);
  // grid: the SATDs lie in a 7 x 7 grid of quarter-pel offsets -3 .. 3 around the integer vector mv0 (subpel_grid49)
  const int idx = grid ? (qy - mvy(mv0) + 3) * 7 + (qx - mvx(mv0) + 3) : pos;
  int cost = lambda * (mvbits(qx - pqx) + mvbits(qy - pqy)) + (S.red[wave][idx] << 5);
  if (pos == 0) cost = first ? incumbent : cost - bonus0;
  u32 key = lane < 9 ? (((u32)(cost + (1 << 27))) << 4) | (u32)pos : 0xffffffffu;       // costs stay far below 2^27 in magnitude
  key = (u32)rfl((int)row16_min_u32(key));
  min_out = (int)(key >> 4) - (1 << 27);
  return (int)(key & 15u);
}

#include "mbpipe_had8.inc"

// BlockMotionSearch (mv_search.c:857-1024) of one (block, reference) by one wave; RDOptimization = 0, SearchMode = -1.
// view: which chain's picture of the macroblock's vectors the predictor reads; wave: this wave's own scratch (row table, SATD sums).
template <bool T8>
__device__ __forceinline__ int block_search(Shared &S, const PipeArgs &A, const u8 *wins, int view, int wave, int avm, int lane, int ref, int bt, int mb_x, int mb_y, int mbx, int mby, int &out_mv,
                                            int list = 0, int cr = -1, int widx = -1, int *pred_out = nullptr)
{
  // B slices: `ref` is the index within `list` (the predictor compares it with the neighbours' reference indices of that list); cr = the reference's number through both
  // lists (planes, vectors), widx = its window in LDS.  P slices: list 0, cr = widx = ref.
  const jmhip_slice_params &P = A.p;
  if (cr < 0) cr = ref;
  if (widx < 0) widx = ref;
  const int BW = bt == 1 || bt == 2 ? 16 : (bt == 3 || bt == 4 || bt == 5 ? 8 : 4);
  const int BH = bt == 1 || bt == 3 ? 16 : (bt == 2 || bt == 4 || bt == 6 ? 8 : 4);
  const int addr_ = mby * A.wmb + mbx, px = mbx * 16 + mb_x, py = mby * 16 + mb_y;     // (the macroblock's position comes from the caller: a division per search otherwise)
  const int R = P.search_range;
  const bool prof_ = APROF && (((A.prof_mode == 1 || A.prof_mode == 10) && mb_x == 0 && bt == 7) || (A.prof_mode >= 5 && A.prof_mode <= 9 && mb_x == 0 && bt == (A.prof_mode == 5 ? 2 : (A.prof_mode == 6 ? 1 : A.prof_mode - 3)))) &&
                     mb_y == 0 && ref == 0 && lane == 0;       // modes 5 .. 9: the first 16x8 / 16x16 / 8x8 / 8x4 / 4x8 search
#define BS_STAMP(k) do { if (prof_) APROF[(long)addr_ * 32 + (k)] = wall_clock64(); } while (0)
  BS_STAMP(18);
  const int pred = rfl(mv_predictor(S, view, avm, ref, mb_x, mb_y, BW, BH, lane, list));
  if (pred_out) *pred_out = pred;
  BS_STAMP(19);
  int cx = ((mvx(pred) + 2) >> 2) * 4, cy = ((mvy(pred) + 2) >> 2) * 4;          // mv_search.c:931-932
  int min_x = -(R << 2), max_x = R << 2, min_y = min_x, max_y = max_x;
  {
    const int ox = cx, oy = cy;
    cx = clampi3(min_x, max_x, cx); cy = clampi3(min_y, max_y, cy);              // :949-950
    if (cx != ox || cy != oy) {                                                  // CheckSearchRange :822-849
      const int lim = P.max_mvd - 2;
      int left = clampi3(ox - lim, ox + lim, cx + min_x), right = clampi3(ox - lim, ox + lim, cx + max_x);
      int top = clampi3(oy - lim, oy + lim, cy + min_y), down = clampi3(oy - lim, oy + lim, cy + max_y);
      if (left < right && top < down) {
        cx = (left + right) >> 1; cy = (top + down) >> 1;
        max_x = min(cx - left, right - cx); max_y = min(cy - top, down - cy);
      } else { cx = ox; cy = oy; }
    }
  }
  cx = clampi3(P.mv_limit[0], P.mv_limit[1], cx); cy = clampi3(P.mv_limit[2], P.mv_limit[3], cy);      // clip_mv_range :957
  if (cx < -(R << 2) || cx > (R << 2) || cy < -(R << 2) || cy > (R << 2) || (cx & 3) || (cy & 3)) {   // outside the staged windows: cannot happen
    if (lane == 0) S.err = 2;                                                                          // with the ranges accepted by the host
    cx = clampi3(-(R << 2), R << 2, cx) & ~3; cy = clampi3(-(R << 2), R << 2, cy) & ~3;
  }
  FsCost c;
  c.lambda = P.lambda_mf[0]; c.cqx = cx; c.cqy = cy; c.pqx = mvx(pred); c.pqy = mvy(pred);
  c.Rs = min(max(min(max_x, max_y) >> 2, 0), R); c.R = R;
  c.check00 = bt == 1 && ref == 0 && P.slice_type != 1;         // me_fullsearch.c:78: not in B slices
  const bool ffs = P.search_mode == 1;
  if (ffs) {
    // fast_full_search_motion_estimation (me_fullfast.c:618-689, rdopt == 0): every block of the macroblock is searched around ONE centre per reference, the
    // rounded 16x16 predictor (setup_fast_full_search :310-328; its neighbours lie outside the macroblock, so any chain's view gives it), over
    // imax(max_x, max_y) >> 2 rings (:633); the (0,0) vector is tried first (below); the max_mvd guard (:638, :671) cannot fire (checked by the host)
    const int p16 = rfl(mv_predictor(S, view, avm, ref, 0, 0, 16, 16, lane, list));
    cx = clampi3(-(R << 2), R << 2, ((mvx(p16) + 2) >> 2) * 4); cy = clampi3(-(R << 2), R << 2, ((mvy(p16) + 2) >> 2) * 4);
    cx = clampi3(P.mv_limit[0] + (R << 2), P.mv_limit[1] - (R << 2), cx); cy = clampi3(P.mv_limit[2] + (R << 2), P.mv_limit[3] - (R << 2), cy);
    c.cqx = cx; c.cqy = cy; c.Rs = min(max(max(max_x, max_y) >> 2, 0), R); c.check00 = 0;
  }
  c.count = APROF != nullptr && A.prof_mode == 11;
  const u8 *win = wins + (size_t)widx * A.win_h * A.win_p;
  const int X0 = mb_x + (cx >> 2) - c.Rs + A.win_ox, Y0 = mb_y + (cy >> 2) - c.Rs + 2 * R;
  const u32 *cur = S.cur_y + mb_y * 4 + (mb_x >> 2);
  const int i_lo = 0, i_hi = 2 * c.Rs;
  u64 key;
  // the 4x4 searches (the macroblock's longest dependent chain) fetch their sub-pel neighbourhood while the rest of the integer search runs
  u32 sr[4] = {0, 0, 0, 0};
  int spec_mv = 0x7fffffff;
  const int g_oy = (lane < 49 ? lane : 0) / 7 - 3, g_ox = (lane < 49 ? lane : 0) - (g_oy + 3) * 7 - 3;
  auto nospec = [](int) {};
  auto spec4 = [&](int mvg) {
    if (!P.subpel) return;
    spec_mv = mvg;
    ref_rows4(A, cr, (px << 2) + mvx(mvg) + g_ox, (py << 2) + mvy(mvg) + g_oy, sr);
    __builtin_amdgcn_sched_barrier(0);
  };
  switch (bt) {
  case 1: key = fs_wave<16, 16>(win, A.win_p, cur, X0, Y0, c, lane, S.ytab[wave], S.ytab4[0], prof_ ? APROF + (long)addr_ * 32 : nullptr, i_lo, i_hi, true, nospec); break;
  case 2: key = fs_wave<16, 8>(win, A.win_p, cur, X0, Y0, c, lane, S.ytab[wave], S.ytab4[0], prof_ ? APROF + (long)addr_ * 32 : nullptr, i_lo, i_hi, true, nospec); break;
  case 3: key = fs_wave<8, 16>(win, A.win_p, cur, X0, Y0, c, lane, S.ytab[wave], S.ytab4[0], prof_ ? APROF + (long)addr_ * 32 : nullptr, i_lo, i_hi, true, nospec); break;
  case 4: key = fs_wave<8, 8>(win, A.win_p, cur, X0, Y0, c, lane, S.ytab[wave], S.ytab4[0], prof_ ? APROF + (long)addr_ * 32 : nullptr, i_lo, i_hi, true, nospec); break;
  case 5: key = fs_wave<8, 4>(win, A.win_p, cur, X0, Y0, c, lane, S.ytab[wave], S.ytab4[0], prof_ ? APROF + (long)addr_ * 32 : nullptr, i_lo, i_hi, true, nospec); break;
  case 6: key = fs_wave<4, 8>(win, A.win_p, cur, X0, Y0, c, lane, S.ytab[wave], S.ytab4[0], prof_ ? APROF + (long)addr_ * 32 : nullptr, i_lo, i_hi, true, nospec); break;
  default: key = fs_wave<4, 4>(win, A.win_p, cur, X0, Y0, c, lane, S.ytab[wave], S.ytab4[0], prof_ ? APROF + (long)addr_ * 32 : nullptr, i_lo, i_hi, true, spec4); break;
  }
  const int klo = rfl((int)(u32)key), khi = rfl((int)(u32)(key >> 32));
  BS_STAMP(20);
  int mv = mvpack(cx + 4 * ((klo & 255) - 128), cy + 4 * (((klo >> 8) & 255) - 128));
  int min_mcost = khi;
  if (ffs) {                                                   // the (0,0) vector first (me_fullfast.c:650-657): it holds the minimum against any later equal cost
    const u8 *p00 = win + (size_t)(mb_y + 2 * R) * A.win_p + mb_x + A.win_ox;       // the block's own position in the window
    u32 v = 0;
    if (lane < BH) {
      for (int k = 0; k < BW / 4; k++) v = __builtin_amdgcn_sad_u8(ldu32(p00 + lane * A.win_p + 4 * k), cur[lane * 4 + k], v);
    }
    const int sad00 = rfl(group_sum((int)v, 16));
    const int cost00 = (sad00 << 5) + c.lambda * (mvbits(0 - c.pqx) + mvbits(0 - c.pqy));
    if (cost00 <= min_mcost) { min_mcost = cost00; mv = 0; }
  }

  if (P.subpel) {                                             // sub_pel_motion_estimation me_fullsearch.c:186-289 (start_me_refinement_hp = 0)
    const int check0 = ref == 0 && bt == 1 && mv == 0 && P.slice_type != 1;
    const bool grid = bt == 7;
    const int mv0 = mv;
    const bool had8 = T8 && bt <= 4;                           // mv_block.test8x8 (mv_search.c:1624, :1768): 8x8 Hadamard sub-blocks in computeSATD
    auto sx9 = [](int k) { return sp9x(k); };
    auto sy9 = [](int k) { return sp9y(k); };
    if (grid) {                                               // subpel_grid49 with the rows fetched ahead, if the guess held
      if (mv != spec_mv) ref_rows4(A, cr, (px << 2) + mvx(mv) + g_ox, (py << 2) + mvy(mv) + g_oy, sr);
      u32 o[4];
#pragma unroll
      for (int j = 0; j < 4; j++) o[j] = S.cur_y[(mb_y + j) * 4 + (mb_x >> 2)];
      if (lane < 49) S.red[wave][lane] = satd4_rows(o, sr);
      wave_sync();
    }
    else if (had8) subpel_satds8(S, A, S.red[wave], lane, cr, px, py, mb_x, mb_y, BW / 8, BH / 8, mv, 2, sx9, sy9);
    else subpel_satds(S, A, wave, lane, cr, px, py, mb_x, mb_y, BW / 4, BH / 4, mv, 2);
    int best = scan9(S, wave, lane, mv, 2, P.lambda_mf[1], c.pqx, c.pqy, check0 ? P.lambda_mf[1] * 16 : 0, 0, 0, min_mcost, grid, mv0);
    mv = mvpack(mvx(mv) + 2 * sp9x(best), mvy(mv) + 2 * sp9y(best));
    BS_STAMP(21);
    if (had8) subpel_satds8(S, A, S.red[wave], lane, cr, px, py, mb_x, mb_y, BW / 8, BH / 8, mv, 1, sx9, sy9);
    else if (!grid) subpel_satds(S, A, wave, lane, cr, px, py, mb_x, mb_y, BW / 4, BH / 4, mv, 1);
    best = scan9(S, wave, lane, mv, 1, P.lambda_mf[2], c.pqx, c.pqy, 0, min_mcost, P.start_qp, min_mcost, grid, mv0);
    mv = mvpack(mvx(mv) + sp9x(best), mvy(mv) + sp9y(best));
  }
  mv = mvpack(clampi3(P.mv_limit[0], P.mv_limit[1], mvx(mv)), clampi3(P.mv_limit[2], P.mv_limit[3], mvy(mv)));   // :981

  if (bt == 1 && P.slice_type == 0) {                          // the skip vector against the 16x16 result: mv_search.c:983-998, GetSkipCostMB :1257
    const int sv = rfl(skip_vector(S, view, avm, lane));
    if (lane == 0) S.skip_mv = sv;
    const int qx = (mbx * 64) + mvx(sv), qy = (mby * 64) + mvy(sv);
    const int iy = clampi3(-JMHIP_PAD_Y, A.H + 3, qy >> 2), ix = clampi3(-JMHIP_PAD_X, A.W + 15, qx >> 2);     // one origin for the 16x16 block
    const int b = lane & 15, bx = (b & 3) * 4, by = (b >> 2) * 4;
    const u8 *p = PV.ref_y[0] + (long)((qy & 3) * 4 + (qx & 3)) * A.plane_stride + (long)(iy + JMHIP_PAD_Y + by) * A.ref_pitch + ix + JMHIP_PAD_X + bx;
    u32 r[4], o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { r[j] = ldref32(p + (long)j * A.ref_pitch); o[j] = S.cur_y[(by + j) * 4 + (bx >> 2)]; }
    int v = lane < 16 ? satd4_rows(o, r) : 0;
    v = rfl(group_sum(v, 16));
    if (T8) {                                                  // GetSkipCostMB with Transform8x8Mode: distortion8x8 of the four 8x8 blocks of the 16x16 prediction (one origin)
      int d[64];
      const int b8 = lane & 3, x8 = (b8 & 1) * 8, y8 = (b8 >> 1) * 8;
      const u8 *p8 = PV.ref_y[0] + (long)((qy & 3) * 4 + (qx & 3)) * A.plane_stride + (long)(iy + JMHIP_PAD_Y + y8) * A.ref_pitch + ix + JMHIP_PAD_X + x8;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const u32 r0 = ldref32(p8 + (long)j * A.ref_pitch), r1 = ldref32(p8 + (long)j * A.ref_pitch + 4), o0 = S.cur_y[(y8 + j) * 4 + (x8 >> 2)], o1 = S.cur_y[(y8 + j) * 4 + (x8 >> 2) + 1];
#pragma unroll
        for (int i = 0; i < 4; i++) { d[8 * j + i] = (int)((o0 >> (8 * i)) & 255) - (int)((r0 >> (8 * i)) & 255); d[8 * j + 4 + i] = (int)((o1 >> (8 * i)) & 255) - (int)((r1 >> (8 * i)) & 255); }
      }
      int v8 = lane < 4 ? hadamard8(d) : 0;
      v = rfl(group_sum(v8, 4));
    }
    const int cost = (v << 5) - P.lambda_mf[2] * 8;
    if (cost < min_mcost) { min_mcost = cost; mv = sv; }
  }
  BS_STAMP(22);
  out_mv = mv;
  return min_mcost;
}

// set_me_parameters (mv_search.c:100-113) on this wave's view
__device__ __forceinline__ void set_mvi(Shared &S, int wave, int lane, int mv, int ref, int x4, int y4, int w4, int h4)
{
  if (lane < 16) {
    const int x = lane & 3, y = lane >> 2;
    if (x >= x4 && x < x4 + w4 && y >= y4 && y < y4 + h4) { S.mvi[wave][lane][0] = mv; S.mvi[wave][lane][1] = ref; }
  }
  wave_sync();
}

// list_prediction_cost (mode_decision.c:275, LIST_0) with update_mcost (:253)
__device__ __forceinline__ int list0_cost(const Shared &S, const jmhip_slice_params &P, int mode, int block, int &bref)
{
  const int ref_lambda = P.lambda_mf[2] >> 2;
  int bm = MAXC;
  for (int ref = 0; ref < P.num_ref; ref++) {
    int mc = rfl(S.mcost[mode][ref][block]);
    if (mc < bm) {
      mc += P.num_ref <= 1 ? 0 : ref_lambda * P.refbits[ref];
      if (mc < bm) { bm = mc; bref = ref; }
    }
  }
  return bm;
}

// the largest level the entropy coder takes: CAVLC_LEVEL_LIMIT with CAVLC (quant4x4_normal.c:84), none with CABAC
__device__ __forceinline__ int lev_max_of(const jmhip_slice_params &P) { return P.symbol_mode == 0 ? 2063 : 0x7fffffff; }

// ------------------------------------------------------------------ transform / quantisation of one 4x4 block in registers
__device__ __forceinline__ void fwd4_(int &a, int &b, int &c, int &d) { int e0 = a + d, e1 = b + c, o0 = b - c, o1 = a - d; a = e0 + e1; b = (o1 << 1) + o0; c = e0 - e1; d = o1 - (o0 << 1); }
__device__ __forceinline__ void inv4_(int &a, int &b, int &c, int &d) { int e0 = a + c, e1 = a - c, o0 = (b >> 1) - d, o1 = b + (d >> 1); a = e0 + o1; b = e1 + o0; c = e1 - o0; d = e0 - o1; }
__device__ __forceinline__ void forward4x4(int (&m)[16])
{
#pragma unroll
  for (int i = 0; i < 4; i++) fwd4_(m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]);
#pragma unroll
  for (int i = 0; i < 4; i++) fwd4_(m[i], m[4 + i], m[8 + i], m[12 + i]);
}
__device__ __forceinline__ void inverse4x4(int (&m)[16])
{
#pragma unroll
  for (int i = 0; i < 4; i++) inv4_(m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]);
#pragma unroll
  for (int i = 0; i < 4; i++) inv4_(m[i], m[4 + i], m[8 + i], m[12 + i]);
}

// quant_4x4_normal / quant_ac4x4_normal (quant4x4_normal.c:39 / :117) on the transformed block m (raster); first = 0 or 1 (AC only).
// lev[16]: levels at their scan positions; m receives the dequantised coefficients.  Returns nonzero; cost accumulates the coefficient cost.
__device__ __forceinline__ int quant4x4(int (&m)[16], const jmhip_qparam *q, int qp_per, int first, int16_t *lev, int &cost, int lev_max)
{
  const int q_bits = 15 + qp_per;
  int run = 0, nz = 0;
  int qs[16], qo[16], qi[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { qo[k] = q[k].OffsetComp; qs[k] = q[k].ScaleComp; qi[k] = q[k].InvScaleComp; }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    if (k < first) { lev[k] = 0; continue; }
    constexpr int ZZ[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
    const int idx = ZZ[k], cf = m[idx];
    int l = 0;
    if (cf != 0) {
      l = (iabs_(cf) * qs[idx] + qo[idx]) >> q_bits;
      if (l != 0) {
        l = min(l, lev_max);                                   // CAVLC_LEVEL_LIMIT, or none (CABAC)
        cost += l > 1 ? 999999 : (run == 0 ? 3 : (run <= 2 ? 2 : (run <= 5 ? 1 : 0)));       // COEFF_COST4x4[0][run], block.c:72
        l = cf < 0 ? -l : l;
        m[idx] = (((l * qi[idx]) << qp_per) + 8) >> 4;
        run = 0; nz = 1;
      } else { m[idx] = 0; run++; }
    } else run++;
    lev[k] = (int16_t)l;
  }
  return nz;
}
// residual_transform_quant_luma_4x4 (block.c:661-725): o, p = source and prediction rows; rec = reconstructed rows
__device__ __forceinline__ int tq_luma4(const u32 (&o)[4], const u32 (&p)[4], const jmhip_qparam *q, int qp_per, int16_t *lev, int &cost, u32 (&rec)[4], int lev_max)
{
  int m[16], pr[16], any = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) { pr[k] = (p[k >> 2] >> (8 * (k & 3))) & 255; m[k] = (int)((o[k >> 2] >> (8 * (k & 3))) & 255) - pr[k]; any |= m[k]; }
  int nz = 0;
  if (any) { forward4x4(m); nz = quant4x4(m, q, qp_per, 0, lev, cost, lev_max); }
  else {
#pragma unroll
    for (int k = 0; k < 16; k++) lev[k] = 0;
  }
  if (nz) {
    inverse4x4(m);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      u32 w = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) w |= (u32)clampi3(0, 255, ((m[4 * j + i] + 32) >> 6) + pr[4 * j + i]) << (8 * i);
      rec[j] = w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) rec[j] = p[j];
  }
  return nz;
}

// The same for a block that EVERY lane of the wave holds alike (the Intra4x4 chain: one block after the other): the sixteen coefficients are
// quantised by sixteen lanes at once (lane = scan position) instead of a sixteen-step chain in every lane; the dequantised coefficients come
// back to all lanes by wave shuffles.  lev: sixteen int16 in LDS (written by the sixteen lanes).  No coefficient cost (Intra4x4 does not use it).
__device__ __forceinline__ int tq_luma4_wave(const u32 (&o)[4], const u32 (&p)[4], const jmhip_qparam *q, int qp_per, int lane, int16_t *lev, u32 (&rec)[4], int lev_max)
{
  constexpr int ZZ[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
  const int k = lane & 15;
  int zz = 0;
#pragma unroll
  for (int c = 0; c < 16; c++) zz = (c == k) ? ZZ[c] : zz;
  const jmhip_qparam qk = q[zz];                               // on its way while the transform runs
  int m[16], pr[16];
#pragma unroll
  for (int c = 0; c < 16; c++) { pr[c] = (p[c >> 2] >> (8 * (c & 3))) & 255; m[c] = (int)((o[c >> 2] >> (8 * (c & 3))) & 255) - pr[c]; }
  forward4x4(m);                                               // an all-zero residual transforms to zeros and quantises to zeros: no special case needed
  int cf = 0;
#pragma unroll
  for (int c = 0; c < 16; c++) cf = (c == zz) ? m[c] : cf;
  int l = (iabs_(cf) * qk.ScaleComp + qk.OffsetComp) >> (15 + qp_per);
  l = min(l, lev_max);                                         // CAVLC_LEVEL_LIMIT, or none (CABAC)
  l = cf < 0 ? -l : l;
  const int dq = (((l * qk.InvScaleComp) << qp_per) + 8) >> 4;
  if (lane < 16) lev[k] = (int16_t)l;
  const int nz = (__ballot(l != 0) & 0xffffull) != 0;
#pragma unroll
  for (int c = 0; c < 16; c++) m[ZZ[c]] = __shfl(dq, c, 64);
  if (nz) {
    inverse4x4(m);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      u32 w = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) w |= (u32)clampi3(0, 255, ((m[4 * j + i] + 32) >> 6) + pr[4 * j + i]) << (8 * i);
      rec[j] = w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) rec[j] = p[j];
  }
  return nz;
}

// ------------------------------------------------------------------ intra prediction samples
// get_intrapred_4x4 (intra4x4.c:521; modes :72-308): e[0] = above left, e[1..8] = above and above right, e[9..12] = left
__device__ __forceinline__ int e4t(const u8 *e, int x) { return x < 0 ? e[0] : e[1 + x]; }
__device__ __forceinline__ int e4l(const u8 *e, int y) { return y < 0 ? e[0] : e[9 + y]; }
__device__ int ipred4_sample(const u8 *e, int mode, int x, int y, int left, int up)
{
  switch (mode) {
  case 0: return e4t(e, x);
  case 1: return e4l(e, y);
  case 2:
    if (up && left) return (e[1] + e[2] + e[3] + e[4] + e[9] + e[10] + e[11] + e[12] + 4) >> 3;
    if (left) return (e[9] + e[10] + e[11] + e[12] + 2) >> 2;
    if (up) return (e[1] + e[2] + e[3] + e[4] + 2) >> 2;
    return e[1];
  case 3: return (x == 3 && y == 3) ? (e4t(e, 6) + 3 * e4t(e, 7) + 2) >> 2 : (e4t(e, x + y) + 2 * e4t(e, x + y + 1) + e4t(e, x + y + 2) + 2) >> 2;
  case 4:
    if (x > y) return (e4t(e, x - y - 2) + 2 * e4t(e, x - y - 1) + e4t(e, x - y) + 2) >> 2;
    if (x < y) return (e4l(e, y - x - 2) + 2 * e4l(e, y - x - 1) + e4l(e, y - x) + 2) >> 2;
    return (e4t(e, 0) + 2 * e[0] + e4l(e, 0) + 2) >> 2;
  case 5: {
    const int z = 2 * x - y, k = x - (y >> 1);
    if (z >= 0 && !(z & 1)) return (e4t(e, k - 1) + e4t(e, k) + 1) >> 1;
    if (z > 0) return (e4t(e, k - 2) + 2 * e4t(e, k - 1) + e4t(e, k) + 2) >> 2;
    if (z == -1) return (e4l(e, 0) + 2 * e[0] + e4t(e, 0) + 2) >> 2;
    return (e4l(e, y - 1) + 2 * e4l(e, y - 2) + e4l(e, y - 3) + 2) >> 2; }
  case 6: {
    const int z = 2 * y - x, k = y - (x >> 1);
    if (z >= 0 && !(z & 1)) return (e4l(e, k - 1) + e4l(e, k) + 1) >> 1;
    if (z > 0) return (e4l(e, k - 2) + 2 * e4l(e, k - 1) + e4l(e, k) + 2) >> 2;
    if (z == -1) return (e4l(e, 0) + 2 * e[0] + e4t(e, 0) + 2) >> 2;
    return (e4t(e, x - 1) + 2 * e4t(e, x - 2) + e4t(e, x - 3) + 2) >> 2; }
  case 7: {
    const int k = x + (y >> 1);
    return (y & 1) ? (e4t(e, k) + 2 * e4t(e, k + 1) + e4t(e, k + 2) + 2) >> 2 : (e4t(e, k) + e4t(e, k + 1) + 1) >> 1; }
  default: {
    const int z = x + 2 * y, k = y + (x >> 1);
    if (z > 5) return e4l(e, 3);
    if (z == 5) return (e4l(e, 2) + 3 * e4l(e, 3) + 2) >> 2;
    if (z & 1) return (e4l(e, k) + 2 * e4l(e, k + 1) + e4l(e, k + 2) + 2) >> 2;
    return (e4l(e, k) + e4l(e, k + 1) + 1) >> 1; }
  }
}

// luma sample of the reconstruction at (x, y) relative to the macroblock, for intra prediction: inside from `own` (16x16 bytes), outside
// from the neighbours' edge records (words 0..1 bottom row, 2..3 right column)
__device__ __forceinline__ int edge_byte(const u64 *w, int k) { return (int)((w[k >> 3] >> (8 * (k & 7))) & 255); }
__device__ __forceinline__ int rec_luma_at(const Shared &S, const u8 *own, int x, int y)
{
  if (x < 0) return y < 0 ? edge_byte(&S.nb[3][0], 15) : edge_byte(&S.nb[0][2], y);
  if (x < 16) return y < 0 ? edge_byte(&S.nb[1][0], x) : own[y * 16 + x];
  return edge_byte(&S.nb[2][0], x - 16);
}

// JM's hadamard4x4 (transform.c:121-168): rows, then columns with >> 1
__device__ __forceinline__ void hadamard4x4_jm(int (&m)[16])
{
  int u[16];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int e0 = m[4 * i] + m[4 * i + 3], e1 = m[4 * i + 1] + m[4 * i + 2], o0 = m[4 * i + 1] - m[4 * i + 2], o1 = m[4 * i] - m[4 * i + 3];
    u[4 * i] = e0 + e1; u[4 * i + 1] = o1 + o0; u[4 * i + 2] = e0 - e1; u[4 * i + 3] = o1 - o0;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int e0 = u[i] + u[12 + i], e1 = u[4 + i] + u[8 + i], o0 = u[4 + i] - u[8 + i], o1 = u[i] - u[12 + i];
    m[i] = (e0 + e1) >> 1; m[4 + i] = (o0 + o1) >> 1; m[8 + i] = (e0 - e1) >> 1; m[12 + i] = (o1 - o0) >> 1;
  }
}
// ihadamard4x4 (transform.c:170-218): no scaling
__device__ __forceinline__ void ihadamard4x4_jm(int (&m)[16])
{
  int u[16];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int e0 = m[4 * i] + m[4 * i + 2], e1 = m[4 * i] - m[4 * i + 2], o0 = m[4 * i + 1] - m[4 * i + 3], o1 = m[4 * i + 1] + m[4 * i + 3];
    u[4 * i] = e0 + o1; u[4 * i + 1] = e1 + o0; u[4 * i + 2] = e1 - o0; u[4 * i + 3] = e0 - o1;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int e0 = u[i] + u[8 + i], e1 = u[i] - u[8 + i], o0 = u[4 + i] - u[12 + i], o1 = u[4 + i] + u[12 + i];
    m[i] = e0 + o1; m[4 + i] = e1 + o0; m[8 + i] = e1 - o0; m[12 + i] = e0 - o1;
  }
}

// ------------------------------------------------------------------ staging a reference's integer window in LDS
// Thread t of NT copies dwords t, t + NT, ... of the window (win_h rows of win_p bytes: picture rows wy0 .., columns wx0 ..; beyond the picture the edge samples).  (round 6)
// Sixteen loads in flight per thread: the loop used to wait for every dword before it stored it -- fourteen trips to the L2 one after the other, 6 of the staging's 7 us.  A step
// of NT dwords through the window is qn rows and rn dwords (no division per dword); a dword that crosses the picture's left or right edge (rare: the clamped load is issued
// with the others, the edge bytes are fetched afterwards) goes the old way.
__device__ __forceinline__ void stage_window(const PipeArgs &A, const u8 *pl, u32 *w, int wx0, int wy0, int t, int NT)
{
  const int dpr = A.win_p >> 2, nd = A.win_h * dpr;
  const int qn = NT / dpr, rn = NT - qn * dpr;
  constexpr int CH = 16;
  int y = t / dpr, xq = t - y * dpr;
  for (int d0 = t; d0 < nd; d0 += CH * NT) {
    u32 v[CH], edge = 0;
#pragma unroll
    for (int k = 0; k < CH; k++) {
      const int py = clampi3(0, A.H - 1, wy0 + y), px = wx0 + 4 * xq;
      edge |= (px >= 0 && px + 3 < A.W) ? 0u : 1u << k;
      v[k] = ldref32(pl + (long)py * A.ref_pitch + clampi3(0, A.W - 4, px));
      xq += rn; y += qn;
      if (xq >= dpr) { xq -= dpr; y++; }
    }
#pragma unroll
    for (int k = 0; k < CH; k++) {
      const int d = d0 + k * NT;
      if (d < nd) {
        u32 vv = v[k];
        if ((edge >> k) & 1u) {
          const int yy = d / dpr, px = wx0 + (d - yy * dpr) * 4;
          const u8 *row = pl + (long)clampi3(0, A.H - 1, wy0 + yy) * A.ref_pitch;
          vv = ldref8(row + clampi3(0, A.W - 1, px)) | (ldref8(row + clampi3(0, A.W - 1, px + 1)) << 8) | (ldref8(row + clampi3(0, A.W - 1, px + 2)) << 16) | (ldref8(row + clampi3(0, A.W - 1, px + 3)) << 24);
        }
        w[d] = vv;
      }
    }
  }
}

#include "mbpipe_intra.inc"
#include "mbpipe_t8.inc"
#include "mbpipe_final.inc"
#include "mbpipe_epzs.inc"
#include "mbpipe_post.inc"
#include "mbpipe_kernel.inc"
