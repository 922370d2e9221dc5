"""ctypes binding of libjmhip.so (include/jmhip.h).

Mirrors the reference's operator surface for the hot path with JM's names in the docstrings:
set_reference = getSubImagesLuma, me_fullsearch = full_search_motion_estimation /
fast_full_search_motion_estimation, me_subpel = sub_pel_motion_estimation, tq_luma4x4 =
residual_transform_quant_luma_4x4, deblock_frame = DeblockFrame.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NPART = 41
PAD_X, PAD_Y = 32, 20

# (blocktype, bx, by, w, h) of the 41 partitions in ABI order (units: luma samples)
PARTITIONS = ([(1, 0, 0, 16, 16)] + [(2, 0, by, 16, 8) for by in (0, 8)] + [(3, bx, 0, 8, 16) for bx in (0, 8)] +
              [(4, bx, by, 8, 8) for by in (0, 8) for bx in (0, 8)] +
              [(5, bx, by, 8, 4) for by in (0, 4, 8, 12) for bx in (0, 8)] +
              [(6, bx, by, 4, 8) for by in (0, 8) for bx in (0, 4, 8, 12)] +
              [(7, bx, by, 4, 4) for by in (0, 4, 8, 12) for bx in (0, 4, 8, 12)])
assert len(PARTITIONS) == NPART

ME_JOB = np.dtype([("mb_x", "<i2"), ("mb_y", "<i2"), ("center_x", "<i2"), ("center_y", "<i2"), ("search_range", "<i2"),
                   ("max_mvd", "<i2"), ("lambda", "<i4"), ("part_mask", "<u8"), ("pred", "<i2", (NPART, 2)), ("reserved_", "<i2", (2,))])
ME_BEST = np.dtype([("mv_x", "<i2"), ("mv_y", "<i2"), ("cost", "<i4")])
ME_RESULT = np.dtype([("best", ME_BEST, (NPART,))])
CAND = np.dtype([("pos_x", "<i2"), ("pos_y", "<i2"), ("bsx", "<i2"), ("bsy", "<i2"), ("cand_x", "<i2"), ("cand_y", "<i2"),
                 ("metric", "<i2"), ("test8x8", "<i2")])
PRED_CAND = np.dtype([("pos_x", "<i2"), ("pos_y", "<i2"), ("bsx", "<i2"), ("bsy", "<i2"), ("cand_x", "<i2", (2,)), ("cand_y", "<i2", (2,)),
                      ("slot", "i1", (2,)), ("metric", "i1"), ("test8x8", "i1"), ("pred", "i1"), ("shift", "i1"),
                      ("weight", "<i2", (2,)), ("offset", "<i2"), ("round", "<i2"), ("reserved_", "<i2")])
PRED_AVG, PRED_BI_WP, PRED_UNI_WP, PRED_UNI = 0, 1, 2, 3
SUBPEL_JOB = np.dtype([("pos_x", "<i2"), ("pos_y", "<i2"), ("bsx", "<i2"), ("bsy", "<i2"), ("pred_x", "<i2"), ("pred_y", "<i2"),
                       ("mv_x", "<i2"), ("mv_y", "<i2"), ("lambda_h", "<i4"), ("lambda_q", "<i4"), ("metric_h", "i1"),
                       ("metric_q", "i1"), ("start_hp", "i1"), ("start_qp", "i1"), ("test8x8", "i1"), ("reserved_", "i1", (3,)),
                       ("min_mcost", "<i4")])
REFINE_PARAMS = np.dtype([("lambda_h", "<i4"), ("lambda_q", "<i4"), ("metric_h", "i1"), ("metric_q", "i1"), ("start_hp", "i1"),
                          ("start_qp", "i1"), ("transform8x8_mode", "<i4")])
TQ_PARAMS = np.dtype([("q", "<i4", (16, 3)), ("qp_per", "<i4"), ("cavlc", "<i4"), ("adaptive_rounding", "<i4"),
                      ("adapt_rnd_weight", "<i4"), ("max_pel", "<i4"), ("reserved_", "<i4", (3,))])
TQ_OUT = np.dtype([("level", "<i2", (16,)), ("run", "u1", (16,)), ("coeff_cost", "<i4"), ("nonzero", "u1"), ("any_residual", "u1"),
                   ("ncoef", "u1"), ("reserved_", "u1"), ("rec", "u1", (16,)), ("fadjust", "<i2", (16,))])
TQ8_PARAMS = np.dtype([("q", "<i4", (64, 3)), ("qp_per", "<i4"), ("cavlc", "<i4"), ("adaptive_rounding", "<i4"),
                       ("adapt_rnd_weight", "<i4"), ("max_pel", "<i4"), ("reserved_", "<i4", (3,))])
TQ8_OUT = np.dtype([("level", "<i2", (68,)), ("run", "u1", (68,)), ("coeff_cost", "<i4"), ("nonzero", "u1"), ("any_residual", "u1"),
                    ("ncoef", "u1", (4,)), ("reserved_", "u1", (2,)), ("rec", "u1", (64,)), ("fadjust", "<i2", (64,))])
TQC_PARAMS = np.dtype([("q_ac", "<i4", (16, 3)), ("q_dc", "<i4", (3,)), ("qp_per_ac", "<i4"), ("qp_per_dc", "<i4"), ("yuv_format", "<i4"),
                       ("cavlc", "<i4"), ("adaptive_rounding", "<i4"), ("adapt_rnd_weight", "<i4"), ("max_pel", "<i4"), ("reserved_", "<i4", (2,))])
TQC_MB = np.dtype([("cbp_blk", "<i8"), ("cr_cbp", "<i4"), ("uv", "<i4")])
TQC_OUT = np.dtype([("dc_level", "<i2", (9,)), ("dc_run", "u1", (9,)), ("dc_nonzero", "u1"), ("ac_level", "<i2", (8, 16)), ("ac_run", "u1", (8, 16)),
                    ("ac_ncoef", "u1", (8,)), ("rec", "u1", (128,)), ("fadjust", "<i2", (128,)), ("reserved_", "u1", (4,))])
assert TQC_PARAMS.itemsize == 240 and TQC_MB.itemsize == 16 and TQC_OUT.itemsize == 808
DC_OUT = np.dtype([("level", "<i2", (17,)), ("run", "u1", (17,)), ("nonzero", "u1")])
TQ16_OUT = np.dtype([("rec", "u1", (256,)), ("fadjust", "<i2", (4, 16)), ("ac_level", "<i2", (16, 16)), ("dc_level", "<i2", (17,)), ("ac_run", "u1", (16, 16)),
                     ("ac_ncoef", "u1", (16,)), ("dc_run", "u1", (17,)), ("dc_nonzero", "u1"), ("ac_coef", "u1"), ("reserved_", "u1", (3,))])
assert TQ16_OUT.itemsize == 1224
assert TQ8_PARAMS.itemsize == 800 and TQ8_OUT.itemsize == 408 and DC_OUT.itemsize == 52
DC_KINDS = {"hadamard4x4": (0, 16), "ihadamard4x4": (1, 16), "hadamard4x2": (2, 8), "ihadamard4x2": (3, 8), "hadamard2x2": (4, 4), "ihadamard2x2": (5, 4)}
DB_MB = np.dtype([("mb_type", "<i2"), ("slice_type", "<i2"), ("qp", "<i2"), ("qpc", "<i2", (2,)), ("cbp", "<i2"), ("cbp_blk", "<u4"),
                  ("slice_nr", "<i2"), ("df_disable_idc", "<i2"), ("df_alpha_c0", "<i2"), ("df_beta", "<i2"), ("transform8x8", "<i2"),
                  ("reserved_", "<i2")])
DB_MOTION = np.dtype([("mv", "<i2", (2, 2)), ("ref_id", "<i4", (2,))])
MC_LUMA_BLK = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("dir", "u1"), ("reserved_", "u1"), ("slot", "i1", (2,)),
                        ("mv", "<i2", (2, 2)), ("reserved2_", "<i2")])
MC_CHROMA_BLK = np.dtype([("x", "<i2"), ("y", "<i2"), ("dir", "u1"), ("plane", "u1"), ("slot", "i1", (2,)), ("mv", "<i2", (2, 4, 2, 2))])
MC_WEIGHTS = np.dtype([("weight", "<i2", (2,)), ("offset", "<i2"), ("round", "<i2"), ("shift", "i1"), ("reserved_", "i1", (3,))])
assert MC_LUMA_BLK.itemsize == 20 and MC_CHROMA_BLK.itemsize == 72 and MC_WEIGHTS.itemsize == 12
IP4_BLK = np.dtype([("edge", "u1", (13,)), ("mode", "u1"), ("left", "u1"), ("up", "u1")])
I16_MB = np.dtype([("edge", "u1", (33,)), ("left", "u1"), ("up", "u1"), ("mode_mask", "u1"), ("metric", "u1"), ("reserved_", "u1", (3,))])
I16_OUT = np.dtype([("cost", "<i8"), ("mode", "<i4"), ("reserved_", "<i4"), ("pred", "u1", (4, 256))])
IP8_BLK = np.dtype([("edge", "u1", (25,)), ("mode", "u1"), ("left", "u1"), ("up", "u1")])
assert IP8_BLK.itemsize == 28
IC_MB = np.dtype([("up", "u1", (2, 8)), ("left", "u1", (2, 16)), ("corner", "u1", (2,)), ("up_avail", "u1"), ("left_avail", "u1"), ("upleft_avail", "u1"),
                  ("reserved_", "u1", (3,))])
assert IP4_BLK.itemsize == 16 and I16_MB.itemsize == 40 and I16_OUT.itemsize == 1040 and IC_MB.itemsize == 56
MB_MAX_REF = 16
MB_RECORD = np.dtype([("mb_type", "i1"), ("i16mode", "i1"), ("c_ipred_mode", "i1"), ("transform8x8", "i1"), ("cbp", "<i2"), ("reserved1_", "<i2"),
                      ("cbp_blk", "<u8"), ("min_rdcost", "<i8"), ("b8mode", "i1", (4,)), ("b8ref", "i1", (4,)), ("ipredmode", "i1", (16,)),
                      ("ipred_syntax", "i1", (16,)), ("mv", "<i2", (16, 2)), ("luma", "<i2", (16, 16)), ("luma_dc", "<i2", (16,)),
                      ("chroma_dc", "<i2", (2, 8)), ("chroma_ac", "<i2", (2, 8, 16)),
                      ("mv1", "<i2", (16, 2)), ("b8ref1", "i1", (4,)), ("b8pdir", "i1", (4,)), ("b8bipred", "i1", (4,)), ("reserved2_", "i1", (4,))])      # B slices
FRAME_FORMAT = np.dtype([("yuv_format", "<i4"), ("src_w", "<i4"), ("src_h", "<i4"), ("out_w", "<i4"), ("out_h", "<i4"), ("coded_w", "<i4"), ("coded_h", "<i4"),
                         ("symbol_bytes", "<i4"), ("src_depth", "<i4", (3,)), ("out_depth", "<i4", (3,))])
# jmhip_seq_picture (include/jmhip.h): 8-byte pointers, the record pointer behind 36 ints
SEQ_PICTURE = np.dtype({"names": ["d_raw", "src_w", "src_h", "out_slot", "ref_slot", "ref_id", "poc_offset", "d_records"],
                        "formats": ["<u8", "<i4", "<i4", "<i4", ("<i4", (16,)), ("<i4", (16,)), "<i4", "<u8"],
                        "offsets": [0, 8, 12, 16, 20, 84, 148, 152], "itemsize": 160})
SLICE_PARAMS = np.dtype([("slice_type", "<i4"), ("first_mb", "<i4"), ("num_mb", "<i4"), ("slice_nr", "<i4"), ("qp", "<i4"), ("qpc", "<i4"),
                         ("search_range", "<i4"), ("num_ref", "<i4"), ("ref_slot", "<i4", (MB_MAX_REF,)), ("ref_id", "<i4", (MB_MAX_REF,)),
                         ("lambda_mf", "<i4", (3,)), ("lambda_mdfp", "<i4"), ("max_mvd", "<i4"), ("mv_limit", "<i4", (4,)),
                         ("inter_valid", "<i4", (8,)), ("intra4_valid", "<i4"), ("intra16_valid", "<i4"), ("subpel", "<i4"), ("start_qp", "<i4"),
                         ("refbits", "<i4", (MB_MAX_REF,)), ("q_luma", "<i4", (2, 16, 3)), ("q_chroma", "<i4", (2, 2, 16, 3)),
                         ("df_disable_idc", "<i4"), ("df_alpha_c0", "<i4"), ("df_beta", "<i4"), ("num_slices", "<i4"), ("symbol_mode", "<i4"), ("search_mode", "<i4"), ("qpc_cr_delta", "<i4"), ("num_ref1", "<i4"),
                         ("epzs_pattern", "<i4"), ("epzs_dual", "<i4"), ("epzs_fixed", "<i4"), ("epzs_aggressive", "<i4"), ("epzs_temporal", "<i4"), ("epzs_spatial_mem", "<i4"),
                         ("epzs_blocktype", "<i4"), ("epzs_min_scale", "<i4"), ("epzs_med_scale", "<i4"), ("epzs_max_scale", "<i4"), ("epzs_sub_scale", "<i4"), ("b_switches", "<i4"),
                         ("poc_cur", "<i4"), ("poc_ref", "<i4", (MB_MAX_REF,)),
                         ("transform8x8", "<i4"), ("intra8_valid", "<i4"), ("q_luma8", "<i4", (2, 64, 3)), ("q_chroma_dc", "<i4", (2, 2, 3))])
assert MB_RECORD.itemsize == 1296 and SLICE_PARAMS.itemsize == 3200
assert ME_JOB.itemsize == 192 and ME_RESULT.itemsize == 328 and SUBPEL_JOB.itemsize == 36 and TQ_OUT.itemsize == 104
assert TQ_PARAMS.itemsize == 224 and DB_MB.itemsize == 28 and DB_MOTION.itemsize == 16 and CAND.itemsize == 16 and PRED_CAND.itemsize == 32

EXPORTS = ["jmhip_create", "jmhip_destroy", "jmhip_last_error", "jmhip_synchronize", "jmhip_set_stream", "jmhip_plane_geometry",
           "jmhip_set_current", "jmhip_set_current_dev", "jmhip_set_current_frame", "jmhip_set_current_frame_dev", "jmhip_load_frame", "jmhip_load_frame_dev", "jmhip_set_current_planes", "jmhip_current_planes_dev", "jmhip_get_current_planes", "jmhip_set_reference", "jmhip_set_reference_dev",
           "jmhip_get_subplanes", "jmhip_subplanes_dev", "jmhip_me_fullsearch", "jmhip_me_fullsearch_dev",
           "jmhip_me_sad_tables", "jmhip_me_eval", "jmhip_me_eval_pred", "jmhip_me_eval_pred_dev", "jmhip_me_subpel", "jmhip_me_subpel_dev", "jmhip_me_refine_dev", "jmhip_tq_luma4x4",
           "jmhip_tq_luma4x4_dev", "jmhip_forward4x4", "jmhip_inverse4x4", "jmhip_forward8x8", "jmhip_inverse8x8",
           "jmhip_tq_luma8x8", "jmhip_tq_luma8x8_dev", "jmhip_tq_luma16x16", "jmhip_tq_luma16x16_dev", "jmhip_dc_transform", "jmhip_quant_dc4x4", "jmhip_tq_chroma",
           "jmhip_set_reference_chroma", "jmhip_set_reference_chroma_dev", "jmhip_get_chroma_subplanes", "jmhip_mc_luma", "jmhip_mc_luma_dev", "jmhip_mc_chroma", "jmhip_mc_chroma_dev", "jmhip_mc_luma_wp", "jmhip_mc_luma_wp_dev", "jmhip_mc_chroma_wp", "jmhip_mc_chroma_wp_dev", "jmhip_distortion", "jmhip_intrapred4x4", "jmhip_intrapred8x8", "jmhip_intra_chroma", "jmhip_intra_chroma_dev", "jmhip_intra16_search", "jmhip_intra16_search_dev", "jmhip_mc_mb16_dev", "jmhip_tq_rec_to_plane_dev", "jmhip_mb16_recon_luma_dev", "jmhip_mc_mb16_chroma_dev", "jmhip_tqc_rec_to_planes_dev", "jmhip_tq_chroma_dev",
           "jmhip_deblock_frame", "jmhip_deblock_frame_dev", "jmhip_enable_timing", "jmhip_last_kernel_ms",
           "jmhip_encode_slice", "jmhip_encode_slice_dev", "jmhip_encode_slice_begin", "jmhip_slice_record", "jmhip_encode_slice_end", "jmhip_recon_planes_dev", "jmhip_deblock_side_info_dev", "jmhip_get_recon", "jmhip_deblock_picture_dev", "jmhip_reference_from_recon", "jmhip_set_pipeline_workgroups",
           "jmhip_seq_open", "jmhip_seq_b_workgroups", "jmhip_seq_close", "jmhip_seq_set_frame", "jmhip_seq_set_planes", "jmhip_seq_set_frame_dev", "jmhip_seq_encode", "jmhip_seq_record", "jmhip_seq_wait", "jmhip_seq_records", "jmhip_seq_records_dev",
           "jmhip_seq_recon_dev", "jmhip_seq_get_recon", "jmhip_seq_kernel_ms", "jmhip_seq_batch", "jmhip_seq_batch_reserve", "jmhip_seq_batch_lag", "jmhip_allgather_bands"]


# return codes of include/jmhip.h (JmHipError.code)
OK, EINVAL, ENODEV, ENOMEM, EHIP, EUNSUPPORTED, EREACH = 0, -1, -2, -3, -4, -5, -6


class JmHipError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("yuv_format", C.c_int32),
                ("bit_depth", C.c_int32), ("search_range", C.c_int32), ("num_ref_slots", C.c_int32), ("stream", C.c_void_p)]


def library_path():
    return os.environ.get("JMHIP_LIB") or os.path.join(_HERE, "libjmhip.so")


def load_library():
    """Load libjmhip.so; raises if it has not been built (python -m jm_amd.build)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise JmHipError(f"{path} is missing: build it with `python -m jm_amd.build` (there is no CPU fallback)")
        # One HIP runtime per process: PyTorch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever copy is
        # loaded first serves both, and torch cannot see a device through a foreign copy ("No HIP GPUs are available"),
        # so when the harness uses torch at all its runtime has to be the one in the process.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        lib.jmhip_last_error.restype = C.c_char_p
        lib.jmhip_last_error.argtypes = [C.c_void_p]
        lib.jmhip_subplanes_dev.restype = C.c_void_p
        lib.jmhip_subplanes_dev.argtypes = [C.c_void_p, C.c_int32]
        lib.jmhip_destroy.restype = None
        lib.jmhip_destroy.argtypes = [C.c_void_p]
        _LIB = lib
    return _LIB


def _vp(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))          # a device pointer (int), e.g. torch.Tensor.data_ptr()


class JmHip:
    """One libjmhip context = one GPU, one picture size, one stream."""

    def __init__(self, width, height, search_range=32, num_ref_slots=1, yuv_format=1, device=0, stream=None):
        self.lib = load_library()
        self.cfg = _Config(device, width, height, yuv_format, 8, search_range, num_ref_slots, stream)
        self.h = C.c_void_p()
        rc = self.lib.jmhip_create(C.byref(self.h), C.byref(self.cfg))
        if rc != 0:
            raise JmHipError(f"jmhip_create failed ({rc}): {self.lib.jmhip_last_error(None).decode()}")
        self.W, self.H, self.R, self.yuv_format = width, height, search_range, yuv_format
        p, r, s = C.c_int32(), C.c_int32(), C.c_int64()
        self._ck(self.lib.jmhip_plane_geometry(self.h, C.byref(p), C.byref(r), C.byref(s)))
        self.pitch, self.rows, self.plane_stride = p.value, r.value, s.value

    def close(self):
        if self.h:
            self.lib.jmhip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            err = JmHipError(f"libjmhip error {rc}: {self.lib.jmhip_last_error(self.h).decode()}")
            err.code = rc                                                  # (JMHIP_EREACH = -6: an EPZS batch to be coded again picture by picture)
            raise err

    def synchronize(self):
        self._ck(self.lib.jmhip_synchronize(self.h))

    def enable_timing(self, on=True):
        self._ck(self.lib.jmhip_enable_timing(self.h, int(on)))

    def last_kernel_ms(self, kind):
        ms = C.c_float()
        self._ck(self.lib.jmhip_last_kernel_ms(self.h, kind, C.byref(ms)))
        return ms.value

    # ---- frames
    def set_pipeline_workgroups(self, n):
        """how many workgroups a slice's launch may occupy (0 = 256): a context's share of the chip when several sequences are encoded at once"""
        self._ck(self.lib.jmhip_set_pipeline_workgroups(self.h, C.c_int32(int(n))))

    def set_stream(self, hip_stream):
        """launch every later call on this HIP stream (an int handle, e.g. torch.cuda.Stream.cuda_stream); ordering is the caller's"""
        self._ck(self.lib.jmhip_set_stream(self.h, C.c_void_p(int(hip_stream))))

    def set_current(self, luma):
        """p_Vid->pCurImg := luma (H x W, any integer dtype, values 0..255)."""
        a = np.ascontiguousarray(luma, np.uint16)
        assert a.shape == (self.H, self.W)
        self._ck(self.lib.jmhip_set_current(self.h, _vp(a), self.W))

    def set_current_dev(self, dptr, pitch):
        self._ck(self.lib.jmhip_set_current_dev(self.h, _vp(dptr), pitch))

    def set_current_frame(self, raw, src_w, src_h):
        """read_one_frame + pad_borders: one frame's bytes as they lie in the YUV file (planar Y, U, V at src_w x src_h) -> the coded-size
        planes on the device; the luma plane becomes the current picture"""
        a = np.ascontiguousarray(np.frombuffer(raw, np.uint8) if not isinstance(raw, np.ndarray) else raw, np.uint8)
        cw, chh = src_w // 2, (src_h // 2 if self.yuv_format == 1 else src_h)
        assert a.size == src_w * src_h + (2 * cw * chh if self.yuv_format else 0), (a.size, src_w, src_h)
        self._ck(self.lib.jmhip_set_current_frame(self.h, _vp(a), src_w, src_h))

    def set_current_frame_dev(self, d_raw, src_w, src_h):
        self._ck(self.lib.jmhip_set_current_frame_dev(self.h, _vp(d_raw), src_w, src_h))

    def load_frame(self, raw, yuv, src_w, src_h, out_w, out_h, symbol_bytes, src_depth, out_depth):
        """jmhip_load_frame: the general reader; (y, u, v) uint16 planes of the coded size (u, v None at 4:0:0)"""
        raw = np.ascontiguousarray(np.frombuffer(raw, np.uint8) if not isinstance(raw, np.ndarray) else raw, np.uint8)
        sx, sy = (1 if yuv in (1, 2) else 0), (1 if yuv == 1 else 0)
        need = (src_w * src_h + (2 * (src_w >> sx) * (src_h >> sy) if yuv else 0)) * symbol_bytes
        if symbol_bytes in (1, 2) and raw.size < need:        # the library copies `need` bytes from this buffer (a format it refuses is refused there)
            raise ValueError(f"load_frame: {raw.size} bytes for a {src_w}x{src_h} frame of format {yuv} with {symbol_bytes}-byte samples ({need} needed)")
        W, H = (int(out_w) + 15) // 16 * 16, (int(out_h) + 15) // 16 * 16
        f = np.zeros(1, FRAME_FORMAT)
        f["yuv_format"], f["src_w"], f["src_h"], f["out_w"], f["out_h"], f["coded_w"], f["coded_h"], f["symbol_bytes"] = yuv, src_w, src_h, out_w, out_h, W, H, symbol_bytes
        f["src_depth"], f["out_depth"] = src_depth, out_depth
        sx, sy = (1 if yuv in (1, 2) else 0), (1 if yuv == 1 else 0)
        y = np.zeros((H, W), np.uint16)
        u = np.zeros((H >> sy, W >> sx), np.uint16) if yuv else None
        v = np.zeros_like(u) if yuv else None
        self._ck(self.lib.jmhip_load_frame(self.h, _vp(f), _vp(raw), _vp(y), _vp(u) if yuv else None, _vp(v) if yuv else None))
        return y, u, v

    def get_current_planes(self):
        """(y, u, v) uint8 planes of the coded size as jmhip_set_current_frame left them (u, v None at 4:0:0)"""
        cw, chh = self.W // 2, (self.H // 2 if self.yuv_format == 1 else self.H)
        y = np.zeros((self.H, self.W), np.uint16)
        u = np.zeros((chh, cw), np.uint16) if self.yuv_format else None
        v = np.zeros((chh, cw), np.uint16) if self.yuv_format else None
        self._ck(self.lib.jmhip_get_current_planes(self.h, _vp(y), _vp(u), _vp(v)))
        return y.astype(np.uint8), (u.astype(np.uint8) if u is not None else None), (v.astype(np.uint8) if v is not None else None)

    def current_planes_dev(self):
        """device pointers (y, pitch_y, u, v, pitch_c) of the current picture's planes"""
        py, pu, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        a, b = C.c_int32(), C.c_int32()
        self._ck(self.lib.jmhip_current_planes_dev(self.h, C.byref(py), C.byref(a), C.byref(pu), C.byref(pv), C.byref(b)))
        return py.value, a.value, pu.value, pv.value, b.value

    def set_reference(self, slot, luma):
        """getSubImagesLuma for reference `slot` (lencod/src/img_luma.c:611)."""
        a = np.ascontiguousarray(luma, np.uint16)
        assert a.shape == (self.H, self.W)
        self._ck(self.lib.jmhip_set_reference(self.h, slot, _vp(a), self.W))

    def set_reference_dev(self, slot, dptr, pitch):
        self._ck(self.lib.jmhip_set_reference_dev(self.h, slot, _vp(dptr), pitch))

    def get_subplanes(self, slot):
        out = np.zeros((16, self.H + 2 * PAD_Y, self.W + 2 * PAD_X), np.uint16)
        self._ck(self.lib.jmhip_get_subplanes(self.h, slot, _vp(out)))
        return out

    def subplanes_dev(self, slot):
        return self.lib.jmhip_subplanes_dev(self.h, slot)

    # ---- motion estimation
    def me_fullsearch(self, slot, jobs):
        jobs = np.ascontiguousarray(jobs, ME_JOB)
        res = np.zeros(len(jobs), ME_RESULT)
        self._ck(self.lib.jmhip_me_fullsearch(self.h, slot, _vp(jobs), len(jobs), _vp(res)))
        return res

    def me_fullsearch_dev(self, slot, d_jobs, n, d_results):
        self._ck(self.lib.jmhip_me_fullsearch_dev(self.h, slot, _vp(d_jobs), n, _vp(d_results)))

    def me_sad_tables(self, slot, jobs):
        jobs = np.ascontiguousarray(jobs, ME_JOB)
        R = int(jobs["search_range"][0])
        out = np.zeros((len(jobs), 7, 16, (2 * R + 1) ** 2), np.uint16)
        self._ck(self.lib.jmhip_me_sad_tables(self.h, slot, _vp(jobs), len(jobs), _vp(out)))
        return out

    def me_eval(self, slot, cands):
        cands = np.ascontiguousarray(cands, CAND)
        out = np.zeros(len(cands), np.int32)
        self._ck(self.lib.jmhip_me_eval(self.h, slot, _vp(cands), len(cands), _vp(out)))
        return out

    def me_eval_pred(self, cands):
        """weighted / bi-predictive candidate distortions (compute*WP, computeBiPred*1 / *2): PRED_CAND records -> distortion << 5"""
        cands = np.ascontiguousarray(cands, PRED_CAND)
        out = np.zeros(len(cands), np.int32)
        self._ck(self.lib.jmhip_me_eval_pred(self.h, _vp(cands), len(cands), _vp(out)))
        return out

    def me_eval_pred_dev(self, d_cands, n, d_out):
        self._ck(self.lib.jmhip_me_eval_pred_dev(self.h, C.c_void_p(d_cands), n, C.c_void_p(d_out)))

    def me_subpel(self, slot, jobs):
        jobs = np.ascontiguousarray(jobs, SUBPEL_JOB)
        out = np.zeros(len(jobs), ME_BEST)
        self._ck(self.lib.jmhip_me_subpel(self.h, slot, _vp(jobs), len(jobs), _vp(out)))
        return out

    def me_subpel_dev(self, slot, d_jobs, n, d_out):
        self._ck(self.lib.jmhip_me_subpel_dev(self.h, slot, _vp(d_jobs), n, _vp(d_out)))

    def me_refine_dev(self, slot, d_jobs, n, d_int, prm, d_out):
        self._ck(self.lib.jmhip_me_refine_dev(self.h, slot, _vp(d_jobs), n, _vp(d_int), _vp(prm), _vp(d_out)))

    @staticmethod
    def refine_params(lambda_h, lambda_q, metric_h=2, metric_q=2, start_hp=0, start_qp=0, transform8x8_mode=0):
        p = np.zeros(1, REFINE_PARAMS)
        p["lambda_h"], p["lambda_q"], p["metric_h"], p["metric_q"] = lambda_h, lambda_q, metric_h, metric_q
        p["start_hp"], p["start_qp"], p["transform8x8_mode"] = start_hp, start_qp, transform8x8_mode
        return p

    # ---- transform / quant
    @staticmethod
    def tq_params(q, qp_per, cavlc=1, adaptive_rounding=0, adapt_rnd_weight=0, max_pel=255):
        p = np.zeros(1, TQ_PARAMS)
        p["q"][0] = np.asarray(q, np.int32).reshape(16, 3)
        p["qp_per"], p["cavlc"], p["adaptive_rounding"] = qp_per, cavlc, adaptive_rounding
        p["adapt_rnd_weight"], p["max_pel"] = adapt_rnd_weight, max_pel
        return p

    def tq_luma4x4(self, prm, orig, pred):
        orig = np.ascontiguousarray(orig, np.uint8).reshape(-1, 16)
        pred = np.ascontiguousarray(pred, np.uint8).reshape(-1, 16)
        out = np.zeros(len(orig), TQ_OUT)
        self._ck(self.lib.jmhip_tq_luma4x4(self.h, _vp(prm), _vp(orig), _vp(pred), len(orig), _vp(out)))
        return out

    def tq_luma4x4_dev(self, prm, d_orig, d_pred, n, d_out):
        self._ck(self.lib.jmhip_tq_luma4x4_dev(self.h, _vp(prm), _vp(d_orig), _vp(d_pred), n, _vp(d_out)))

    def _xform(self, fn, x, sz):
        x = np.ascontiguousarray(x, np.int32).reshape(-1, sz * sz)
        out = np.zeros_like(x)
        self._ck(fn(self.h, _vp(x), len(x), _vp(out)))
        return out

    def forward4x4(self, x):
        return self._xform(self.lib.jmhip_forward4x4, x, 4)

    def inverse4x4(self, x):
        return self._xform(self.lib.jmhip_inverse4x4, x, 4)

    def forward8x8(self, x):
        return self._xform(self.lib.jmhip_forward8x8, x, 8)

    def inverse8x8(self, x):
        return self._xform(self.lib.jmhip_inverse8x8, x, 8)

    @staticmethod
    def tq8_params(q, qp_per, cavlc=0, adaptive_rounding=0, adapt_rnd_weight=0, max_pel=255):
        p = np.zeros(1, TQ8_PARAMS)
        p["q"][0] = np.asarray(q, np.int32).reshape(64, 3)
        p["qp_per"], p["cavlc"], p["adaptive_rounding"] = qp_per, cavlc, adaptive_rounding
        p["adapt_rnd_weight"], p["max_pel"] = adapt_rnd_weight, max_pel
        return p

    def tq_luma8x8(self, prm, orig, pred):
        """residual_transform_quant_luma_8x8 (lencod/src/transform8x8.c:522 / :604 with prm.cavlc), batched."""
        orig = np.ascontiguousarray(orig, np.uint8).reshape(-1, 64)
        pred = np.ascontiguousarray(pred, np.uint8).reshape(-1, 64)
        out = np.zeros(len(orig), TQ8_OUT)
        self._ck(self.lib.jmhip_tq_luma8x8(self.h, _vp(prm), _vp(orig), _vp(pred), len(orig), _vp(out)))
        return out

    def tq_luma8x8_dev(self, prm, d_orig, d_pred, n, d_out):
        self._ck(self.lib.jmhip_tq_luma8x8_dev(self.h, _vp(prm), _vp(d_orig), _vp(d_pred), n, _vp(d_out)))

    def tq_luma16x16(self, prm, orig, pred):
        """residual_transform_quant_luma_16x16 (lencod/src/block.c:208) of whole macroblocks; prm from tq_params() with the intra
        quantiser; orig / pred: (n, 256) uint8"""
        o = np.ascontiguousarray(orig, np.uint8).reshape(-1, 256); p = np.ascontiguousarray(pred, np.uint8).reshape(-1, 256)
        out = np.zeros(len(o), TQ16_OUT)
        self._ck(self.lib.jmhip_tq_luma16x16(self.h, _vp(prm), _vp(o), _vp(p), len(o), _vp(out)))
        return out

    def tq_luma16x16_dev(self, prm, d_orig, d_pred, n, d_out):
        self._ck(self.lib.jmhip_tq_luma16x16_dev(self.h, _vp(prm), _vp(d_orig), _vp(d_pred), n, _vp(d_out)))

    def tq_chroma(self, yuv, q_ac, q_dc, qp_per_ac, qp_per_dc, cavlc, adaptive_rounding, adapt_rnd_weight, mbs, orig, pred, max_pel=255):
        """residual_transform_quant_chroma_4x4 (lencod/src/block.c:954), batched over (macroblock, plane) items.
        mbs: TQC_MB array (updated copy returned); orig / pred: (n, 128) uint8, rows of 8 samples."""
        p = np.zeros(1, TQC_PARAMS)
        p["q_ac"][0] = np.asarray(q_ac, np.int32).reshape(16, 3); p["q_dc"][0] = np.asarray(q_dc, np.int32).reshape(3)
        p["qp_per_ac"], p["qp_per_dc"], p["yuv_format"], p["cavlc"] = qp_per_ac, qp_per_dc, yuv, cavlc
        p["adaptive_rounding"], p["adapt_rnd_weight"], p["max_pel"] = adaptive_rounding, adapt_rnd_weight, max_pel
        mbs = np.ascontiguousarray(mbs, TQC_MB).copy()
        orig = np.ascontiguousarray(orig, np.uint8).reshape(-1, 128); pred = np.ascontiguousarray(pred, np.uint8).reshape(-1, 128)
        out = np.zeros(len(mbs), TQC_OUT)
        self._ck(self.lib.jmhip_tq_chroma(self.h, _vp(p), _vp(mbs), _vp(orig), _vp(pred), len(mbs), _vp(out)))
        return mbs, out

    def dc_transform(self, name, x):
        """hadamard4x4 / ihadamard4x4 / hadamard4x2 / ihadamard4x2 / hadamard2x2 / ihadamard2x2 (lcommon/src/transform.c:121-330)."""
        kind, per = DC_KINDS[name]
        x = np.ascontiguousarray(x, np.int32).reshape(-1, per)
        out = np.zeros_like(x)
        self._ck(self.lib.jmhip_dc_transform(self.h, kind, _vp(x), len(x), _vp(out)))
        return out

    def quant_dc4x4(self, qparam, qp_per, cavlc, blocks):
        """quant_dc4x4_normal (lencod/src/quant4x4_normal.c:200); returns (levels left in the block, DC_OUT records)."""
        q = np.ascontiguousarray(qparam, np.int32).reshape(3)
        b = np.ascontiguousarray(blocks, np.int32).reshape(-1, 16).copy()
        out = np.zeros(len(b), DC_OUT)
        self._ck(self.lib.jmhip_quant_dc4x4(self.h, _vp(q), qp_per, cavlc, _vp(b), len(b), _vp(out)))
        return b, out

    # ---- motion-compensated prediction
    def set_reference_chroma(self, slot, u, v):
        """the integer chroma planes of the reference picture in `slot` (StorablePicture.imgUV)"""
        U = np.ascontiguousarray(u, np.uint16); V = np.ascontiguousarray(v, np.uint16)
        assert U.shape == V.shape
        self._ck(self.lib.jmhip_set_reference_chroma(self.h, slot, _vp(U), _vp(V), U.shape[1]))

    def set_reference_chroma_dev(self, slot, d_u, d_v, pitch):
        self._ck(self.lib.jmhip_set_reference_chroma_dev(self.h, slot, _vp(d_u), _vp(d_v), pitch))

    def get_chroma_subplanes(self, slot, plane):
        """getSubImagesChroma (lencod/src/img_chroma.c:338) of plane 0 = U / 1 = V: (ny, 8, ch + 2 pad_y, cw + 2 pad_x) uint16"""
        fmt = self.yuv_format
        ny, pad_y = (4, 20) if fmt == 2 else (8, 10)
        ch = self.H if fmt == 2 else self.H // 2
        out = np.zeros((ny, 8, ch + 2 * pad_y, self.W // 2 + 32), np.uint16)
        self._ck(self.lib.jmhip_get_chroma_subplanes(self.h, slot, plane, _vp(out)))
        return out

    def intrapred8x8(self, blks):
        """get_intrapred_8x8: IP8_BLK records (25 filtered predictor samples, mode, left, up) -> (n, 64) uint8"""
        b = np.ascontiguousarray(blks, IP8_BLK)
        out = np.zeros((len(b), 64), np.uint8)
        self._ck(self.lib.jmhip_intrapred8x8(self.h, _vp(b), len(b), _vp(out)))
        return out

    def intra_chroma(self, mbs):
        """intra_chroma_prediction: IC_MB records -> (n, 4 modes, 2 planes, 16, 8) uint8 (rows >= 8 unused at 4:2:0)"""
        m = np.ascontiguousarray(mbs, IC_MB)
        out = np.zeros((len(m), 4, 2, 16, 8), np.uint8)
        self._ck(self.lib.jmhip_intra_chroma(self.h, _vp(m), len(m), _vp(out)))
        return out

    def intra_chroma_dev(self, d_mbs, n, d_out):
        self._ck(self.lib.jmhip_intra_chroma_dev(self.h, _vp(d_mbs), n, _vp(d_out)))

    def mc_luma(self, blocks):
        """luma_prediction (lencod/src/mc_prediction.c:144), un-weighted; blocks: MC_LUMA_BLK array -> (n, 256) uint8, w*h samples first"""
        b = np.ascontiguousarray(blocks, MC_LUMA_BLK)
        out = np.zeros((len(b), 256), np.uint8)
        self._ck(self.lib.jmhip_mc_luma(self.h, _vp(b), len(b), _vp(out)))
        return out

    def mc_luma_dev(self, d_blocks, n, d_out):
        self._ck(self.lib.jmhip_mc_luma_dev(self.h, _vp(d_blocks), n, _vp(d_out)))

    def mc_luma_wp(self, blocks, weights):
        """luma_prediction with weighted prediction (weighted_mc_prediction / weighted_bi_prediction): weights[i] (MC_WEIGHTS) for blocks[i]"""
        b, w = np.ascontiguousarray(blocks, MC_LUMA_BLK), np.ascontiguousarray(weights, MC_WEIGHTS)
        assert len(b) == len(w)
        out = np.zeros((len(b), 256), np.uint8)
        self._ck(self.lib.jmhip_mc_luma_wp(self.h, _vp(b), _vp(w), len(b), _vp(out)))
        return out

    def mc_luma_wp_dev(self, d_blocks, d_weights, n, d_out):
        self._ck(self.lib.jmhip_mc_luma_wp_dev(self.h, _vp(d_blocks), _vp(d_weights), n, _vp(d_out)))

    def mc_chroma_wp(self, blocks, weights):
        b, w = np.ascontiguousarray(blocks, MC_CHROMA_BLK), np.ascontiguousarray(weights, MC_WEIGHTS)
        assert len(b) == len(w)
        out = np.zeros((len(b), 16), np.uint8)
        self._ck(self.lib.jmhip_mc_chroma_wp(self.h, _vp(b), _vp(w), len(b), _vp(out)))
        return out

    def mc_chroma_wp_dev(self, d_blocks, d_weights, n, d_out):
        self._ck(self.lib.jmhip_mc_chroma_wp_dev(self.h, _vp(d_blocks), _vp(d_weights), n, _vp(d_out)))

    def mc_chroma(self, blocks):
        """chroma_prediction_4x4 (lencod/src/mc_prediction.c:568, ChromaMCBuffer = 1), un-weighted; MC_CHROMA_BLK array -> (n, 16) uint8"""
        b = np.ascontiguousarray(blocks, MC_CHROMA_BLK)
        out = np.zeros((len(b), 16), np.uint8)
        self._ck(self.lib.jmhip_mc_chroma(self.h, _vp(b), len(b), _vp(out)))
        return out

    def mc_chroma_dev(self, d_blocks, n, d_out):
        self._ck(self.lib.jmhip_mc_chroma_dev(self.h, _vp(d_blocks), n, _vp(d_out)))

    def mc_mb16_dev(self, slot, d_jobs, d_results, n, y_offset, blocks_per_row, d_pred):
        self._ck(self.lib.jmhip_mc_mb16_dev(self.h, slot, _vp(d_jobs), _vp(d_results), n, y_offset, blocks_per_row, _vp(d_pred)))

    def tq_rec_to_plane_dev(self, d_out, n, blocks_per_row, d_plane, pitch):
        self._ck(self.lib.jmhip_tq_rec_to_plane_dev(self.h, _vp(d_out), n, blocks_per_row, _vp(d_plane), pitch))

    def mb16_recon_luma_dev(self, slot, prm, d_jobs, d_results, n, y_offset, blocks_per_row, d_orig, d_out, d_pred, d_plane, pitch):
        """mc_mb16_dev + tq_luma4x4_dev + tq_rec_to_plane_dev in one launch (d_pred may be 0)"""
        self._ck(self.lib.jmhip_mb16_recon_luma_dev(self.h, slot, _vp(prm), _vp(d_jobs), _vp(d_results), n, y_offset, blocks_per_row, _vp(d_orig), _vp(d_out),
                                                    _vp(d_pred), _vp(d_plane), pitch))

    def mc_mb16_chroma_dev(self, slot, d_jobs, d_results, n, d_pred):
        self._ck(self.lib.jmhip_mc_mb16_chroma_dev(self.h, slot, _vp(d_jobs), _vp(d_results), n, _vp(d_pred)))

    def tqc_rec_to_planes_dev(self, d_jobs, d_out, n, y_offset, d_u, d_v, pitch):
        self._ck(self.lib.jmhip_tqc_rec_to_planes_dev(self.h, _vp(d_jobs), _vp(d_out), n, y_offset, _vp(d_u), _vp(d_v), pitch))

    @staticmethod
    def tqc_params(yuv, q_ac, q_dc, qp_per_ac, qp_per_dc, cavlc=1, adaptive_rounding=0, adapt_rnd_weight=0, max_pel=255):
        p = np.zeros(1, TQC_PARAMS)
        p["q_ac"][0] = np.asarray(q_ac, np.int32).reshape(16, 3); p["q_dc"][0] = np.asarray(q_dc, np.int32).reshape(3)
        p["qp_per_ac"], p["qp_per_dc"], p["yuv_format"], p["cavlc"] = qp_per_ac, qp_per_dc, yuv, cavlc
        p["adaptive_rounding"], p["adapt_rnd_weight"], p["max_pel"] = adaptive_rounding, adapt_rnd_weight, max_pel
        return p

    def tq_chroma_dev(self, prm, d_mbs, d_orig, d_pred, n, d_out):
        self._ck(self.lib.jmhip_tq_chroma_dev(self.h, _vp(prm), _vp(d_mbs), _vp(d_orig), _vp(d_pred), n, _vp(d_out)))

    def intrapred4x4(self, blocks):
        """get_intrapred_4x4 (lencod/src/intra4x4.c:521): IP4_BLK array -> (n, 16) uint8"""
        b = np.ascontiguousarray(blocks, IP4_BLK)
        out = np.zeros((len(b), 16), np.uint8)
        self._ck(self.lib.jmhip_intrapred4x4(self.h, _vp(b), len(b), _vp(out)))
        return out

    def intra16_search(self, mbs, orig):
        """find_sad_16x16_JM (lencod/src/intra16x16.c:463): I16_MB array, orig (n, 256) uint8 -> I16_OUT array"""
        m = np.ascontiguousarray(mbs, I16_MB); o = np.ascontiguousarray(orig, np.uint8).reshape(-1, 256)
        out = np.zeros(len(m), I16_OUT)
        self._ck(self.lib.jmhip_intra16_search(self.h, _vp(m), _vp(o), len(m), _vp(out)))
        return out

    def distortion(self, metric, size, diff):
        """distortion4x4 / distortion8x8 {SAD 0, SSE 1, SATD 2} (lencod/src/me_distortion.c:38-146) of int16 difference blocks -> int64 << 5"""
        d = np.ascontiguousarray(diff, np.int16).reshape(-1, size * size)
        out = np.zeros(len(d), np.int64)
        self._ck(self.lib.jmhip_distortion(self.h, metric, size, _vp(d), len(d), _vp(out)))
        return out

    # ---- deblocking
    def deblock_frame(self, y, u, v, mbs, motion, direct8x8=1):
        """DeblockFrame (lencod/src/loopFilter.c:63); returns the filtered planes."""
        Y = np.ascontiguousarray(y, np.uint16).copy()
        U = np.ascontiguousarray(u, np.uint16).copy() if u is not None else None
        V = np.ascontiguousarray(v, np.uint16).copy() if v is not None else None
        mbs = np.ascontiguousarray(mbs, DB_MB)
        motion = np.ascontiguousarray(motion, DB_MOTION)
        self._ck(self.lib.jmhip_deblock_frame(self.h, _vp(Y), Y.shape[1], _vp(U), _vp(V), U.shape[1] if U is not None else 0,
                                              _vp(mbs), _vp(motion), direct8x8))
        return Y, U, V

    def deblock_frame_dev(self, dY, pitchY, dU, dV, pitchC, d_mbs, d_motion, direct8x8=1):
        self._ck(self.lib.jmhip_deblock_frame_dev(self.h, _vp(dY), pitchY, _vp(dU), _vp(dV), pitchC, _vp(d_mbs), _vp(d_motion), direct8x8))


    # ---- the RDO-off macroblock pipeline of a slice (encode_one_macroblock_low, lencod/src/md_low.c:104)
    def encode_slice(self, prm):
        """prm: one SLICE_PARAMS record -> MB_RECORD per macroblock of the slice; reconstruction and loop-filter side information stay on the device"""
        prm = np.ascontiguousarray(prm, SLICE_PARAMS).reshape(1)
        out = np.zeros(self.slice_call_macroblocks(prm), MB_RECORD)
        self._ck(self.lib.jmhip_encode_slice(self.h, _vp(prm), _vp(out)))
        return out

    def slice_call_macroblocks(self, prm):
        """macroblocks one jmhip_encode_slice call covers (num_slices slices of num_mb, the last one ends with the picture)"""
        first, num, ns = int(prm["first_mb"][0]), int(prm["num_mb"][0]), max(1, int(prm["num_slices"][0]))
        return min(num * ns, (self.W // 16) * (self.H // 16) - first)

    def encode_slice_dev(self, prm, d_out=None):
        prm = np.ascontiguousarray(prm, SLICE_PARAMS).reshape(1)
        self._ck(self.lib.jmhip_encode_slice_dev(self.h, _vp(prm), _vp(d_out)))

    def encode_slice_streamed(self, prm):
        """jmhip_encode_slice_begin / _record (raster order, while the device is still encoding) / _end; returns the records"""
        prm = np.ascontiguousarray(prm, SLICE_PARAMS).reshape(1)
        first, num = int(prm["first_mb"][0]), self.slice_call_macroblocks(prm)
        out = np.zeros(num, MB_RECORD)
        self._ck(self.lib.jmhip_encode_slice_begin(self.h, _vp(prm)))
        p = C.c_void_p()
        for k in range(num):
            self._ck(self.lib.jmhip_slice_record(self.h, first + k, C.byref(p)))
            out[k] = np.frombuffer((C.c_char * MB_RECORD.itemsize).from_address(p.value), MB_RECORD)[0]
        self._ck(self.lib.jmhip_encode_slice_end(self.h))
        return out

    def recon_planes_dev(self):
        py, pu, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        a, b = C.c_int32(), C.c_int32()
        self._ck(self.lib.jmhip_recon_planes_dev(self.h, C.byref(py), C.byref(a), C.byref(pu), C.byref(pv), C.byref(b)))
        return py.value, a.value, pu.value, pv.value, b.value

    def deblock_side_info_dev(self):
        """(pointer, bytes) of the per-macroblock and of the per-4x4-block loop filter side information of the current picture on the device"""
        pm, po = C.c_void_p(), C.c_void_p()
        self._ck(self.lib.jmhip_deblock_side_info_dev(self.h, C.byref(pm), C.byref(po)))
        nmb = (self.W // 16) * (self.H // 16)
        return pm.value, nmb * 28, po.value, nmb * 16 * 16

    def get_recon(self):
        """(y, u, v) uint8: the reconstruction on the device (before or after deblock_picture_dev)"""
        y = np.zeros((self.H, self.W), np.uint16)
        ch = self.H if self.yuv_format == 2 else self.H // 2
        u = np.zeros((ch, self.W // 2), np.uint16)
        v = np.zeros((ch, self.W // 2), np.uint16)
        self._ck(self.lib.jmhip_get_recon(self.h, _vp(y), self.W, _vp(u), _vp(v), self.W // 2))
        return y.astype(np.uint8), u.astype(np.uint8), v.astype(np.uint8)

    def deblock_picture_dev(self, direct8x8=1):
        self._ck(self.lib.jmhip_deblock_picture_dev(self.h, direct8x8))

    def reference_from_recon(self, slot):
        self._ck(self.lib.jmhip_reference_from_recon(self.h, slot))

    # ---- consecutive pictures of one sequence in flight side by side (include/jmhip.h: jmhip_seq_*)
    def seq_open(self, depth, workgroups=0, ready=False):
        """jmhip_seq_open.  The library makes the entries beyond the first beside the first picture's coding (a thread, joined by the first call that names such an entry);
        ready=True waits for them here -- what a measurement wants, so that no stream is being made inside its clock"""
        self._ck(self.lib.jmhip_seq_open(self.h, int(depth), int(workgroups)))
        if ready and depth > 1:
            self.seq_wait(depth - 1)

    def seq_close(self):
        self._ck(self.lib.jmhip_seq_close(self.h))

    def seq_set_frame(self, entry, raw, src_w, src_h):
        a = np.ascontiguousarray(np.frombuffer(raw, np.uint8) if not isinstance(raw, np.ndarray) else raw, np.uint8)
        cw, chh = src_w // 2, (src_h // 2 if self.yuv_format == 1 else src_h)
        assert a.size == src_w * src_h + (2 * cw * chh if self.yuv_format else 0), (a.size, src_w, src_h)
        self._ck(self.lib.jmhip_seq_set_frame(self.h, int(entry), _vp(a), src_w, src_h))

    def seq_set_frame_dev(self, entry, d_raw, src_w, src_h):
        self._ck(self.lib.jmhip_seq_set_frame_dev(self.h, int(entry), _vp(d_raw), src_w, src_h))

    def seq_b_workgroups(self, workgroups):
        self._ck(self.lib.jmhip_seq_b_workgroups(self.h, int(workgroups)))

    def seq_encode(self, entry, prm, out_slot, direct8x8=1, to_host=False, d_out=None):
        prm = np.ascontiguousarray(prm, SLICE_PARAMS).reshape(1)
        self._ck(self.lib.jmhip_seq_encode(self.h, int(entry), _vp(prm), int(out_slot), int(direct8x8), int(bool(to_host)), _vp(d_out)))

    def seq_kernel_ms(self, entry):
        ms = C.c_float()
        self._ck(self.lib.jmhip_seq_kernel_ms(self.h, int(entry), C.byref(ms)))
        return ms.value

    def seq_wait(self, entry):
        self._ck(self.lib.jmhip_seq_wait(self.h, int(entry)))

    def seq_records(self, entry, num_mb=None):
        out = np.zeros((self.W // 16) * (self.H // 16) if num_mb is None else num_mb, MB_RECORD)
        self._ck(self.lib.jmhip_seq_records(self.h, int(entry), _vp(out)))
        return out

    def seq_records_streamed(self, entry, first, num):
        """the records of a picture launched with to_host, read in raster order while the device is still encoding"""
        out = np.zeros(num, MB_RECORD)
        p = C.c_void_p()
        for k in range(num):
            self._ck(self.lib.jmhip_seq_record(self.h, int(entry), first + k, C.byref(p)))
            out[k] = np.frombuffer((C.c_char * MB_RECORD.itemsize).from_address(p.value), MB_RECORD)[0]
        return out

    def seq_batch(self, prm, pictures, direct8x8=1):
        """jmhip_seq_batch: consecutive P pictures in one launch.  pictures: dicts with d_raw, src_w, src_h, out_slot, ref_slot (list), ref_id (list), d_records
        (device pointers as ints); asynchronous -- synchronize() reports the launch's errors"""
        prm = np.ascontiguousarray(prm, SLICE_PARAMS).reshape(1)
        a = np.zeros(len(pictures), SEQ_PICTURE)
        for k, q in enumerate(pictures):
            a[k]["d_raw"], a[k]["src_w"], a[k]["src_h"], a[k]["out_slot"], a[k]["d_records"] = int(q["d_raw"]), q["src_w"], q["src_h"], q["out_slot"], int(q["d_records"])
            a[k]["ref_slot"][:len(q["ref_slot"])] = q["ref_slot"]
            a[k]["ref_id"][:len(q["ref_id"])] = q["ref_id"]
            a[k]["poc_offset"] = q.get("poc_offset", 0)                    # EPZS: the picture's order counts relative to prm's
        self._ck(self.lib.jmhip_seq_batch(self.h, _vp(prm), int(direct8x8), len(pictures), _vp(a)))

    def seq_batch_lag(self, lag):
        """jmhip_seq_batch_lag: the queue lag of EPZS launches of several pictures (0: the library's)"""
        self._ck(self.lib.jmhip_seq_batch_lag(self.h, int(lag)))

    def seq_batch_reserve(self, n):
        """device memory for launches of up to n pictures (jmhip_seq_batch), ahead of the first one"""
        self._ck(self.lib.jmhip_seq_batch_reserve(self.h, int(n)))

    def seq_get_recon(self, slot):
        """(y, u, v) uint8: the filtered reconstruction a sequence launch left in `slot`"""
        y = np.zeros((self.H, self.W), np.uint16)
        ch = self.H if self.yuv_format == 2 else self.H // 2
        u = np.zeros((ch, self.W // 2), np.uint16)
        v = np.zeros((ch, self.W // 2), np.uint16)
        self._ck(self.lib.jmhip_seq_get_recon(self.h, int(slot), _vp(y), self.W, _vp(u), _vp(v), self.W // 2))
        return y.astype(np.uint8), u.astype(np.uint8), v.astype(np.uint8)


def allgather_bands(contexts, band_mb_rows):
    """jmhip_allgather_bands: every context of the list (one per device, or several on one) gets every band of the picture whose slices they coded"""
    lib = load_library()
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    rc = lib.jmhip_allgather_bands(arr, len(contexts), int(band_mb_rows))
    if rc != 0:
        raise JmHipError(f"libjmhip error {rc}: {lib.jmhip_last_error(contexts[0].h).decode()}")


def db_arrays_from_tap(mbs12, mot):
    """Convert the (N,12) macroblock rows and (H/4,W/4,2,3) motion array of the golden fixtures
    (order of oracle/ref_tap.c) into DB_MB / DB_MOTION arrays."""
    n = len(mbs12)
    m = np.zeros(n, DB_MB)
    m["mb_type"], m["slice_type"], m["qp"] = mbs12[:, 0], mbs12[:, 1], mbs12[:, 2]
    m["qpc"][:, 0], m["qpc"][:, 1] = mbs12[:, 3], mbs12[:, 4]
    m["cbp"], m["cbp_blk"], m["slice_nr"] = mbs12[:, 5], mbs12[:, 6] & 0xFFFF, mbs12[:, 7]
    m["df_disable_idc"], m["df_alpha_c0"], m["df_beta"], m["transform8x8"] = mbs12[:, 8], mbs12[:, 9], mbs12[:, 10], mbs12[:, 11]
    h4, w4 = mot.shape[:2]
    mo = np.zeros((h4, w4), DB_MOTION)
    mo["mv"] = mot[:, :, :, 0:2]
    mo["ref_id"] = mot[:, :, :, 2]
    return m, mo.reshape(-1)
