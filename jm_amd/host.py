"""Host-side mirror of the reference's ME driver logic that sits directly above the kernels.

These helpers restate, for the product path (no oracle involved), the few scalar steps
BlockMotionSearch performs before it calls IntPelME / SubPelME, so a caller can turn a list of
JM-style block searches into libjmhip jobs:

  search_center()   lencod/src/mv_search.c:924-957 (round predictor to full-pel, clip_mv_range)
  level_mv_limits() lencod/src/conformance.c:37-67,604-646 (MaxHmvR/MaxVmvR for a LevelIDC)
  group_fs_jobs()   one jmhip_me_job per (macroblock, distinct search centre); this is how
                    full_search_motion_estimation's per-block windows map onto window jobs
  (the Lagrangian factors are never recomputed here: they are JM's -ffloat-store double arithmetic and reach the library as inputs)
"""
import math
import numpy as np
from .lib import ME_JOB, SUBPEL_JOB, PARTITIONS, NPART

_LEVELS = [10, 9, 11, 12, 13, 20, 21, 22, 30, 31, 32, 40, 41, 42, 50, 51, 52, 60, 61, 62]
_VLIM_Q = [255, 255, 511, 511, 511, 511, 1023, 1023, 1023] + [2047] * 8 + [32767] * 3


def level_mv_limits(level_idc):
    """(min_x, max_x, min_y, max_y) quarter-pel MV limits (MaxHmvR[4..5], MaxVmvR[4..5])."""
    idx = _LEVELS.index(level_idc)
    v = _VLIM_Q[idx]
    h = 8191 if idx < 17 else 32767
    return (-h - 1, h, -v - 1, v)


def search_center(pred, limits):
    """mv_search.c:931-957 with RDOptimization on and DisableMEPrediction off."""
    cx = ((int(pred[0]) + 2) >> 2) * 4
    cy = ((int(pred[1]) + 2) >> 2) * 4
    cx = min(max(cx, limits[0]), limits[1])
    cy = min(max(cy, limits[2]), limits[3])
    return cx, cy


def group_fs_jobs(mb_x, mb_y, preds, limits, search_range, lam):
    """preds: (41,2) predictors of the macroblock's partitions (ABI order).
    Returns (jobs, owner) -- `owner[p]` is the index into `jobs` of the job that searches partition p."""
    centers = {}
    owner = np.zeros(NPART, np.int32)
    jobs = []
    for p in range(NPART):
        c = search_center(preds[p], limits)
        if c not in centers:
            j = np.zeros(1, ME_JOB)
            j["mb_x"], j["mb_y"], j["center_x"], j["center_y"] = mb_x, mb_y, c[0], c[1]
            j["search_range"], j["lambda"], j["max_mvd"] = search_range, lam, 0
            j["pred"][0] = preds
            centers[c] = len(jobs)
            jobs.append(j)
        k = centers[c]
        jobs[k]["part_mask"] |= np.uint64(1 << p)
        owner[p] = k
    return np.concatenate(jobs), owner


def subpel_jobs_for(mb_x, mb_y, preds, int_mvs, lam_h, lam_q, metric_h=2, metric_q=2, start_hp=0, start_qp=0, test8x8=0):
    """SUBPEL_JOB array for the 41 partitions of one macroblock."""
    j = np.zeros(NPART, SUBPEL_JOB)
    for p, (_, bx, by, w, h) in enumerate(PARTITIONS):
        j[p]["pos_x"], j[p]["pos_y"], j[p]["bsx"], j[p]["bsy"] = mb_x + bx, mb_y + by, w, h
        j[p]["pred_x"], j[p]["pred_y"] = preds[p]
        j[p]["mv_x"], j[p]["mv_y"] = int_mvs[p]
    j["lambda_h"], j["lambda_q"], j["metric_h"], j["metric_q"] = lam_h, lam_q, metric_h, metric_q
    j["start_hp"], j["start_qp"], j["test8x8"], j["min_mcost"] = start_hp, start_qp, test8x8, 0x7fffffff
    return j
