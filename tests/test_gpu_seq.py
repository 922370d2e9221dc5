"""gpu: consecutive pictures of one sequence in flight side by side (jmhip_seq_*, jm_amd/csrc/mbpipe_post.inc) through the C ABI.

What is checked: a sequence coded with several pictures in flight -- the loop filter and the quarter-pel interpolation of a macroblock following its coding inside
the launch, the next picture's macroblocks starting as soon as THEIR part of the reference is there -- leaves exactly what the picture-after-picture path
(jmhip_encode_slice -> jmhip_deblock_picture_dev -> jmhip_reference_from_recon) leaves: every macroblock record, every filtered picture, all sixteen sub-pel planes
with their padding.  That path is pinned to the real encoder and to the oracle by tests/test_gpu_mbenc.py; the 1080p case here is compared with the real
encoder's records directly (tests/golden/mb_low_g2r.npz) and with the oracle's loop filter and interpolation."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, G)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mb_tap  # noqa: E402
import mbenc_util  # noqa: E402
from oracle import pyjmo  # noqa: E402
from test_gpu_mbenc import DevSeqEncoder, LAMBDAS, slice_params, synthetic_clip, hard_clip, as_oracle_records, first_difference  # noqa: E402

pytestmark = pytest.mark.gpu


class FlightEncoder:
    """IPPP with `depth` pictures in flight: picture k goes to entry k % depth and to slot k % (num_ref + depth + 1)."""

    def __init__(self, W, H, qp, R, num_ref, lambdas, depth, workgroups=0, cabac=0, search_mode=-1, transform8x8=0, yuv_format=1, stream_records=False, epzs=None, slice_mbs=0, disable_idc=0):
        import jm_amd.lib as L
        self.L = L
        self.slice_mbs, self.disable_idc = slice_mbs, disable_idc                      # slice_mbs > 0: the picture's slices of that many macroblocks in the one launch (num_slices)
        self.W, self.H, self.qp, self.R, self.num_ref, self.lambdas, self.depth = W, H, qp, R, num_ref, lambdas, depth
        self.cabac, self.search_mode, self.transform8x8, self.yuv_format, self.epzs = cabac, search_mode, transform8x8, yuv_format, dict(epzs or {})
        self.nslots = num_ref + depth + 1
        self.J = L.JmHip(W, H, search_range=max(R, 1), num_ref_slots=self.nslots, yuv_format=yuv_format)
        self.J.seq_open(depth, workgroups)
        self.stream_records = stream_records
        self.npic = 0
        self.results = {}

    def collect(self, k):
        J = self.J
        nmb = (self.W // 16) * (self.H // 16)
        if self.stream_records:
            recs = J.seq_records_streamed(k % self.depth, 0, nmb)
            J.seq_wait(k % self.depth)
        else:
            J.seq_wait(k % self.depth)
            recs = J.seq_records(k % self.depth)
        self.results[k] = (recs, J.seq_get_recon(k % self.nslots), J.get_subplanes(k % self.nslots))

    def submit(self, raw, sw, sh):
        k, L, J = self.npic, self.L, self.J
        if k >= self.depth:
            self.collect(k - self.depth)                     # the entry's previous picture, before its buffers are reused
        nmb = (self.W // 16) * (self.H // 16)
        st = 2 if k == 0 else 0
        nref = min(self.num_ref, k) if st == 0 else 0
        lam_mf, lam_md = self.lambdas[st]
        per = self.slice_mbs if 0 < self.slice_mbs < nmb else nmb
        cfg = pyjmo.mbenc_cfg(self.W, self.H, st, 0, per, self.qp, self.R, nref, lam_mf, lam_md, cabac=self.cabac, search_mode=self.search_mode, transform8x8=self.transform8x8, yuv_format=self.yuv_format)
        prm = slice_params(L, cfg, 0, [(k - 1 - r) % self.nslots for r in range(nref)], [k - 1 - r for r in range(nref)], self.disable_idc, self.epzs, 2 * k)
        if per < nmb:
            prm["num_slices"] = (nmb + per - 1) // per
        J.seq_set_frame(k % self.depth, raw, sw, sh)
        J.seq_encode(k % self.depth, prm, k % self.nslots, 1, self.stream_records)
        self.npic += 1

    def finish(self):
        for k in range(max(0, self.npic - self.depth), self.npic):
            self.collect(k)
        self.J.synchronize()
        return [self.results[k] for k in range(self.npic)]


class BatchEncoder:
    """IPPP with the P pictures in launches of several pictures each (jmhip_seq_batch): the I picture and the first P pictures (fewer references than num_ref) through
    jmhip_seq_encode, then batches of `sizes` pictures in turn; picture k goes to slot k % nslots."""

    def __init__(self, W, H, qp, R, num_ref, lambdas, sizes, nslots, cabac=0, search_mode=-1, transform8x8=0, yuv_format=1, workgroups=0, slice_mbs=0, disable_idc=0, reserve=0, epzs=None):
        import torch
        self.slice_mbs, self.disable_idc, self.epzs = slice_mbs, disable_idc, dict(epzs or {})
        import jm_amd.lib as L
        self.L, self.torch = L, torch
        self.W, self.H, self.qp, self.R, self.num_ref, self.lambdas, self.sizes, self.nslots = W, H, qp, R, num_ref, lambdas, list(sizes), nslots
        self.kw = dict(cabac=cabac, search_mode=search_mode, transform8x8=transform8x8, yuv_format=yuv_format)
        self.J = L.JmHip(W, H, search_range=max(R, 1), num_ref_slots=nslots, yuv_format=yuv_format)
        self.J.seq_open(1)
        if workgroups:
            self.J.set_pipeline_workgroups(workgroups)
        if reserve:
            self.J.seq_batch_reserve(reserve)                 # the launches' scratch ahead of the first launch (fewer pictures than the later launches have: it grows again)

    def params(self, k, st, nref):
        nmb = (self.W // 16) * (self.H // 16)
        lam_mf, lam_md = self.lambdas[st]
        per = self.slice_mbs if 0 < self.slice_mbs < nmb else nmb
        cfg = pyjmo.mbenc_cfg(self.W, self.H, st, 0, per, self.qp, self.R, nref, lam_mf, lam_md, **self.kw)
        prm = slice_params(self.L, cfg, 0, [(k - 1 - r) % self.nslots for r in range(nref)], [k - 1 - r for r in range(nref)], self.disable_idc, self.epzs, 2 * k)
        if per < nmb:
            prm["num_slices"] = (nmb + per - 1) // per
        return prm

    def run(self, frames, sw, sh):
        J, torch, L = self.J, self.torch, self.L
        nmb = (self.W // 16) * (self.H // 16)
        out = {}

        def keep(k, recs):
            out[k] = (recs, J.seq_get_recon(k % self.nslots), J.get_subplanes(k % self.nslots))
        k = 0
        while k < len(frames) and k < self.num_ref:              # pictures with fewer references than the batch's
            J.seq_set_frame(0, frames[k], sw, sh)
            J.seq_encode(0, self.params(k, 2 if k == 0 else 0, min(self.num_ref, k)), k % self.nslots, 1, False)
            J.seq_wait(0)
            keep(k, J.seq_records(0))
            k += 1
        turn = 0
        while k < len(frames):
            n = min(self.sizes[turn % len(self.sizes)], len(frames) - k)
            turn += 1
            d_raw = [torch.from_numpy(np.ascontiguousarray(frames[k + i])).cuda() for i in range(n)]
            d_rec = torch.zeros(n * nmb * L.MB_RECORD.itemsize, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            pics = [dict(d_raw=d_raw[i].data_ptr(), src_w=sw, src_h=sh, out_slot=(k + i) % self.nslots, ref_slot=[(k + i - 1 - r) % self.nslots for r in range(self.num_ref)],
                         ref_id=[k + i - 1 - r for r in range(self.num_ref)], poc_offset=2 * i, d_records=d_rec.data_ptr() + i * nmb * L.MB_RECORD.itemsize) for i in range(n)]
            J.seq_batch(self.params(k, 0, self.num_ref), pics)
            J.synchronize()
            recs = np.frombuffer(d_rec.cpu().numpy().tobytes(), L.MB_RECORD).reshape(n, nmb)
            # every picture's planes can only be read while its slot still holds it: the last nslots pictures of the batch
            for i in range(n):
                if i >= n - self.nslots:
                    keep(k + i, recs[i].copy())
                else:
                    out[k + i] = (recs[i].copy(), None, None)
            k += n
        return [out[i] for i in range(len(frames))]


def classic(W, H, qp, R, num_ref, lam, frames, sw=None, sh=None, slice_mbs=0, **kw):
    dev = DevSeqEncoder(W, H, qp, R, num_ref, lam, slice_mbs, together=slice_mbs > 0, **kw)      # slices: the picture's slices in one launch, their wavefronts side by side
    out = []
    for raw in frames:
        recs, pre, post = dev.encode(raw, sw or W, sh or H)
        out.append((recs, post, dev.J.get_subplanes(dev.refs[0][0])))
    dev.J.close()
    return out


def compare(want, got, what):
    assert len(want) == len(got)
    for n, (a, b) in enumerate(zip(want, got)):
        d = first_difference(mb_tap.canonical(as_oracle_records(a[0])), mb_tap.canonical(as_oracle_records(b[0])))
        assert d is None, (what, "picture", n, "record", d)
        if b[1] is None:                                      # a batch's picture whose slot a later picture of the batch took: the records say it all
            continue
        for p, (x, y) in enumerate(zip(a[1], b[1])):
            if not np.array_equal(x, y):
                bad = np.argwhere(x != y)
                raise AssertionError((what, "picture", n, "filtered plane", p, "first differences (row, column)", bad[:8].tolist(), len(bad)))
        if not np.array_equal(a[2], b[2]):
            bad = np.argwhere(a[2] != b[2])
            raise AssertionError((what, "picture", n, "sub-pel planes: first differences (plane, row, column)", bad[:8].tolist(), len(bad)))


@pytest.mark.parametrize("W,H,R,num_ref,qp,depth,wg,seed,kw", [
    (176, 144, 16, 1, 28, 2, 0, 1, {}),
    (176, 144, 16, 1, 28, 8, 0, 1, {}),
    (320, 192, 32, 1, 28, 4, 0, 2, {}),
    (320, 192, 32, 3, 36, 4, 0, 3, {}),                      # three references, all of them possibly still in the making
    (640, 368, 32, 1, 28, 4, 0, 4, {}),
    (640, 368, 8, 2, 20, 8, 12, 5, {}),                      # a short reach: pictures three diagonals apart; few workgroups per picture
    (64, 48, 32, 2, 28, 4, 0, 6, {}),                        # smaller than the search window: every macroblock waits for the reference's last one
    (16, 16, 16, 1, 28, 3, 0, 7, {}),                        # a single macroblock: all four borders of the planes from one workgroup
    (208, 160, 16, 1, 28, 1, 0, 8, {}),                      # depth 1: the launch alone (filter + interpolation inside it)
    (320, 192, 16, 2, 28, 4, 0, 9, {"cabac": 1}),
    (320, 192, 16, 1, 28, 4, 0, 10, {"transform8x8": 1, "cabac": 1}),
    (320, 192, 32, 2, 28, 4, 0, 11, {"search_mode": 1}),      # fast full search
    (320, 192, 16, 1, 28, 4, 0, 12, {"yuv_format": 2}),
    (320, 192, 32, 2, 32, 3, 5, 13, {"yuv_format": 2, "transform8x8": 1, "cabac": 1, "search_mode": 1}),
    # EPZS: a search asks for what it is about to read of a reference in the making when its centre is known; the temporal predictors read the reference's motion directly
    (320, 192, 16, 1, 28, 4, 0, 14, {"search_mode": 3}),
    (320, 192, 32, 3, 28, 8, 0, 15, {"search_mode": 3, "cabac": 1}),
    (640, 368, 32, 2, 36, 4, 0, 16, {"search_mode": 3, "transform8x8": 1, "cabac": 1}),
    (208, 160, 8, 5, 24, 3, 7, 17, {"search_mode": 3, "epzs": dict(pattern=5, dual=6, fixed=3, aggressive=1, temporal=1, spatial_mem=1, blocktype=1)}),
    (320, 192, 16, 2, 28, 4, 0, 18, {"search_mode": 3, "yuv_format": 2, "epzs": dict(temporal=0, spatial_mem=0, blocktype=0)}),
    (64, 48, 32, 2, 28, 4, 0, 19, {"search_mode": 3}),
    # EPZS P pictures run as four-wave workgroups, two to a compute unit (k_mb_pipe_epzs4*), where two workgroups' LDS fit one (up to five references): sixteen pictures in
    # flight; the eight-wave form (what six references and more get) forced on; six references
    (320, 192, 16, 1, 28, 16, 0, 40, {"search_mode": 3, "cabac": 1}),
    (320, 192, 32, 2, 28, 4, 0, 41, {"search_mode": 3, "transform8x8": 1, "cabac": 1, "waves": 8}),
    (320, 192, 16, 5, 32, 6, 0, 42, {"search_mode": 3}),
    (208, 160, 16, 6, 28, 4, 0, 43, {"search_mode": 3}),
    # pictures of several slices (SliceMode 1): one launch per picture in the picture's wavefront order; the loop filter across the slices' edges, or not (disable_idc 2)
    (320, 192, 32, 1, 28, 4, 0, 44, {"slice_mbs": 60}),                  # four slices of three macroblock rows
    (320, 192, 16, 2, 32, 4, 0, 45, {"slice_mbs": 50, "cabac": 1}),       # slices that start mid-row
    (640, 368, 32, 1, 28, 8, 0, 46, {"slice_mbs": 120, "disable_idc": 2}),
    (320, 192, 16, 2, 28, 4, 0, 47, {"slice_mbs": 70, "search_mode": 3}),
    (320, 192, 16, 1, 28, 3, 0, 48, {"slice_mbs": 33, "transform8x8": 1, "cabac": 1, "search_mode": 1}),
])
def test_pictures_in_flight_equal_picture_after_picture(W, H, R, num_ref, qp, depth, wg, seed, kw, monkeypatch):
    kw = dict(kw)
    if kw.pop("waves", 0) == 8:
        monkeypatch.setenv("JMHIP_EPZS_WAVES", "8")
    f = int(192 * 2 ** ((qp - 28) / 6))
    lam = LAMBDAS if qp == 28 else {2: ([f] * 3, f), 0: ([f, f + 3, f + 5], f + 1)}
    nfr = max(7, depth + 3)
    if kw.get("yuv_format") == 2:
        frames = [np.concatenate([fr[:W * H], np.repeat(fr[W * H:].reshape(2, H // 2, W // 2), 2, axis=1).ravel()]) for fr in synthetic_clip(W, H, nfr, seed)]
    else:
        frames = synthetic_clip(W, H, nfr, seed)
    want = classic(W, H, qp, R, num_ref, lam, frames, **kw)
    fl = FlightEncoder(W, H, qp, R, num_ref, lam, depth, wg, **kw)
    for raw in frames:
        fl.submit(raw, W, H)
    got = fl.finish()
    fl.J.close()
    compare(want, got, (W, H, R, num_ref, depth))


@pytest.mark.parametrize("W,H,R,num_ref,qp,sizes,nslots,wg,seed,kw", [
    (176, 144, 16, 1, 28, [3], 4, 0, 21, {}),
    (320, 192, 32, 1, 28, [8], 10, 0, 22, {}),                # every picture its own slot
    (320, 192, 32, 1, 28, [9], 3, 0, 23, {}),                 # three slots in turn: a picture starts when the one two before it is done
    (320, 192, 32, 1, 28, [7], 2, 0, 24, {}),                 # two slots: one picture after the other, inside one launch
    (320, 192, 32, 3, 36, [6], 6, 0, 25, {}),                 # three references
    (640, 368, 32, 2, 28, [2, 5, 1], 5, 0, 26, {"reserve": 3}),   # batch after batch: references from the batch before; a batch of one; scratch reserved for three pictures, then five come
    (640, 368, 8, 2, 20, [8], 12, 0, 27, {}),                 # a short reach: pictures three diagonals apart
    (64, 48, 32, 2, 28, [6], 5, 0, 28, {}),                   # smaller than the search window
    (16, 16, 16, 1, 28, [5], 3, 0, 29, {}),                   # a single macroblock per picture
    (320, 192, 16, 2, 28, [6], 8, 3, 30, {"cabac": 1, "transform8x8": 1}),     # three workgroups for everything
    (320, 192, 32, 2, 28, [6], 8, 0, 31, {"search_mode": 1}),
    (320, 192, 16, 1, 30, [6], 4, 0, 32, {"yuv_format": 2, "search_mode": 1, "transform8x8": 1}),
    (320, 192, 32, 1, 28, [8], 10, 0, 33, {"slice_mbs": 60}),             # pictures of four slices
    (640, 368, 32, 2, 28, [6], 8, 0, 34, {"slice_mbs": 130, "disable_idc": 2, "cabac": 1}),      # slices that start mid-row, no filtering across their edges
    # EPZS in the one queue (round 5): every search checks what it reaches against the queue's order (ez_ensure_ref)
    (320, 192, 16, 2, 28, [6], 8, 0, 41, {"search_mode": 3}),
    (208, 160, 8, 3, 24, [7], 9, 0, 42, {"search_mode": 3, "epzs": dict(pattern=5, dual=6, fixed=3, aggressive=1, temporal=1, spatial_mem=1, blocktype=1)}),
    (320, 192, 16, 1, 28, [7], 4, 0, 43, {"search_mode": 3, "cabac": 1, "transform8x8": 1}),     # four slots in turn
    (320, 192, 16, 2, 28, [5], 8, 0, 44, {"search_mode": 3, "yuv_format": 2, "epzs": dict(temporal=0, spatial_mem=0, blocktype=0)}),
    (640, 368, 32, 2, 28, [2, 5, 1], 6, 0, 45, {"search_mode": 3, "cabac": 1}),                   # batch after batch, temporal predictors from the batch before
    (640, 368, 32, 1, 32, [9], 12, 5, 46, {"search_mode": 3}),                                      # five workgroups (ten four-wave ones) for everything
])
def test_pictures_in_one_launch_equal_picture_after_picture(W, H, R, num_ref, qp, sizes, nslots, wg, seed, kw):
    """jmhip_seq_batch: the macroblocks of several consecutive pictures from one queue, ordered by wavefront index + lag x picture"""
    f = int(192 * 2 ** ((qp - 28) / 6))
    lam = LAMBDAS if qp == 28 else {2: ([f] * 3, f), 0: ([f, f + 3, f + 5], f + 1)}
    nfr = num_ref + sum(sizes) + (1 if len(sizes) == 1 else 0)
    if kw.get("yuv_format") == 2:
        frames = [np.concatenate([fr[:W * H], np.repeat(fr[W * H:].reshape(2, H // 2, W // 2), 2, axis=1).ravel()]) for fr in synthetic_clip(W, H, nfr, seed)]
    else:
        frames = synthetic_clip(W, H, nfr, seed)
    want = classic(W, H, qp, R, num_ref, lam, frames, **{k: v for k, v in kw.items() if k != "reserve"})
    be = BatchEncoder(W, H, qp, R, num_ref, lam, sizes, nslots, workgroups=wg, **kw)
    got = be.run(frames, W, H)
    be.J.close()
    compare(want, got, (W, H, R, num_ref, sizes, nslots))


def test_epzs_pictures_in_one_launch_with_the_eight_wave_kernels(monkeypatch):
    """EPZS in the one queue with JMHIP_EPZS_WAVES=8 (k_mb_pipe_epzs / _t8: the form a slice with more references than two four-wave workgroups' LDS allow falls back to)"""
    monkeypatch.setenv("JMHIP_EPZS_WAVES", "8")
    for (W, H, R, num_ref, kw) in ((320, 192, 16, 2, {}), (320, 192, 16, 1, {"cabac": 1, "transform8x8": 1})):
        frames = synthetic_clip(W, H, num_ref + 7, 47)
        want = classic(W, H, 28, R, num_ref, LAMBDAS, frames, search_mode=3, **kw)
        be = BatchEncoder(W, H, 28, R, num_ref, LAMBDAS, [6], 8, search_mode=3, **kw)
        got = be.run(frames, W, H)
        be.J.close()
        compare(want, got, ("eight waves", kw))


def test_epzs_launch_of_several_pictures_given_up_and_coded_again(monkeypatch):
    """EPZS in the one queue: with a queue lag far below what the searches reach and one workgroup drawing the tickets, a search asks for a macroblock whose ticket is not
    out -- the launch is given up (JMHIP_EREACH, -6) instead of waiting for good, and the same context then codes the same pictures launch by launch, with the results of
    the pictures coded one after another; with the lag the library chooses the launch goes through"""
    import jm_amd.lib as L
    W, H, R = 320, 192, 16
    frames = synthetic_clip(W, H, 9, 5)
    want = classic(W, H, 28, R, 1, LAMBDAS, frames, search_mode=3)
    monkeypatch.setenv("JMHIP_EPZS_BATCH_LAG", "4")
    be = BatchEncoder(W, H, 28, R, 1, LAMBDAS, [8], 10, workgroups=1, search_mode=3)
    with pytest.raises(L.JmHipError) as ei:
        be.run(frames, W, H)
    assert ei.value.code == -6, ei.value
    J, got = be.J, []
    for k, raw in enumerate(frames):                             # the same context, a launch per picture
        J.seq_set_frame(0, raw, W, H)
        J.seq_encode(0, be.params(k, 2 if k == 0 else 0, min(1, k)), k % be.nslots, 1, False)
        J.seq_wait(0)
        got.append((J.seq_records(0), J.seq_get_recon(k % be.nslots), J.get_subplanes(k % be.nslots)))
    J.synchronize()
    compare(want, got, "coded again")
    monkeypatch.delenv("JMHIP_EPZS_BATCH_LAG")
    be2 = BatchEncoder(W, H, 28, R, 1, LAMBDAS, [8], 10, workgroups=1, search_mode=3)
    compare(want, be2.run(frames, W, H), "the library's lag")
    J.close()
    be2.J.seq_batch_lag(4)                                       # the caller's knob (jmhip_seq_batch_lag): given up again, then a whole wavefront apart -- picture after picture in one launch
    with pytest.raises(L.JmHipError) as ei:
        be2.run(frames, W, H)
    assert ei.value.code == L.EREACH
    be2.J.seq_batch_lag(W // 16 + 2 * (H // 16 - 1) + 1)
    compare(want, be2.run(frames, W, H), "a whole wavefront apart")
    be2.J.close()


def test_batch_refuses_what_it_does_not_cover():
    import torch
    import jm_amd.lib as L
    W, H = 176, 144
    nmb = (W // 16) * (H // 16)
    J = L.JmHip(W, H, search_range=16, num_ref_slots=4, yuv_format=1)
    raw = torch.zeros(W * H * 3 // 2, dtype=torch.uint8, device="cuda")
    rec = torch.zeros(nmb * L.MB_RECORD.itemsize, dtype=torch.uint8, device="cuda")
    pic = dict(d_raw=raw.data_ptr(), src_w=W, src_h=H, out_slot=1, ref_slot=[0], ref_id=[0], d_records=rec.data_ptr())
    p = slice_params(L, pyjmo.mbenc_cfg(W, H, 0, 0, nmb, 28, 16, 1, *LAMBDAS[0]), 0, [0], [0])
    with pytest.raises(L.JmHipError, match="jmhip_seq_open"):
        J.seq_batch(p, [pic])
    J.seq_open(1)
    with pytest.raises(L.JmHipError, match="holds no picture"):
        J.seq_batch(p, [pic])                                   # slot 0 was never filled
    with pytest.raises(L.JmHipError, match="its reference"):
        J.seq_batch(p, [dict(pic, out_slot=0)])
    with pytest.raises(L.JmHipError, match="P pictures"):
        J.seq_batch(slice_params(L, pyjmo.mbenc_cfg(W, H, 2, 0, nmb, 28, 16, 0, *LAMBDAS[2]), 0, [], []), [pic])
    with pytest.raises(L.JmHipError):
        J.seq_batch(p, [dict(pic, d_records=0)])
    J.close()


@pytest.mark.parametrize("kind,R,num_ref,qp", [("flat", 32, 1, 44), ("noise", 16, 2, 12), ("stripes", 32, 2, 28), ("still", 32, 1, 30)])
def test_pictures_in_flight_hard_content(kind, R, num_ref, qp):
    """content that makes every edge filter (noise, coarse quantiser) or none (still), and vectors at the limit of the reach (stripes)"""
    W, H = 256, 160
    f = int(192 * 2 ** ((qp - 28) / 6))
    lam = {2: ([f] * 3, f), 0: ([f, f + 3, f + 5], f + 1)}
    frames = hard_clip(kind, W, H, 6, 77)
    want = classic(W, H, qp, R, num_ref, lam, frames)
    fl = FlightEncoder(W, H, qp, R, num_ref, lam, 4)
    for raw in frames:
        fl.submit(raw, W, H)
    compare(want, fl.finish(), kind)
    fl.J.close()


def test_pictures_in_flight_streamed_records_and_repeated_runs():
    """records read macroblock by macroblock from pinned memory while pictures are in flight; the same sequence five times over gives the same bytes"""
    W, H, R = 320, 192, 16
    frames = synthetic_clip(W, H, 9, 31)
    want = classic(W, H, 28, R, 1, LAMBDAS, frames)
    for rep in range(5):
        fl = FlightEncoder(W, H, 28, R, 1, LAMBDAS, 4, stream_records=(rep & 1) == 0)
        for raw in frames:
            fl.submit(raw, W, H)
        compare(want, fl.finish(), ("repetition", rep))
        fl.J.close()


def test_streamed_records_are_still_there_after_synchronize():
    """jmhip_synchronize waits for every picture in flight and takes it out of flight, but an entry that streams its records to the host keeps answering jmhip_seq_record for
    its picture afterwards (the entry's streaming state used to be cleared, and the call then failed with JMHIP_EINVAL)."""
    W, H, R = 208, 160, 16
    frames = synthetic_clip(W, H, 3, 77)
    want = classic(W, H, 28, R, 1, LAMBDAS, frames)
    fl = FlightEncoder(W, H, 28, R, 1, LAMBDAS, 3, stream_records=True)
    nmb = (W // 16) * (H // 16)
    for raw in frames:
        fl.submit(raw, W, H)                                  # three pictures, three entries: nothing collected yet
    fl.J.synchronize()
    for k in range(3):
        recs = fl.J.seq_records_streamed(k, 0, nmb)
        d = first_difference(mb_tap.canonical(as_oracle_records(want[k][0])), mb_tap.canonical(as_oracle_records(recs)))
        assert d is None, ("picture", k, "record", d)
    fl.J.close()


def test_entries_made_beside_the_first_picture(monkeypatch):
    """jmhip_seq_open makes its first entry itself and the others on a thread that starts with the first launch (mbpipe_host.inc: seq_join): a context closed before any launch,
    one closed right after its first launch, a sequence whose FIRST launch names the last entry, and the same sequence with every entry made inside jmhip_seq_open all behave"""
    import jm_amd.lib as L
    W, H, R = 176, 144, 16
    frames = synthetic_clip(W, H, 6, 77)
    want = classic(W, H, 28, R, 1, LAMBDAS, frames)
    J = L.JmHip(W, H, search_range=R, num_ref_slots=8, yuv_format=1)
    J.seq_open(6)
    J.seq_close()                                            # nothing was launched: the thread has nothing to make
    J.seq_open(4)
    J.seq_set_frame(3, frames[0], W, H)                     # the last entry first: joins the thread
    J.seq_close()
    J.close()
    for inline in (False, True):
        if inline:
            monkeypatch.setenv("JMHIP_SEQ_OPEN_INLINE", "1")
        fl = FlightEncoder(W, H, 28, R, 1, LAMBDAS, 4)
        fl.submit(frames[0], W, H)
        if not inline:
            fl.J.seq_close()                                 # right behind the first launch: the thread is under way
            fl.J.seq_open(4)
            fl.npic, fl.results = 0, {}
            fl.submit(frames[0], W, H)
        for raw in frames[1:]:
            fl.submit(raw, W, H)
        compare(want, fl.finish(), ("inline", inline))
        fl.J.close()


def test_seq_refuses_what_it_does_not_cover():
    import jm_amd.lib as L
    W, H = 176, 144
    nmb = (W // 16) * (H // 16)
    J = L.JmHip(W, H, search_range=16, num_ref_slots=4, yuv_format=1)
    with pytest.raises(L.JmHipError):
        J.seq_encode(0, slice_params(L, pyjmo.mbenc_cfg(W, H, 2, 0, nmb, 28, 16, 0, *LAMBDAS[2]), 0, [], []), 0)       # no jmhip_seq_open
    J.seq_open(2)
    raw = synthetic_clip(W, H, 1, 1)[0]
    J.seq_set_frame(0, raw, W, H)
    cfg = pyjmo.mbenc_cfg(W, H, 2, 0, nmb // 2, 28, 16, 0, *LAMBDAS[2])
    with pytest.raises(L.JmHipError, match="slices that cover the picture"):                        # half a picture
        J.seq_encode(0, slice_params(L, cfg, 0, [], []), 0)
    with pytest.raises(L.JmHipError):
        J.seq_encode(2, slice_params(L, pyjmo.mbenc_cfg(W, H, 2, 0, nmb, 28, 16, 0, *LAMBDAS[2]), 0, [], []), 0)       # entry out of range
    with pytest.raises(L.JmHipError):
        J.seq_encode(0, slice_params(L, pyjmo.mbenc_cfg(W, H, 2, 0, nmb, 28, 16, 0, *LAMBDAS[2]), 0, [], []), 9)       # slot out of range
    J.close()


def test_1080p_sequence_in_flight_equals_the_reference_encoder():
    """configs[1] at its own size: the clip of SURVEY Appendix A, G2r's flags, eight pictures in flight; the first six pictures' records against the REAL
    encoder's (tests/golden/mb_low_g6r.npz: the tapped lencod, six pictures), the later ones and every filtered picture / plane against the picture-after-picture path"""
    import bench
    from test_gpu_mbenc import load_case
    c = load_case("g6r")
    W, H, R = c["W"], c["H"], c["R"]
    nmb = (W // 16) * (H // 16)
    frames = bench.yuv_frames(10)
    want = classic(W, H, c["qp"], R, 1, c["lam"], frames[:4], c["sw"], c["sh"])
    fl = FlightEncoder(W, H, c["qp"], R, 1, c["lam"], 8)
    for raw in frames:
        fl.submit(raw, c["sw"], c["sh"])
    got = fl.finish()
    fl.J.close()
    compare(want, got[:4], "1080p")
    for n in range(c["nfr"]):
        d = first_difference(c["records"][n * nmb:(n + 1) * nmb], mb_tap.canonical(as_oracle_records(got[n][0])))
        assert d is None, ("against the reference encoder, picture", n, d)


def test_1080p_sequence_in_one_launch_equals_the_reference_encoder():
    """configs[1] at its own size, the P pictures in one launch (jmhip_seq_batch, what bench.py times): the six pictures of tests/golden/mb_low_g6r.npz against the REAL encoder's
    records, four more against the picture-after-picture path, with four slots in turn (pictures wait for their slot inside the launch)"""
    import bench
    from test_gpu_mbenc import load_case
    c = load_case("g6r")
    W, H, R = c["W"], c["H"], c["R"]
    nmb = (W // 16) * (H // 16)
    frames = bench.yuv_frames(10)
    want = classic(W, H, c["qp"], R, 1, c["lam"], frames, c["sw"], c["sh"])
    be = BatchEncoder(W, H, c["qp"], R, 1, c["lam"], [9], 4)
    got = be.run(frames, c["sw"], c["sh"])
    be.J.close()
    compare(want, got, "1080p, one launch")
    for n in range(c["nfr"]):
        d = first_difference(c["records"][n * nmb:(n + 1) * nmb], mb_tap.canonical(as_oracle_records(got[n][0])))
        assert d is None, ("against the reference encoder, picture", n, d)


def test_1080p_epzs_sequence_in_flight_equals_the_reference_encoder():
    """configs[2]'s search at its own size with pictures in flight: six pictures of the 1080p clip, EPZS with the shipped switches, CABAC, up to five references -- every
    record against the REAL encoder's (tests/golden/mb_low_g6e.npz)"""
    import bench
    from test_gpu_mbenc import load_case
    c = load_case("g6e")
    W, H = c["W"], c["H"]
    nmb = (W // 16) * (H // 16)
    frames = bench.yuv_frames(c["nfr"])
    fl = FlightEncoder(W, H, c["qp"], c["R"], c["num_ref"], c["lam"], 6, cabac=c["cabac"], search_mode=3, epzs=c["epzs"])
    for raw in frames:
        fl.submit(raw, c["sw"], c["sh"])
    got = fl.finish()
    fl.J.close()
    for n in range(c["nfr"]):
        d = first_difference(c["records"][n * nmb:(n + 1) * nmb], mb_tap.canonical(as_oracle_records(got[n][0])))
        assert d is None, ("against the reference encoder, picture", n, d)


def test_1080p_epzs_pictures_in_one_launch_equal_the_reference_encoder():
    """configs[2]'s search at its own size with the P pictures in ONE launch (jmhip_seq_batch, search_mode 3: every search checked against the queue's order): eight pictures of
    the 1080p clip, EPZS with the shipped switches, CABAC, one reference -- so that all seven P pictures are one launch -- every record against the REAL encoder's
    (tests/golden/mb_low_g8e.npz), and the same pictures in flight (sixteen entries: all eight at once) against the same records"""
    import bench
    from test_gpu_mbenc import load_case
    c = load_case("g8e")
    W, H = c["W"], c["H"]
    nmb = (W // 16) * (H // 16)
    assert c["num_ref"] == 1 and c["nfr"] == 8
    frames = bench.yuv_frames(c["nfr"])
    be = BatchEncoder(W, H, c["qp"], c["R"], 1, c["lam"], [7], 10, cabac=c["cabac"], search_mode=3, epzs=c["epzs"])
    got = be.run(frames, c["sw"], c["sh"])
    be.J.close()
    fl = FlightEncoder(W, H, c["qp"], c["R"], 1, c["lam"], 8, cabac=c["cabac"], search_mode=3, epzs=c["epzs"])
    for raw in frames:
        fl.submit(raw, c["sw"], c["sh"])
    got_f = fl.finish()
    fl.J.close()
    for what, g in (("in one launch", got), ("in flight", got_f)):
        for n in range(c["nfr"]):
            d = first_difference(c["records"][n * nmb:(n + 1) * nmb], mb_tap.canonical(as_oracle_records(g[n][0])))
            assert d is None, (what, "against the reference encoder, picture", n, d)


def test_forty_epzs_pictures_in_flight_and_in_one_launch_equal_the_reference_encoder():
    """A long EPZS sequence (tests/golden/mb_low_m2e40.npz: forty pictures, two references, the shipped switches -- JM's 16-bit visited-map stamp wraps round several times; the
    oracle counts no aliased candidate on it, tests/test_oracle_mbenc.py): sixteen pictures in flight, and the P pictures in launches of 17 / 9 / the rest, every record
    against the REAL encoder's"""
    from test_gpu_mbenc import load_case, clip_bytes
    c = load_case("m2e40")
    W, H, nfr = c["W"], c["H"], c["nfr"]
    nmb = (W // 16) * (H // 16)
    data = clip_bytes("m2e40", c)
    fs = len(data) // nfr
    frames = [data[n * fs:(n + 1) * fs] for n in range(nfr)]
    fl = FlightEncoder(W, H, c["qp"], c["R"], c["num_ref"], c["lam"], 16, cabac=c["cabac"], search_mode=3, epzs=c["epzs"])
    for raw in frames:
        fl.submit(raw, c["sw"], c["sh"])
    got = fl.finish()
    fl.J.close()
    be = BatchEncoder(W, H, c["qp"], c["R"], c["num_ref"], c["lam"], [17, 9, 40], 20, cabac=c["cabac"], search_mode=3, epzs=c["epzs"])
    got1 = be.run(frames, c["sw"], c["sh"])
    be.J.close()
    for what, g in (("in flight", got), ("in one launch", got1)):
        for n in range(nfr):
            d = first_difference(c["records"][n * nmb:(n + 1) * nmb], mb_tap.canonical(as_oracle_records(g[n][0])))
            assert d is None, (what, "against the reference encoder, picture", n, d)


def oracle_sequence(W, H, qp, R, num_ref, lam, frames, sw, sh, slice_mbs=0, disable_idc=0, **kw):
    """the ORACLE's picture-after-picture sequence (mbenc_util.SeqEncoder: oracle/jmo_mbenc.c per slice, the oracle's loop filter and interpolation): records, filtered planes,
    the sixteen sub-pel planes -- in the shape `compare` takes"""
    yuv = kw.get("yuv_format", 1)
    enc = mbenc_util.SeqEncoder(W, H, qp, R, num_ref, lam, slice_mbs, disable_idc=disable_idc, cabac=kw.get("cabac", 0), search_mode=kw.get("search_mode", -1),
                                transform8x8=kw.get("transform8x8", 0), yuv_format=yuv)
    out = []
    for raw in frames:
        recs, _, _, post = enc.encode(pyjmo.load_frame(raw, sw, sh, W, H, yuv))
        out.append((recs, [np.asarray(p, np.uint8) for p in post], enc.refs[0][0].planes))
    return out


# The forms of this file compared with the ORACLE directly (not with the device's own picture-after-picture path: a defect common to both forms of mbpipe_final.inc would pass
# there): slices that start mid-row + CABAC, 4:2:2, the 8x8 transform + CABAC, content chosen against the pruning and the tie-breaking -- each in flight AND in one launch
@pytest.mark.parametrize("W,H,R,num_ref,qp,seed,kw", [
    (320, 192, 16, 2, 28, 101, {"slice_mbs": 50, "cabac": 1}),
    (320, 192, 16, 1, 28, 102, {"yuv_format": 2}),
    (320, 192, 16, 1, 28, 103, {"transform8x8": 1, "cabac": 1}),
    (208, 160, 16, 1, 28, 104, {"hard": "stripes"}),
    (208, 160, 16, 1, 28, 105, {"hard": "noise"}),
])
def test_pictures_in_flight_and_in_one_launch_equal_the_oracle(W, H, R, num_ref, qp, seed, kw):
    kw = dict(kw)
    hard = kw.pop("hard", None)
    nfr = 6
    if hard:
        frames = hard_clip(hard, W, H, nfr, seed)
    elif kw.get("yuv_format") == 2:
        frames = [np.concatenate([fr[:W * H], np.repeat(fr[W * H:].reshape(2, H // 2, W // 2), 2, axis=1).ravel()]) for fr in synthetic_clip(W, H, nfr, seed)]
    else:
        frames = synthetic_clip(W, H, nfr, seed)
    want = oracle_sequence(W, H, qp, R, num_ref, LAMBDAS, frames, W, H, **kw)
    fl = FlightEncoder(W, H, qp, R, num_ref, LAMBDAS, 4, 0, **kw)
    for raw in frames:
        fl.submit(raw, W, H)
    got = fl.finish()
    fl.J.close()
    compare(want, got, ("in flight against the oracle", W, H, kw, hard))
    be = BatchEncoder(W, H, qp, R, num_ref, LAMBDAS, [nfr - num_ref], nfr + 1, **kw)
    got = be.run(frames, W, H)
    be.J.close()
    compare(want, got, ("in one launch against the oracle", W, H, kw, hard))
