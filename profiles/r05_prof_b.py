"""Where a B macroblock's time goes in k_mb_pipe_b (JMHIP_MB_PROF=1 time stamps; golden g3b's B picture: 1080p, fast full search SR 32, 2 + 1 references, the bi-predictive search on / off):
python profiles/r05_prof_b.py"""
import os, sys, ctypes as C
os.environ["JMHIP_MB_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import test_gpu_bslice as TB
import test_oracle_mbenc as TO

tag = "g3b"
c = TO.load_case(tag)
z = c["z"]
ov = dict(s.split("=") for s in z["overrides"])
args = (c["W"], c["H"], c["qp"], c["R"], c["num_ref"], c["lam"], c["slice_mbs"], c["mv_limit"], c["didc"])
kw = dict(cabac=c.get("cabac", 0), search_mode=c["search_mode"], transform8x8=c["t8"], yuv_format=c["yuv"], offsets=c["offsets"], inter_valid=c["inter_valid"], qpc=c["qpc"], qpc_cr_delta=c["qpc_cr_delta"], qp_p=c["qp_p"])
raw = TB.raw_frames(c, tag)
lam_b = ([int(x) for x in z["lambda_b"][:3]], int(z["lambda_b"][3]))
nmb = 8160
for me in (1, 0):
    dev = TB.DevSeqEncoderB(*args, qpc_p=c["qpc_p"], qpc_cr_delta_p=c["qpc_cr_delta_p"], **kw)
    dev.J.enable_timing(True)
    bsw = dict(TO.b_switches(ov, z), bipred_me=me)
    for n in range(len(z["slice_type"])):
        st, poc = int(z["slice_type"][n]), int(z["poc"][n])
        if st != 1:
            dev.encode_ref(raw[poc // 2], c["sw"], c["sh"], poc)
            continue
        l0 = [int(p) for p in z["ref_poc"][n][:int(z["num_ref_pic"][n])]]
        l1 = [int(p) for p in z["poc_l1"][n][:int(z["num_ref1_pic"][n])]]
        dev.encode_b(raw[poc // 2], c["sw"], c["sh"], l0, l1, lam_b, int(z["qp_b"]), bsw, qpc_b=int(z["qpc_b"]), qpc_cr_delta_b=int(z["qpc_v_b"]) - int(z["qpc_b"]))
        ms = dev.J.last_kernel_ms(5)
        st_ = np.zeros((nmb, 32), np.uint64)
        assert dev.J.lib.jmhip_debug_read_mb_prof(dev.J.h, st_.ctypes.data_as(C.c_void_p), st_.nbytes) == 0
        t = st_.astype(np.int64)
        us = lambda a, b: np.median((t[:, b] - t[:, a]) / 100.0)
        print(f"BiPredMotionEstimation {me}: k_mb_pipe_b {ms:.1f} ms; per macroblock, median us from the barrier behind the staging (windows in LDS):")
        print("  wave 0: direct vectors + direct 8x8 costs %.1f; P8x8 blocks done at %.1f %.1f %.1f %.1f; everything coded and published at %.1f" % (us(1, 2), us(1, 3), us(1, 4), us(1, 5), us(1, 6), us(1, 16)))
        print("  roles done at (waves 0-3: block 0 of 8x8, 8x4, 4x8, 4x4; 4-6: 16x16 (+ Intra16x16, chroma decision), 16x8, 8x16; 7: Intra4x4): " + " ".join("%.1f" % us(1, 8 + w) for w in range(8)))
        print("  the 16x16 wave: list 0 reference 0 searched at %.1f, its bi-predictive search done at %.1f, reference 1 at %.1f, list 1 reference 0 at %.1f, its bi-predictive search at %.1f" % (us(1, 17), us(1, 18), us(1, 19), us(1, 21), us(1, 22)))
    dev.J.close()
