"""Kernel time of k_mb_pipe_b at 1080p (golden g3b: encoder_main.cfg's search and B settings, RDO off, I P B of the synthetic clip): python profiles/r05_b_timing.py [tag]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import test_gpu_bslice as TB
import test_oracle_mbenc as TO

tag = sys.argv[1] if len(sys.argv) > 1 else "g3b"
c = TO.load_case(tag)
z = c["z"]
ov = dict(s.split("=") for s in z["overrides"])
args = (c["W"], c["H"], c["qp"], c["R"], c["num_ref"], c["lam"], c["slice_mbs"], c["mv_limit"], c["didc"])
kw = dict(cabac=c.get("cabac", 0), search_mode=c["search_mode"], transform8x8=c["t8"], yuv_format=c["yuv"], offsets=c["offsets"], inter_valid=c["inter_valid"], qpc=c["qpc"], qpc_cr_delta=c["qpc_cr_delta"], qp_p=c["qp_p"])
raw = TB.raw_frames(c, tag)
lam_b = ([int(x) for x in z["lambda_b"][:3]], int(z["lambda_b"][3]))
for me in (1, 0):
    dev = TB.DevSeqEncoderB(*args, qpc_p=c["qpc_p"], qpc_cr_delta_p=c["qpc_cr_delta_p"], **kw)
    dev.J.enable_timing(True)
    bsw = dict(TO.b_switches(ov, z), bipred_me=me)
    for n in range(len(z["slice_type"])):
        st, poc = int(z["slice_type"][n]), int(z["poc"][n])
        if st == 1:
            l0 = [int(p) for p in z["ref_poc"][n][:int(z["num_ref_pic"][n])]]
            l1 = [int(p) for p in z["poc_l1"][n][:int(z["num_ref1_pic"][n])]]
            for rep in range(3):
                dev.encode_b(raw[poc // 2], c["sw"], c["sh"], l0, l1, lam_b, int(z["qp_b"]), bsw, qpc_b=int(z["qpc_b"]), qpc_cr_delta_b=int(z["qpc_v_b"]) - int(z["qpc_b"]))
                print(f"{tag}: B picture {n} (BiPredMotionEstimation {me}, {len(l0)} + {len(l1)} references, search_mode {c['search_mode']}): k_mb_pipe_b{'_t8' if c['t8'] else ''} {dev.J.last_kernel_ms(5):.2f} ms", flush=True)
        else:
            dev.encode_ref(raw[poc // 2], c["sw"], c["sh"], poc)
            print(f"{tag}: picture {n} type {st}: {dev.J.last_kernel_ms(5):.2f} ms", flush=True)
    dev.J.close()
