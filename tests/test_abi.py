"""The C-ABI boundary: libjmhip.so loads without a GPU, exports every entry point include/jmhip.h declares, its POD
structs have the sizes the header documents, and creating a context without a device fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "jmhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(jmhip_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_entry_point_is_exported():
    from jm_amd.lib import load_library
    lib = load_library()
    names = declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_sizes_match_the_header():
    from jm_amd import lib as L
    assert L.ME_JOB.itemsize == 192 and L.ME_RESULT.itemsize == 328
    assert L.SUBPEL_JOB.itemsize == 36 and L.TQ_OUT.itemsize == 104
    assert L.DB_MB.itemsize == 28 and L.DB_MOTION.itemsize == 16
    assert L.TQC_PARAMS.itemsize == 240 and L.TQC_MB.itemsize == 16 and L.TQC_OUT.itemsize == 808
    assert L.TQ8_PARAMS.itemsize == 800 and L.TQ8_OUT.itemsize == 408 and L.DC_OUT.itemsize == 52
    assert L.MC_LUMA_BLK.itemsize == 20 and L.MC_CHROMA_BLK.itemsize == 72 and L.TQ16_OUT.itemsize == 1224
    assert L.IP4_BLK.itemsize == 16 and L.I16_MB.itemsize == 40 and L.I16_OUT.itemsize == 1040
    assert L.CAND.itemsize == 16 and L.PRED_CAND.itemsize == 32 and L.MC_WEIGHTS.itemsize == 12 and L.IC_MB.itemsize == 56 and L.IP8_BLK.itemsize == 28


def test_sequence_picture_layout_is_the_headers(tmp_path):
    """jmhip_seq_picture (jmhip_seq_batch's per-picture argument): the numpy layout of jm_amd/lib.py against what the C compiler makes of include/jmhip.h"""
    import os
    import subprocess
    from jm_amd import lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "jmhip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(jmhip_seq_picture), '
                   'offsetof(jmhip_seq_picture, d_raw), offsetof(jmhip_seq_picture, src_w), offsetof(jmhip_seq_picture, src_h), offsetof(jmhip_seq_picture, out_slot), '
                   'offsetof(jmhip_seq_picture, ref_slot), offsetof(jmhip_seq_picture, ref_id), offsetof(jmhip_seq_picture, poc_offset), offsetof(jmhip_seq_picture, d_records), sizeof(jmhip_slice_params)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    f = L.SEQ_PICTURE.fields
    assert got == [L.SEQ_PICTURE.itemsize] + [f[n][1] for n in ("d_raw", "src_w", "src_h", "out_slot", "ref_slot", "ref_id", "poc_offset", "d_records")] + [L.SLICE_PARAMS.itemsize], got


def test_return_codes_are_the_headers():
    """jm_amd.lib's return codes (JmHipError.code; EREACH = an EPZS launch of several pictures to be coded again launch by launch) against include/jmhip.h"""
    import os
    import re
    from jm_amd import lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "jmhip.h")).read()
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+JMHIP_(OK|E[A-Z]+)\s+(-?\d+)", text)}
    assert codes == {"OK": L.OK, "EINVAL": L.EINVAL, "ENODEV": L.ENODEV, "ENOMEM": L.ENOMEM, "EHIP": L.EHIP, "EUNSUPPORTED": L.EUNSUPPORTED, "EREACH": L.EREACH}, codes


def test_partition_table_is_the_abi_order():
    from jm_amd.lib import PARTITIONS, NPART
    assert NPART == 41 and len(PARTITIONS) == 41
    # (blocktype, x, y, w, h): 1 + 2 + 2 + 4 + 8 + 8 + 16 partitions, each tiling the macroblock
    for bt, n in zip(range(1, 8), (1, 2, 2, 4, 8, 8, 16)):
        parts = [p for p in PARTITIONS if p[0] == bt]
        assert len(parts) == n
        cover = np.zeros((16, 16), int)
        for _, x, y, w, h in parts:
            cover[y:y + h, x:x + w] += 1
        assert (cover == 1).all()


def test_create_without_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from jm_amd import JmHip
    from jm_amd.lib import JmHipError
    with pytest.raises(JmHipError) as e:
        JmHip(176, 144)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_bench_reads_the_timed_launch_from_a_counter_collection():
    """bench.py measures roofline.traffic with two rocprofv3 --pmc passes over its own command (child processes, outside the clock): the value it takes is the THIRD k_mb_pipe
    dispatch's (I picture, warm-up launch, timed launch) -- not the instrumented twin's, not another instance's, whatever order the rows come in; bytes = (2 x FETCH_SIZE +
    WRITE_SIZE) x 1024."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rows = [{"Dispatch_Id": "9", "Kernel_Name": "k_mb_pipe(PipeArgs)", "Counter_Name": "FETCH_SIZE", "Counter_Value": "3000.5"},       # the timed launch
            {"Dispatch_Id": "2", "Kernel_Name": "k_mb_pipe(PipeArgs)", "Counter_Name": "FETCH_SIZE", "Counter_Value": "100"},          # the I picture
            {"Dispatch_Id": "5", "Kernel_Name": "k_mb_pipe(PipeArgs)", "Counter_Name": "FETCH_SIZE", "Counter_Value": "700"},          # the warm-up launch
            {"Dispatch_Id": "1", "Kernel_Name": "k_load_frame(unsigned char const*, int)", "Counter_Name": "FETCH_SIZE", "Counter_Value": "1"},
            {"Dispatch_Id": "7", "Kernel_Name": "k_mb_pipe_prof(PipeArgs)", "Counter_Name": "FETCH_SIZE", "Counter_Value": "55555"},
            {"Dispatch_Id": "8", "Kernel_Name": "k_mb_pipe_epzs4_t8(PipeArgs)", "Counter_Name": "FETCH_SIZE", "Counter_Value": "44444"},
            {"Dispatch_Id": "11", "Kernel_Name": "k_mb_pipe(PipeArgs)", "Counter_Name": "FETCH_SIZE", "Counter_Value": "9"}]           # the check's launches behind it
    assert bench.timed_launch_counter(rows, "FETCH_SIZE") == 3000.5
    assert bench.timed_launch_counter(rows[:3][:2], "FETCH_SIZE") is None
    assert bench.timed_launch_counter(rows, "WRITE_SIZE") is None
    t = bench.traffic_bytes(3000.5, 1000.0)
    assert t["bytes_per_launch"] == int(round((2 * 3000.5 + 1000.0) * 1024)) and t["fetch_size_kb"] == 3000.5 and t["write_size_kb"] == 1000.0
