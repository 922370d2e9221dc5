cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_mbenc.py -x -q 2>&1 | tail -1
echo "--- bands on"; python profiles/wg_sweep.py 0 64 96 128 2>&1 | grep workgroups
echo "--- bands off"; JMHIP_MB_NO_BANDS=1 python profiles/wg_sweep.py 0 64 96 128 2>&1 | grep workgroups
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/bands_$c -o t -- python profiles/wg_sweep.py 0 > /dev/null 2>&1; done
