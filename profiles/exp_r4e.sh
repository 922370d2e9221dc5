cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; rm -rf $O; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_seq.py -x -q -m gpu > $O/pytest_seq.txt 2>&1; tail -3 $O/pytest_seq.txt
GPU_MAX_HW_QUEUES=16 timeout 120 python profiles/seq_probe.py 44 8 2>&1 | grep depth
JMHIP_SEQ_BANDS=1 GPU_MAX_HW_QUEUES=16 timeout 120 python profiles/seq_probe.py 44 8 2>&1 | grep depth
for v in nobands bands; do
  if [ $v = bands ]; then export JMHIP_SEQ_BANDS=1; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$v -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --streams 0 > /dev/null 2> $O/pmc1_$v.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$v -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --streams 0 > /dev/null 2> $O/pmc2_$v.err
done
