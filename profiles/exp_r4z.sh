mkdir -p gpurun_out/r4z
for q in 24 16 24; do GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --streams 0 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q:', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"; done > gpurun_out/r4z/bench_queues.txt 2>&1
cat gpurun_out/r4z/bench_queues.txt
