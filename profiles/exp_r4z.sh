mkdir -p gpurun_out/r4z
timeout 600 python -m pytest tests/test_gpu_seq.py -x -q -m gpu 2>&1 | tail -2
for a in "" "--steps 20 --warmup 5"; do timeout 300 python bench.py $a --no-cpu-baseline --no-end-to-end --streams 0 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a:', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"; done
timeout 300 python -m pytest tests/test_lencod_dropin.py -x -q -m gpu -k "epzs or g3e or g3h or g6e or configs2" 2>&1 | tail -2
