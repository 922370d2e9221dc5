# round 4, final verification: the GPU suite, smoke, the bench line as the driver runs it and with defaults
mkdir -p gpurun_out/r4z
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4z/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r4z/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r4z/bench_final.json 2> gpurun_out/r4z/bench_final.err; python -c "
import json; d=json.loads(open('gpurun_out/r4z/bench_final.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('records_equal_jm'), d['config'].get('records_equal_picture_after_picture')); print(d['configs2']['in_flight']); print(d['configs2']['end_to_end'].get('nine_pictures')); print(d['configs3'].get('eight_pictures'), d['configs3'].get('md5_is_g4r')); print(d['end_to_end']['p_frame_ms'], d['end_to_end']['md5_ok'], d['configs4'].get('p_frame_ms_hip'), d['configs4'].get('md5_equal'))"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 > gpurun_out/r4z/bench_final_20.json 2>/dev/null; cut -c1-200 gpurun_out/r4z/bench_final_20.json
