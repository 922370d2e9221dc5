# measurement aid (round 4): the four-wave form of the full-search kernel for 2160p pictures in one launch (bench.configs3_device), against the eight-wave form; the seq tests
mkdir -p gpurun_out/r4z
( JMHIP_FS_WAVES=8 timeout 300 python -c "import bench, json; print('eight waves', json.dumps(bench.configs3_device(0)))"
  timeout 300 python -c "import bench, json; print('by itself (four waves at 2160p)', json.dumps(bench.configs3_device(0)))" ) 2>&1 | grep -E "waves|rror" > gpurun_out/r4z/fs4_2160p.txt
cut -c1-1500 gpurun_out/r4z/fs4_2160p.txt
timeout 600 python -m pytest tests/test_gpu_seq.py -x -q -m gpu > gpurun_out/r4z/pytest_seq.txt 2>&1; tail -5 gpurun_out/r4z/pytest_seq.txt
JMHIP_FS_WAVES=4 timeout 300 python profiles/batch_probe.py 33 20 256,512 fs 2>&1 | tail -4 > gpurun_out/r4z/fs4_1080p.txt; cat gpurun_out/r4z/fs4_1080p.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 0 > gpurun_out/r4z/bench_q24.json 2> gpurun_out/r4z/bench_q24.err; python -c "
import json; d=json.loads(open('gpurun_out/r4z/bench_q24.json').read().strip().splitlines()[-1]); print(d['value'], d['configs2']['in_flight']); print(d['configs3']['device'].get('eight_slice_sequence_in_one_launch')); print(d['configs3'].get('p_frame_ms_hip'), d['configs3'].get('md5_equal'))"
