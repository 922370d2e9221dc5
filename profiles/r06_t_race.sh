cd $GRAFT_REPO_ROOT
python profiles/r06_race_probe.py 300 2>&1 | tail -8
python profiles/r06_race_probe.py 150 JMHIP_ADAPTER_FLIGHT=2 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_seq.py tests/test_gpu_mbenc.py -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 --no-traffic 2>/dev/null | head -c 260; echo
