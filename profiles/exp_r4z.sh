# measurement aid (round 4): configs[3] end to end (lencod_hip.exe, 2160p, 8 slices) with the pictures launched ahead of time and without
mkdir -p gpurun_out/r4z
( for n in 2 8; do for f in "" "JMHIP_ADAPTER_FLIGHT=0"; do echo "--- $n pictures $f"; timeout 300 python profiles/host_time_2160p.py $n $f 2>&1 | grep -E "^0|md5|wall|flight"; done; done ) > gpurun_out/r4z/e2e_2160p_flight.txt 2>&1
cat gpurun_out/r4z/e2e_2160p_flight.txt
