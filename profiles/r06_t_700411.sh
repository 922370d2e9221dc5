# round 6: fuzz_dropin's seed 700411 (one macroblock's chroma of picture 2 different in one run of 21): how often, and with which number of entries
cd $GRAFT_REPO_ROOT
for mode in default 4 2 0; do
  bad=0; n=0
  for k in $(seq 1 ${1:-80}); do
    if [ $mode = default ]; then out=$(python tests/fuzz_dropin.py 0.001 700411 2>&1 | grep -c "^FAILED"); else out=$(JMHIP_ADAPTER_FLIGHT=$mode python tests/fuzz_dropin.py 0.001 700411 2>&1 | grep -c "^FAILED"); fi
    bad=$((bad + out)); n=$((n + 1))
  done
  echo "JMHIP_ADAPTER_FLIGHT=$mode: $n runs, $bad different"
done
