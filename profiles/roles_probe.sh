# which chain shares a SIMD with which (hardware waves w and w + 4): k_mb_pipe's time for several role assignments (JMHIP_MB_ROLES = role of wave 7 .. wave 0 as hex digits)
cd $GRAFT_REPO_ROOT
for r in 76543210 56743210 67543210 57643210 74563210 75463210 76541230 76542310; do
  echo "roles $r: $(JMHIP_MB_ROLES=$r python profiles/wg_sweep.py 0 2>&1 | grep workgroups)"
done
