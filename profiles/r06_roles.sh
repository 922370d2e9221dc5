# round 6: which roles share a SIMD (hardware waves w and w + 4), measured again on the round's kernel: JMHIP_MB_ROLES = role of hardware wave k in nibble k
cd $GRAFT_REPO_ROOT
for perm in 76543210 46573210 65473210 47653210 54763210 74653210 75643210; do
  for rep in 1 2; do
  JMHIP_MB_ROLES=$perm python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$perm', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['config']['records_equal_jm'], d['config']['records_equal_picture_after_picture'])"
  done
done
