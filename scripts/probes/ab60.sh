for L in "" cmlpl_amd/libabl60.so ""  cmlpl_amd/libabl60.so; do
  echo "== lib=${L:-default}"
  CMLPL_LIB=$L CMLPL_ALLOW_STALE=1 CMLPL_BENCH_ALLOW_NONFINITE=1 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --breakdown 2>&1 >/dev/null | grep -A 14 "per-kernel" 
done
