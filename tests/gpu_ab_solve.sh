R=$GRAFT_REPO_ROOT
VS=${VS:-"prev base prev base"}
for v in $VS; do
  echo "== $v"
  GFBE_LIB=$R/ground-fusion2_amd/csrc/variants/libgfbe_$v.so python tests/diag_timing.py 2>&1 | grep -v "k_marg\|k_schur\|k_assemble\|panel\|chol_inv\|amdgpu.ids" | tr '\n' ';' ; echo
  GFBE_LIB=$R/ground-fusion2_amd/csrc/variants/libgfbe_$v.so python tests/diag_single.py 2>&1 | grep resident
done
for v in $VS; do
  echo "== $v"; GFBE_LIB=$R/ground-fusion2_amd/csrc/variants/libgfbe_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline'].get('time_share'))"
done
