cd $GRAFT_REPO_ROOT
REAL=$(python tools/build_bench_spec.py --path)
cp $REAL /tmp/real_spec.so
# (whatever happens below, the library under the production name is the production build again when this script ends)
trap 'cp /tmp/real_spec.so $REAL' EXIT
cp variants/spec_FUSED_DEBUG.so $REAL
python tools/fused_debug.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib"
cp /tmp/real_spec.so $REAL
