cd $GRAFT_REPO_ROOT
REAL=$(python tools/build_bench_spec.py --path)
cp $REAL /tmp/real_spec.so
cp variants/spec_FUSED_DEBUG.so $REAL
python tools/fused_debug.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib"
cp /tmp/real_spec.so $REAL
