#!/bin/bash
# The edge certificate A/B on the GPU box: the default library, then the certificate build (variants/spec_CERT.so, made by
# `tools/build_bench_spec.py CERT:cert=1`) swapped in, over batch sizes.  -> gpurun_out/<tag>_cert_*.json
TAG=${1:-run}   # [variant name: CERT (default) / CERTC = MJPL_SPEC_CERT_COARSE=1]
cd $GRAFT_REPO_ROOT
REAL=$(python tools/build_bench_spec.py --path)
python tools/cert_probe.py 262144 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib" > gpurun_out/${TAG}_cert_default_262144.json
cp $REAL /tmp/real_spec.so
# (whatever happens below, the library under the production name is the production build again when this script ends)
trap 'cp /tmp/real_spec.so $REAL' EXIT
cp variants/spec_${2:-CERT}.so $REAL
for E in 262144 1048576 4194304; do
  python tools/cert_probe.py $E 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib" > gpurun_out/${TAG}_cert_build_$E.json
done
cp /tmp/real_spec.so $REAL
python - <<PY
import json
for f in ("cert_default_262144", "cert_build_262144", "cert_build_1048576", "cert_build_4194304"):
    d = json.load(open("gpurun_out/${TAG}_%s.json" % f))
    print(f, {k: ("%.4f ms %.3g e/s cert %d items %d eq %s" % (v["ms"], v["edges_per_s"], v["certified"], v["items"], v["verdicts_equal"])) for k, v in d.items()})
PY
