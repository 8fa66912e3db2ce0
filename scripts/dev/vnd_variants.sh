#!/bin/bash
# dev: gemm_vn_direct_kernel alone under timing variants (LS_VND_SKIP bit mask: 1 = no epilogue, 2 = no MFMAs, 4 = no tiles at all, 8 = no LDS transpose)
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; mkdir -p gpurun_out
python scripts/dev/build_variants.py vnd1:gemm.hip=-DLS_VND_SKIP=1 vnd2:gemm.hip=-DLS_VND_SKIP=2 vnd3:gemm.hip=-DLS_VND_SKIP=3 vnd4:gemm.hip=-DLS_VND_SKIP=4 vnd11:gemm.hip=-DLS_VND_SKIP=11 > /dev/null || exit 1
V=$R/livingscenes_amd/lib/variants
cd /tmp && export TMPDIR=/tmp
for v in release vnd1 vnd2 vnd3 vnd4 vnd11; do
  [ $v = release ] && unset LS_LIB_PATH || export LS_LIB_PATH=$V/$v/liblivingscenes_hip.so
  rm -rf /tmp/vp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/vp -o p --output-format csv -- python $R/scripts/dev/vnd_time.py > /tmp/vp.log 2>&1
  f=$(find /tmp/vp -name "*kernel_stats.csv" | head -1)
  python - "$v" "$f" >> $R/gpurun_out/vnd_variants.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[2])) if any(k in r["Name"] for k in ("gemm_vn", "glob_mean", "gemm_rowmax"))]
print(sys.argv[1], " | ".join(f'{r["Name"][:34]} {float(r["AverageNs"]) / 1e3:.1f} us' for r in rows))
PY
done
cat $R/gpurun_out/vnd_variants.txt
