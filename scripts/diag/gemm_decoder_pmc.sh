#!/bin/bash
# Where does a wave of the decoder-shape GEMM (M = 262 144, N = K = 768) spend its cycles?  SQ counters, one group per pass.
out=$GRAFT_REPO_ROOT/gpurun_out/gemm_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|TCP|TCC|TA|TD)_[A-Z0-9_]+" | sort -u > $out/counters_avail.txt
cat > /tmp/gemm_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from livingscenes_amd import ops
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in os.environ.get("SHAPE", "262144,768,768").split(",")]
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
am, wm = ops.rowmax(A), ops.rowmax(W)
for _ in range(4):
    ops.gemm_chain(A, W, a_rowmax=am, w_rowmax=wm, want_rowmax=False)
torch.cuda.synchronize()
PY
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d $out/pass_$i -o p --output-format csv -- python /tmp/gemm_one.py > $out/pass_$i.log 2>&1 || tail -3 $out/pass_$i.log
done
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
tot = {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_h2_kernel" in r["Kernel_Name"] or "gemm_w2" in r["Kernel_Name"]:
            tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        line = f"{k:36s} {sum(tot[k]) / len(tot[k]):16.0f}  ({len(tot[k])} launches)"
        print(line); fh.write(line + "\n")
PY
find $out -name "*.csv" -size +4M -delete
