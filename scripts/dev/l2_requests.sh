#!/bin/bash
# dev: L2 requests (TCC_REQ_sum, TCC_HIT_sum, TCC_MISS_sum) per kernel of one bench step (one step in flight, counters serialise the kernels anyway)
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/l2p
timeout 600 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d /tmp/l2p -o p --output-format csv -- python $R/bench.py --inflight 1 --steps 8 --warmup 2 --no-profile --cpu-instances 0 --no-fma-variant > /tmp/l2p.log 2>&1 || tail -5 /tmp/l2p.log
f=$(find /tmp/l2p -name "*counter_collection.csv" | head -1)
python - "$f" > $R/gpurun_out/l2_requests.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"][:70]
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], )
    if key not in seen: seen.add(key); calls[n] += 1
steps = 10.0   # 2 warm-up + 8 timed
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))
tot = sum(v.get("TCC_REQ_sum", 0) for _, v in rows)
print(f"TCC requests per step (all kernels): {tot / steps / 1e6:.1f} M  (x 64 B = {tot / steps * 64 / 1e9:.2f} GB if every request moved one 64-byte sector)")
for n, v in rows[:40]:
    req = v.get("TCC_REQ_sum", 0)
    print(f"{n:72s} calls/step {calls[n] / steps:5.1f}  req/step {req / steps / 1e6:8.2f} M  hit {v.get('TCC_HIT_sum', 0) / max(req, 1):.2f}  miss {v.get('TCC_MISS_sum', 0) / max(req, 1):.2f}")
PY
cat $R/gpurun_out/l2_requests.txt
