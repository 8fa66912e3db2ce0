"""Per-kernel, per-grid-size average durations of the k-NN launches in a rocprofv3 kernel trace of bench.py.
Usage (GPU box): cd /tmp; rocprofv3 --kernel-trace -d /tmp/pb -o x --output-format csv -- python $REPO/bench.py --steps 48 ...
                 python $REPO/scripts/knn_kernel_times.py /tmp/pb [substring ...]"""
import collections, csv, glob, sys
root = sys.argv[1]
subs = sys.argv[2:] or ["knn", "row_norms"]
rows = list(csv.DictReader(open(glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if any(s in n for s in subs):
        g = int(r.get("Grid_Size_X") or r.get("Grid_Size"))
        agg[(n[:56], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    print(f"{k[0]:58s} grid {k[1]:8d}  n={len(v):4d}  avg {sum(v)/len(v)/1e3:8.1f} us")
