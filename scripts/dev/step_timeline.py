"""Timeline of ONE encode step from a rocprofv3 kernel trace of `bench.py --inflight 1`: every kernel in start order with its duration and
the idle gap since the previous kernel ended on the device (all streams merged) -- where a single step's 2 ms go.
   cd /tmp; rocprofv3 --kernel-trace -d /tmp/tl -o x --output-format csv -- python $REPO/bench.py --inflight 1 --steps 24 --warmup 12 --no-profile --cpu-instances 0 --no-fma-variant
   python scripts/dev/step_timeline.py /tmp/tl [step_index]"""
import csv, glob, sys
root = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
rows = list(csv.DictReader(open(glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
starts = [i for i, e in enumerate(ev) if "prologue_kernel" in e[2]]
a = starts[which]
b = starts[which + 1] if which + 1 < 0 and which + 1 < len(starts) else len(ev)
step = ev[a:b]
t0 = step[0][0]
busy_end = t0
tot_busy = tot_gap = 0
print(f"{'start us':>9s} {'dur us':>8s} {'gap us':>7s}  kernel")
for s, e, n in step:
    gap = max(0, s - busy_end)
    tot_gap += gap
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:7.1f}  {n[:90]}")
    busy_end = max(busy_end, e)
print(f"step span {(busy_end - t0) / 1e3:.1f} us, device idle inside it {tot_gap / 1e3:.1f} us, kernels {len(step)}")
