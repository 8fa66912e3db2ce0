#!/bin/bash
# dev (round 6): does a monitoring process polling the GPU (the driver samples rocm-smi every ~5 s during its bench run: smi.*.json in BENCH_r05's pulled files)
# disturb a 20-step block?  The 20-step protocol alone, then with rocm-smi polled back to back in the background.   -> gpurun_out/r6/smi_interference.txt
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1: value', round(d['value']), 'blocks_ms', c['blocks_ms'], 'max/min', c['block_max_over_min'])"; }
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-profile --cpu-instances 0 --no-fma-variant --blocks 40"
{
$B 2>/dev/null | line "quiet"
( while true; do rocm-smi --showuse --showmemuse --showpower --showclocks --showtemp --json > /dev/null 2>&1; done ) &
P=$!
sleep 1
$B 2>/dev/null | line "rocm-smi polled back to back"
kill $P; wait $P 2>/dev/null
( while true; do rocm-smi --showuse --showmemuse --showpower --showclocks --showtemp --json > /dev/null 2>&1; sleep 1; done ) &
P=$!
sleep 1
$B 2>/dev/null | line "rocm-smi every ~1.5 s"
kill $P; wait $P 2>/dev/null
( while true; do cat /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/gpu_busy_percent > /dev/null 2>&1; done ) &
P=$!
sleep 1
$B 2>/dev/null | line "sysfs pp_dpm_sclk + gpu_busy_percent read back to back"
kill $P; wait $P 2>/dev/null
$B 2>/dev/null | line "quiet again"
} 2>&1 | tee gpurun_out/r6/smi_interference.txt
