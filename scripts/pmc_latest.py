#!/usr/bin/env python3
"""Assemble profiles/pmc_latest.json (HBM bytes per k-NN layer launch, read by bench.py) from a pmc_summary.json.

A layer's k-NN is several launches (centre / bf16 image / seed / sweep / finish ...): their per-launch PMC bytes are summed.
Kernel keys are "<name> @ <blocks> blocks" as written by scripts/pmc_summary.py; the per-layer lists below are the launches
of the B = 64 x 1024-point bench workload.  Usage: python scripts/pmc_latest.py profiles/<round>/pmc_summary.json > profiles/pmc_latest.json
"""
import json, sys

LAYERS = {
    "knn[layer 0]": (["ls::knn_xyz_kernel<false, true> @ 1024 blocks"], "raw-cloud k-NN (wave per query)"),
    "knn[layer 1]": (["ls::knn_mean_rows_kernel @ 128 blocks", "ls::knn_prep_bf16_kernel @ 512 blocks", "ls::knn_seed_kernel<32, false> @ 4096 blocks",
                      "ls::knn_sweep_bf16_kernel<96> @ 1024 blocks", "ls::knn_finish_kernel<32, false> @ 4096 blocks"],
                     "centre + bf16 image + seed + bf16 MFMA sweep + finish"),
    "knn[layer 2]": (["ls::knn_mean_rows_kernel @ 128 blocks", "ls::knn_prep_bf16_kernel @ 512 blocks", "ls::knn_seed_kernel<32, false> @ 2048 blocks",
                      "ls::knn_sweep_bf16_kernel<96> @ 1024 blocks", "ls::knn_finish_kernel<32, false> @ 2048 blocks"],
                     "centre + bf16 image + seed + bf16 MFMA sweep + finish"),
    "knn[layer 3]": (["ls::knn_mean_rows_kernel @ 192 blocks", "ls::knn_prep_bf16_kernel @ 256 blocks",
                      "ls::knn_sweep_winners_kernel<192> @ 512 blocks", "ls::knn_autohint_select_kernel @ 8192 blocks",
                      "ls::knn_seed_kernel<64, false> @ 2048 blocks", "ls::knn_sweep_bf16_kernel<192> @ 1024 blocks",
                      "ls::knn_finish_wave_kernel<64, false> @ 8192 blocks"],
                     "centre + bf16 image + class-winner sweep + hint select + seed + bf16 MFMA sweep + wave-per-query finish"),
    "knn[layer 4]": (["ls::knn_mean_rows_kernel @ 192 blocks", "ls::knn_prep_bf16_kernel @ 256 blocks",
                      "ls::knn_sweep_winners_kernel<192> @ 128 blocks", "ls::knn_autohint_select_kernel @ 2048 blocks",
                      "ls::knn_seed_kernel<64, false> @ 512 blocks", "ls::knn_sweep_bf16_kernel<192> @ 256 blocks",
                      "ls::knn_finish_wave_kernel<64, false> @ 2048 blocks"],
                     "centre + bf16 image + class-winner sweep + hint select + seed + bf16 MFMA sweep + wave-per-query finish"),
}
src = sys.argv[1]
d = json.load(open(src))
out = {"_source": f"{src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace, python bench.py --inflight 1 --steps 4 "
                  "--warmup 1 --cpu-instances 0 --no-profile --no-fma-variant; FETCH_SIZE KB x1024 x2 (gfx950 correction), WRITE_SIZE KB x1024; "
                  "averaged per launch over launches of the same kernel and grid, summed over the launches of a layer's k-NN"}
for layer, (keys, note) in LAYERS.items():
    missing = [k for k in keys if k not in d]
    if missing:
        print(f"{layer}: missing {missing}", file=sys.stderr)
        continue
    out[layer] = {"hbm_read_bytes": sum(d[k]["hbm_read_bytes"] for k in keys), "hbm_write_bytes": sum(d[k]["hbm_write_bytes"] for k in keys),
                  "rocprof_keys": keys, "note": note}
json.dump(out, sys.stdout, indent=1)
