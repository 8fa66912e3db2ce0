#!/usr/bin/env python3
"""profiles/pmc_latest.json from the rocprofv3 --pmc passes of scripts/pmc_ops.py (one sub-directory per pass, each with the
counter_collection CSV) + its manifest: per (operator[layer]) HBM read / write bytes, matrix-pipe busy fraction and the
operator's kernels.  FETCH_SIZE is in KB and under-counts wide reads by 2x on gfx950 (MI355X_MICROARCH.md 'HBM'): x1024 x2;
WRITE_SIZE KB x1024.  Matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES (pipe cycles summed over the chip's 1024 SIMDs: 32 per
v_mfma_f32_32x32x16_bf16) / (GRBM_GUI_ACTIVE / 8 x 1024): rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (checked:
value / 8 / hipEvent duration = 2.1-2.2 GHz on the table GEMMs).  valu_issue_frac (round 4) = 4 x SQ_INSTS_VALU / the same denominator: the share of the
chip's non-packed VALU issue slots the operator's kernels use, wave_wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import json
import os
import sys


def dispatches(pass_dir):
    rows = []
    for f in glob.glob(pass_dir + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    by = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        e = by.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "blocks": int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(by.values())


def segments(disp, labels):
    """split at the marker kernels (at::native::...): segment k belongs to labels[k]"""
    segs, cur, started = [], None, False
    for e in disp:
        if e["name"].startswith("void at::native") or e["name"].startswith("at::native"):
            if cur is not None:
                segs.append(cur)
            cur, started = [], True
        elif started:
            cur.append(e)
    if cur is not None:
        segs.append(cur)
    # the set-up phase before the first marker of the op list also contains at::native kernels (gathers): keep the LAST len(labels) segments
    segs = segs[-len(labels):]
    assert len(segs) == len(labels), (len(segs), len(labels))
    return segs


def main(root):
    man = json.load(open(os.path.join(root, "manifest.json")))
    plain = os.path.join(root, "timings_unprofiled.json")
    if os.path.exists(plain):
        man["hipevent_us_last_rep"] = json.load(open(plain))
    labels = man["ops"]
    out = {"_source": f"{root}: rocprofv3 --pmc <counter> --kernel-trace of `python scripts/pmc_ops.py` (B = {man['batch']} x {man['points']} points; every "
                      "operator alone, in order, separated by marker kernels), one pass per counter group; FETCH_SIZE KB x1024 x2 (gfx950), WRITE_SIZE KB x1024; "
                      "per launch of the operator = summed over its kernels, averaged over its repetitions",
           "_hipevent_us": man.get("hipevent_us_last_rep", {})}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    kernels = {}
    for pass_dir in sorted(glob.glob(os.path.join(root, "pass_*"))):
        disp = dispatches(pass_dir)
        if not disp:
            continue
        for lab, seg in zip(labels, segments(disp, labels)):
            tot = collections.defaultdict(float)
            for e in seg:
                for k, v in e.items():
                    if k not in ("name", "blocks"):
                        tot[k] += v
            for k, v in tot.items():
                acc[lab][k].append(v)
            kernels[lab] = [f"{e['name'].split('(')[0].replace('void ', '')} @ {e['blocks']}" for e in seg]
    for lab in dict.fromkeys(labels):
        c = {k: sum(v) / len(v) for k, v in acc[lab].items()}
        e = {"kernels": kernels.get(lab, [])}
        if "FETCH_SIZE" in c:
            e["hbm_read_bytes"] = 2.0 * 1024.0 * c["FETCH_SIZE"]
        if "WRITE_SIZE" in c:
            e["hbm_write_bytes"] = 1024.0 * c["WRITE_SIZE"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]
            e["gui_active_cycles"] = c["GRBM_GUI_ACTIVE"]
            e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "SQ_INSTS_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
            # VALU wave-instructions x 4 cycles (a wave64 instruction on a 16-lane SIMD; transcendental / 64-bit-integer ones take longer, so this
            # is a lower bound of the busy share) over the chip's 1024 SIMDs x the operator's cycles
            e["valu_wave_insts"] = c["SQ_INSTS_VALU"]
            e["valu_issue_frac"] = 4.0 * c["SQ_INSTS_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            if c.get("SQ_WAVE_CYCLES"):
                e["wave_wait_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        kind, layer = lab[:-1].split("[")
        out[f"{kind}[layer {layer}]"] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
