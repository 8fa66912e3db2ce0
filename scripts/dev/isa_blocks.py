"""dev: static instruction mix per basic block of the kernels of one csrc file whose (mangled) name matches a pattern.
    python scripts/dev/isa_blocks.py knn_mfma.hip knn_seed_kernelILi32ELb0ELb0   [min instructions per block to print]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from livingscenes_amd import build as B
src, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
asm = "/tmp/isa_blocks_%s.s" % src.replace(".hip", "")
flags = B.FLAGS + B.EXTRA_FLAGS.get(src, []) + (["-fno-slp-vectorize"] if src in getattr(B, "PACKED_FP32_GUARD", ()) else [])
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-x", "hip", os.path.join(B.CSRC, src), "--cuda-device-only", "-S", "-o", asm], stderr=subprocess.DEVNULL)
L = open(asm).read().splitlines()
i = 0
while i < len(L):
    m = re.match(r"^(_Z\w+):", L[i])
    if m and re.search(pat, m.group(1)):
        name = m.group(1); blocks = []; cur = ["entry", 0, 0, 0, 0, 0]
        i += 1
        while i < len(L) and "s_endpgm" not in L[i]:
            t = L[i].strip()
            if t.startswith(".LBB"):
                blocks.append(cur); cur = [t.split(":")[0], 0, 0, 0, 0, 0]
            elif t and not t.startswith((";", ".")):
                op = t.split()[0]
                k = 5 if "mfma" in op else 1 if op.startswith("v_") else 2 if op.startswith("s_") else 3 if op.startswith("ds_") else 4 if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else 0
                if k: cur[k] += 1
            i += 1
        blocks.append(cur)
        print(name[:100])
        print("  %-12s %6s %6s %6s %6s %6s" % ("block", "VALU", "SALU", "LDS", "VMEM", "MFMA"))
        for b in blocks:
            if sum(b[1:]) >= minn: print("  %-12s %6d %6d %6d %6d %6d" % tuple(b))
        print("  total VALU %d  MFMA %d" % (sum(b[1] for b in blocks), sum(b[5] for b in blocks)))
    i += 1
