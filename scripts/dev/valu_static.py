"""dev: static instruction counts per kernel of a host object's gfx950 code (all instructions, VALU = v_*, packed fp32 = v_pk_{mul,add,fma}_f32, scratch use).
    python scripts/dev/valu_static.py a.o [b.o]     (two objects: side by side, kernels matched by name)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from livingscenes_amd import build as B  # noqa: E402


def counts(obj):
    tmp = tempfile.mkdtemp(prefix="ls_valu_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        od = os.path.join(B.llvm_bin(), "llvm-objdump")
        subprocess.run([od, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(tmp) if "amdgcn" in f][0]
        dis = subprocess.run([od, "-d", os.path.join(tmp, dev)], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out, name = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            name = m.group(1)
            out[name] = [0, 0, 0, 0]
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if name and m:
            op = m.group(1)
            c = out[name]
            c[0] += 1
            c[1] += op.startswith("v_") and not op.startswith("v_mfma")
            c[2] += bool(re.match(r"v_pk_(mul|add|fma)_f32", op))
            c[3] += op.startswith("scratch_")
    return out


def short(k):
    k = re.sub(r"^_ZN2ls\d+", "", k)
    return k[:60]


a = counts(sys.argv[1])
b = counts(sys.argv[2]) if len(sys.argv) > 2 else None
print(f"{'kernel':62s} {'insts':>7s} {'valu':>7s} {'pk_f32':>6s} {'scr':>4s}" + ("   |  " + f"{'insts':>7s} {'valu':>7s} {'pk_f32':>6s} {'scr':>4s}  valu ratio" if b else ""))
for k, c in sorted(a.items(), key=lambda kv: -kv[1][1]):
    if c[1] < 50:
        continue
    line = f"{short(k):62s} {c[0]:7d} {c[1]:7d} {c[2]:6d} {c[3]:4d}"
    if b and k in b:
        d = b[k]
        line += f"   |  {d[0]:7d} {d[1]:7d} {d[2]:6d} {d[3]:4d}  {d[1] / max(c[1], 1):.3f}"
    print(line)
