"""Timing variants of liblivingscenes_hip.so (dev tool): gemm.hip rebuilt with -D<flags>, every other object reused.

    python scripts/dev/build_variants.py NAME[:file.hip[+file2.hip]]=-DFLAG[,-DFLAG2] ...   ->  livingscenes_amd/lib/variants/NAME/liblivingscenes_hip.so
    LS_LIB_PATH=livingscenes_amd/lib/variants/NAME/liblivingscenes_hip.so python bench.py ...

The variants compute WRONG results where a flag removes arithmetic; they exist to price one part of a kernel."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from livingscenes_amd import build as B  # noqa: E402


def main():
    B.build()
    objdir = os.path.join(B.LIBDIR, "obj")
    for spec in sys.argv[1:]:
        name, flags = spec.split("=", 1)
        fnames = ["gemm.hip"]
        if ":" in name:
            name, fl = name.split(":", 1)
            fnames = fl.split("+")      # NAME:a.hip+b.hip=-DFLAG rebuilds several sources with the same flags
        out = os.path.join(B.LIBDIR, "variants", name)
        os.makedirs(out, exist_ok=True)
        rebuilt = {}
        for fname in fnames:
            obj = os.path.join(out, fname.replace(".hip", ".o"))
            extra = B.EXTRA_FLAGS.get(fname, [])
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + extra + [f for f in flags.split(",") if f] + ["-x", "hip", "-c", os.path.join(B.CSRC, fname), "-o", obj])
            rebuilt[os.path.basename(obj)] = obj
        objs = [rebuilt.get(os.path.basename(o), o) for o in (os.path.join(objdir, s.replace(".hip", ".o")) for s in B.SOURCES)]
        objs += [os.path.join(objdir, s.replace(".cpp", ".o")) for s in B.HOST_SOURCES]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "liblivingscenes_hip.so")] + objs)
        print(out)


if __name__ == "__main__":
    main()
