#!/usr/bin/env python3
"""Build the reference's own native extensions OUT OF TREE (dev container only), for the fixture generators:

    python tests/golden/build_ref_native.py            # -> /tmp/mise_build, /tmp/mcubes_build, /tmp/simplify_build
    python tests/golden/make_golden_mise.py            # tests/golden/mise.npz
    python tests/golden/make_golden_mcubes.py          # tests/golden/mcubes.npz (+ livingscenes_amd/csrc/mc_tables.h)
    python tests/golden/make_golden_simplify.py        # tests/golden/simplify.npz

The sources are compiled WHERE THEY LIE under /root/reference (nothing is copied into the repo, nothing is written to the
reference tree): `cython -o /tmp/...` puts the generated C++ into the scratch directory, g++ compiles it together with the
reference's hand-written .cpp files against this interpreter's headers and numpy.  numpy 2 dropped two macros libmcubes'
pywrapper.cpp still uses: they are aliased on the command line (-DPyArray_DOUBLE=NPY_DOUBLE -DPyArray_ULONG=NPY_ULONG); no
reference header, library or tool is replaced by a stand-in.
"""
import os
import subprocess
import sys
import sysconfig

import numpy as np

REF_UTILS = "/root/reference/lib_shape_prior/core/models/utils/occnet_utils/utils"
EXT = sysconfig.get_config_var("EXT_SUFFIX")
INC = ["-I" + sysconfig.get_paths()["include"], "-I" + np.get_include()]

TARGETS = {
    # name: (scratch dir, module name, pyx, extra C++ sources, extra flags)
    "mise": ("/tmp/mise_build", "mise", "libmise/mise.pyx", [], []),
    "mcubes": ("/tmp/mcubes_build", "mcubes", "libmcubes/mcubes.pyx", ["libmcubes/pywrapper.cpp", "libmcubes/marchingcubes.cpp"],
               ["-DPyArray_DOUBLE=NPY_DOUBLE", "-DPyArray_ULONG=NPY_ULONG"]),
    "simplify": ("/tmp/simplify_build", "simplify_mesh", "libsimplify/simplify_mesh.pyx", [], []),
}


def build(name):
    out_dir, mod, pyx, extra, flags = TARGETS[name]
    os.makedirs(out_dir, exist_ok=True)
    src_dir = os.path.dirname(os.path.join(REF_UTILS, pyx))
    gen = os.path.join(out_dir, mod + ".cpp")
    subprocess.check_call([sys.executable, "-m", "cython", "--cplus", "-3", "-o", gen, os.path.join(REF_UTILS, pyx)])
    so = os.path.join(out_dir, mod + EXT)
    cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++14", "-w", "-DNPY_NO_DEPRECATED_API=0"] + flags + INC + ["-I" + src_dir, gen] + \
          [os.path.join(REF_UTILS, e) for e in extra] + ["-o", so]
    subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(TARGETS)):
        print(n, "->", build(n))
