"""Builds the C++ host module tch_geometric/tch_geometric*.so in-tree (g++, pybind11, libtorch).

No GPU or hipcc needed: the host only calls the C ABI of lib/libtchgeo_hip.so."""
import os
import subprocess
import sys
import sysconfig

import torch
from torch.utils import cpp_extension

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "tch_geometric", "tch_geometric" + sysconfig.get_config_var("EXT_SUFFIX"))
SRC = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".cpp"))
LIBDIR = os.path.join(PKG, "lib")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    deps = SRC + [os.path.join(HERE, "host_common.h"), os.path.join(PKG, "..", "include", "tchgeo.h"),
                  os.path.join(LIBDIR, "libtchgeo_hip.so")]
    return all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps if os.path.exists(d))


def main():
    if up_to_date() and "--force" not in sys.argv:
        return
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    inc = []
    for p in cpp_extension.include_paths():
        inc += ["-isystem", p]
    inc += ["-isystem", sysconfig.get_paths()["include"]]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    objs = []
    procs = []
    for s in SRC:
        o = os.path.join(PKG, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DTORCH_EXTENSION_NAME=tch_geometric",
               "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
               # c10's HIP layer (current stream, device guard) -- host-side headers only, nothing is compiled for the GPU
               "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM", "-isystem", "/opt/rocm/include",
               "-Wno-attributes"] + inc + ["-c", s, "-o", o]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("host compile failed")
    link = ["g++", "-shared", "-o", OUT] + objs + ["-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_python",
                                                    "-L" + LIBDIR, "-ltchgeo_hip",
                                                    "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + tlib]
    subprocess.check_call(link)
    print("built", OUT)


if __name__ == "__main__":
    main()
