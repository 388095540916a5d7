"""Build liblfa_hip.so with extra compiler flags into image-analysis_amd/variants/liblfa_<name>.so (git-ignored), for
A/B runs on the GPU box:  LFA_LIB_PATH=image-analysis_amd/variants/liblfa_<name>.so python bench.py ...
usage: python tools/build_variant.py <name> [flags...]"""
import importlib.util
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "image-analysis_amd", "_build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(b.HERE, "variants")
obj = os.path.join(out, "obj_" + name)
os.makedirs(obj, exist_ok=True)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def one(s):
    o = os.path.join(obj, s.replace(".hip", ".o"))
    subprocess.run([hipcc] + b.FLAGS + extra + ["-c", os.path.join(b.CSRC, s), "-o", o], check=True)
    return o


with ThreadPoolExecutor(max_workers=7) as ex:
    objs = list(ex.map(one, b.SOURCES))
lib = os.path.join(out, f"liblfa_{name}.so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
