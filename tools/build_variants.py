"""Dev tool: ablation builds of one source file (a phase of a kernel compiled out with -DSSIE_X_*), linked against the other
objects of the in-tree build into tools/_bin/libssie_hip_<name>.so.  Select one at run time with SSIE_HIP_LIB=<path>.

  python tools/build_variants.py conv_fprop_bf16.hip nomfma=-DSSIE_X_NOMFMA nodma=-DSSIE_X_NODMA nostore=-DSSIE_X_NOSTORE
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssie
ssie.load()
from ssie_amd import build as B


def main():
    src = sys.argv[1]
    B.build(verbose=False)
    base = os.path.basename(src)[:-4]
    others = [os.path.join(B.OBJ, f) for f in os.listdir(B.OBJ) if f.endswith(".o") and f != base + ".o"]
    os.makedirs(os.path.join(ROOT, "tools", "_bin"), exist_ok=True)
    for spec in sys.argv[2:]:
        name, flags = spec.split("=", 1)
        obj = os.path.join(ROOT, "tools", "_bin", f"{base}_{name}.o")
        subprocess.check_call([B.hipcc(), *B.FLAGS, *flags.split(","), "-c", os.path.join(B.CSRC, src), "-o", obj])
        lib = os.path.join(ROOT, "tools", "_bin", f"libssie_hip_{name}.so")
        subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *others])
        print("built", lib)


if __name__ == "__main__":
    main()
