"""In-tree build of libdifferender_hip.so (hipcc, gfx950). Used by __graft_entry__.build()."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdifferender_hip.so")
# compiler and target of every native piece (the Makefile reads the same variables from the environment)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = os.environ.get("ARCH", "gfx950")


def build(force=False, verbose=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4", f"HIPCC={HIPCC}", f"ARCH={ARCH}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce " + LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True))
