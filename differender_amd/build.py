"""In-tree build of libdifferender_hip.so (hipcc, gfx950). Used by __graft_entry__.build()."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdifferender_hip.so")
# compiler and target of every native piece (the Makefile reads the same variables from the environment)
# (DR_HIPCC / DR_ARCH, not HIPCC / ARCH: many build environments export ARCH=x86_64)
HIPCC = os.environ.get("DR_HIPCC", "/opt/rocm/bin/hipcc")
ARCH = os.environ.get("DR_ARCH", "gfx950")
if not ARCH.startswith("gfx"):
    raise RuntimeError(f"DR_ARCH must name an AMD GPU target (gfx950), got {ARCH!r}")


def build(force=False, verbose=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4", f"DR_HIPCC={HIPCC}", f"DR_ARCH={ARCH}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce " + LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True))
