"""Compile the gfx950 C-ABI library and the `cf_c` pybind11 module in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")


def build(verbose=False, targets=()):
    env = dict(os.environ)
    env.setdefault("PY", sys.executable)
    cmd = ["make", "-C", CSRC, "-j4"] + list(targets)
    res = subprocess.run(cmd, env=env, stdout=None if verbose else subprocess.PIPE,
                         stderr=None if verbose else subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("building heat_amd/csrc failed:\n" + (res.stdout or ""))
    return os.path.join(HERE, "lib", "libheat_cf.so")


if __name__ == "__main__":
    print(build(verbose=True))
