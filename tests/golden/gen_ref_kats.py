"""Regenerates tests/golden/ref_kats.json from the reference's own Eigen-free translation
units (compiled where they lie under /root/reference by `make -C oracle ref`).

Run in the build container only (the reference checkout does not travel to the GPU box):
    python tests/golden/gen_ref_kats.py
The fixture is DATA (inputs + expected outputs); no reference source is stored here.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if not os.path.isdir("/root/reference/cf_cpu/src"):
        sys.exit("reference checkout not present; fixture cannot be regenerated here")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = os.path.join(ROOT, "tests", "golden", "ref_kats.json")
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ref_kat"), out], stdout=subprocess.DEVNULL)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
