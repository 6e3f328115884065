"""Builds libuavtrack.so in-tree with the committed Makefile (hipcc, gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB = os.path.join(HERE, "libuavtrack.so")


def build(force: bool = False, jobs: int = 4, verbose: bool = False) -> str:
    cmd = ["make", "-C", CSRC, f"-j{jobs}"] + (["-B"] if force else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        sys.stderr.write(r.stdout)
    if r.returncode != 0:
        raise RuntimeError(f"building libuavtrack.so failed (exit {r.returncode})")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
