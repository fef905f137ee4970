"""Build recipe for the oracle's C restatement (``oracle/csrc/nerfacc_ref.c`` -> ``oracle/_build/libnsr_oracle.so``).

TEST INFRASTRUCTURE ONLY.  There is no ``oracle/_ref`` target: the reference is pure Python and its
arithmetic lives in un-vendored CUDA packages (tinycudann, nerfacc==0.3.3), so no reference source
can be compiled with gcc here ("unbuildable").
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "nerfacc_ref.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libnsr_oracle.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-o", OUT, SRC, "-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
