"""Per-kernel resource usage of a gfx950 object file / shared library, read from the code object's metadata notes:
VGPRs, AGPRs, spilled VGPRs, scratch bytes, LDS bytes, workgroup size.  (`tests/test_capi.py` asserts from this that no
shipped MFMA kernel spills.)

    python tools/kernel_meta.py instant-nsr-pl_amd/csrc/obj/hashgrid.o [name-substring]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    """paths of the gfx950 code objects bundled in `path` (an object file or shared library built by hipcc; a linked
    library holds one clang offload bundle per translation unit, back to back in its .hip_fatbin section)"""
    out = tempfile.mkdtemp(prefix="nsr_co_")
    sec = os.path.join(out, "fatbin.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={sec}", path], capture_output=True)
    if not os.path.exists(sec):
        return []  # host-only object
    blob = open(sec, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m for m in range(len(blob)) if blob.startswith(magic, m)]
    cos = []
    for k, a in enumerate(starts):
        piece = os.path.join(out, f"bundle{k}.bin")
        open(piece, "wb").write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        co = os.path.join(out, f"gfx950_{k}.co")
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={piece}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            cos.append(co)
    return cos


def kernel_meta(path):
    """{demangled-ish kernel name: dict(vgpr, agpr, spill, scratch, lds, wg)}"""
    res = {}
    for co in code_objects(path):
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk
            g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "0"])[1]  # noqa: E731
            name = g("name")
            res[name] = dict(vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), spill=int(g("vgpr_spill_count")),
                             sgpr_spill=int(g("sgpr_spill_count")), scratch=int(g("private_segment_fixed_size")),
                             lds=int(g("group_segment_fixed_size")), wg=int(g("max_flat_workgroup_size")))
    return res


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


if __name__ == "__main__":
    meta = kernel_meta(sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    names = sorted(meta)
    for n, d in zip(names, demangle(names)):
        if sub in d:
            m = meta[n]
            print(f"{d[:110]:110s} vgpr {m['vgpr']:3d} agpr {m['agpr']:3d} spill {m['spill']:3d} scratch {m['scratch']:4d} "
                  f"lds {m['lds']:6d} wg {m['wg']}")
