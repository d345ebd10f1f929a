#!/usr/bin/env python3
"""Build-time check of the register budgets the launch-shape code relies on.

launch_shape() (api.cpp) promises 32 resident waves per CU for render_kernel_8 and 28 for
render_kernel_7 because they are compiled to 80 and 96 scalar registers (and at most 64 and 72
vector registers).  A workgroup promised beyond what the registers really admit is accepted by
the launch and never becomes resident, so this script reads the counts the compiler recorded
in the code object of kernels.o and fails the build when a kernel is over its budget.

usage: check_kernel_resources.py kernels.o [llvm-bin-dir]
"""
import os
import re
import subprocess
import sys
import tempfile

BUDGET = {  # kernel name prefix (mangled) -> (max SGPRs, max VGPRs)
    "_ZN2ht15render_kernel_8": (80, 64),
    "_ZN2ht15render_kernel_7": (96, 72),
}


def main():
    obj = sys.argv[1]
    llvm = sys.argv[2] if len(sys.argv) > 2 else "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", co], text=True)
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.groups()
        if key == "agpr_count":  # first key of a kernel's record
            cur = {}
            kernels.append(cur)
        if cur is not None and key in ("name", "sgpr_count", "vgpr_count", "sgpr_spill_count", "vgpr_spill_count"):
            cur[key] = val
    bad, seen = [], 0
    for k in kernels:
        for prefix, (max_s, max_v) in BUDGET.items():
            if k.get("name", "").startswith(prefix):
                seen += 1
                s, v = int(k["sgpr_count"]), int(k["vgpr_count"])
                print(f"  {k['name']}: {s} SGPRs (<= {max_s}), {v} VGPRs (<= {max_v}), "
                      f"{k.get('sgpr_spill_count', '?')} scalars in lanes, {k.get('vgpr_spill_count', '?')} VGPRs spilled")
                if s > max_s or v > max_v:
                    bad.append(k["name"])
    if seen < 10:
        print(f"check_kernel_resources: found {seen} production kernels in {obj}, expected 10", file=sys.stderr)
        return 1
    if bad:
        print("check_kernel_resources: over budget (launch_shape would promise waves that never run): " + ", ".join(bad),
              file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
