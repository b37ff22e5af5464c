"""Does the installed AMDGPU backend still miscompile tools/compiler_bugs/structurizecfg_hoisted_phi.ll?

    python tools/compiler_bugs/check_structurizecfg.py        -> prints "present" or "absent", exit code 0 either way

The reduced case is the shape trav_other_kind's accept block had with the tie rule inlined (kernels/trace.h): a block entered
both from the tie check and from the end of the tie rule, whose only instruction is a zero-cost insertelement feeding the join's
phi.  StructurizeCFG (hoistZeroCostElseBlockPhiValues + simplifyHoistedPhis, AMD clang 22.0.0git roc-7.2.0) hoists that
instruction above the tie check and replaces the join's phi by the Flow phi [%rej, %tiedone], [%acc, %check]: a lane that goes
check -> tie -> tiedone -> accept leaves with %rej where the input IR says %acc.  "present" = the structurized IR has no phi left
that takes %acc from the accept block or its Flow successor.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LLC = os.environ.get("LLC", "/opt/rocm/lib/llvm/bin/llc")


def structurized_ir():
    p = subprocess.run([LLC, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O3", os.path.join(HERE, "structurizecfg_hoisted_phi.ll"),
                        "-o", os.devnull, "-print-after=structurizecfg"], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, check=True)
    dumps = p.stderr.split("*** IR Dump After")
    return dumps[-1]


def bug_present(ir):
    # correct output keeps a phi fed by %acc from the accept side; the miscompiled one only has [%rej, %tiedone], [%acc, %check]
    for line in ir.splitlines():
        m = re.search(r"phi <2 x i32> (.*)", line)
        if m and "%acc" in m.group(1) and "%check" not in m.group(1):
            return False
    return True


if __name__ == "__main__":
    print("present" if bug_present(structurized_ir()) else "absent")
    sys.exit(0)
