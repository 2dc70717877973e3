#!/usr/bin/env python3
"""Dev-only: is the 2-hop key-rows walk bound by STARTING a one-wave workgroup per root (131,072 per launch)?

Builds variants of csrc/walk_rows.hip in which walk_rows_kernel is PERSISTENT -- the kernel's body becomes a per-root lambda that
resident workgroups call for root after root (virtual block ids b, b + grid, ...: same XCD, same order inside an XCD), P workgroups
per CU -- by a text transform of the product source into /tmp (the product file is not touched), and links them into
tools/build/libsubgacc_persist<P>.so.  Compare on one box with tools/ab_lib.sh:

    python tools/walk_persist_probe.py 8 16 24          # here (hipcc cross-compiles)
    tools/ab_lib.sh "- tools/build/libsubgacc_persist8.so tools/build/libsubgacc_persist16.so ..." "collab twitter" 2   # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "surel_plus_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off".split()


def transform(src, per_cu):
    head = "    int64_t i;\n    if (a.worklist) {   // a dense list of the rows to sample"
    assert src.count(head) == 1
    src = src.replace(head, "    auto one_root = [&](const int64_t vbid, const int64_t vgrid) {\n" + head)
    tail = "// returns 1 when this specialised form took the launch"
    k_end = src.rindex("}\n", 0, src.index(tail))
    loop = ("    };\n"
            "    const int64_t vgrid = ((a.worklist && a.work_cap > 0 && a.work_cap < a.n ? a.work_cap : a.n) + kXcds - 1) / kXcds * kXcds;\n"
            "    for (int64_t vb = blockIdx.x; vb < vgrid; vb += gridDim.x) {\n"
            "        one_root(vb, vgrid);\n"
            "        __syncthreads();       // the next root clears the tables this one may still be reading\n"
            "    }\n")
    src = src[:k_end] + loop + src[k_end:]
    a, b = src.index("auto one_root"), src.index("    };\n    const int64_t vgrid")
    body = src[a:b].replace("(int64_t)gridDim.x", "vgrid").replace("gridDim.x", "vgrid").replace("(int64_t)blockIdx.x", "vbid").replace("blockIdx.x", "vbid")
    src = src[:a] + body + src[b:]
    grid = "    const int64_t grid = xcd_grid(a.worklist && a.work_cap > 0 && a.work_cap < a.n ? a.work_cap : a.n);\n"
    assert src.count(grid) == 1
    return src.replace(grid, grid.replace("const int64_t grid", "const int64_t grid_all") +
                       f"    const int64_t grid = grid_all < {256 * per_cu} ? grid_all : {256 * per_cu};      // persistent: {per_cu} workgroups per CU\n")


def main():
    src = open(os.path.join(CSRC, "walk_rows.hip")).read()
    objs = [os.path.join(CSRC, "build", f) for f in os.listdir(os.path.join(CSRC, "build")) if f.endswith(".o") and f != "walk_rows.o"]
    os.makedirs(os.path.join(ROOT, "tools", "build"), exist_ok=True)
    for p in [int(v) for v in sys.argv[1:]] or [16]:
        tmp = os.path.join(CSRC, f"_persist{p}_tmp.hip")       # (beside its headers; removed again below)
        open(tmp, "w").write(transform(src, p))
        try:
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", tmp, "-o", f"/tmp/walk_rows_persist{p}.o"])
        finally:
            os.remove(tmp)
        out = os.path.join(ROOT, "tools", "build", f"libsubgacc_persist{p}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [f"/tmp/walk_rows_persist{p}.o", "-o", out])
        print("built", out)


if __name__ == "__main__":
    main()
