#!/usr/bin/env python
"""Register / LDS / scratch footprint of every gfx950 kernel in the built objects, from the code objects' own
metadata (``llvm-readelf --notes``), as a Markdown table.

    python tools/kernel_resources.py [filter ...] > profiles/kernel_resources.md

rocprofv3's CSV column ``VGPR_Count`` decodes the kernel descriptor's granulated register count with a
granule of 4, the gfx90a+/gfx950 unified register file allocates in granules of 8: a kernel that the compiler
reports at 146 registers shows up as ``(ceil(146 / 8)) * 4 = 76`` there and occupies 152 of the 512 registers
a SIMD lane has (3 waves per SIMD).  This table is the compiler's number.

"spilled VGPRs" counts registers the allocator moved out of the VGPR file.  With 512 registers per lane and a
64-thread workgroup it moves them into AGPRs (``v_accvgpr_write/read``, one cycle, no memory); only a non-zero
"scratch B" means memory.  For every kernel that spills or has scratch, a note under the table counts both kinds
of instruction in the disassembly.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size",
          ".group_segment_fixed_size", ".max_flat_workgroup_size")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines()


def spill_moves(code_object, symbol):
    """(instructions, AGPR moves, scratch accesses) of one kernel's disassembly."""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f"--disassemble-symbols={symbol}", code_object],
                         capture_output=True, text=True).stdout
    body = [ln for ln in dis.splitlines() if "//" in ln]
    return (len(body), sum("v_accvgpr_write" in ln or "v_accvgpr_read" in ln for ln in body),
            sum("scratch_" in ln for ln in body))


def main():
    filters = sys.argv[1:]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(ROOT, "d4pg_pytorch_amd", "csrc", "build", "*.o"))):
            local = os.path.join(tmp, os.path.basename(obj))
            # the objects hold their gfx950 code compressed (build.py: --offload-compress): take the fat binary section
            # and let the offload bundler inflate the code object
            fat, co_path = local + ".fatbin", local + ".gfx950.co"
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat],
                           capture_output=True)
            if not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co_path}"], capture_output=True)
            for co in glob.glob(co_path):
                notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
                for block in notes.split("- .agpr_count:")[1:]:
                    block = ".agpr_count:" + block
                    rec = {}
                    for f in FIELDS + (".name",):
                        m = re.search(re.escape(f) + r":\s+(\S+)", block)
                        rec[f] = m.group(1) if m else "?"
                    rec["obj"] = os.path.basename(obj)
                    if rec[".vgpr_spill_count"] not in ("0", "?") or rec[".private_segment_fixed_size"] not in ("0", "?"):
                        rec["moves"] = spill_moves(co, rec[".name"])
                    rows.append(rec)
    names = demangle([r[".name"] for r in rows])
    notes = []
    print("| kernel | object | VGPRs | AGPRs | SGPRs | spilled VGPRs | scratch B | static LDS B | max workgroup | waves / SIMD by registers |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*\)$", "", n).replace("void ", "")
        if filters and not any(f in n for f in filters):
            continue
        v = int(r[".vgpr_count"]) + int(r[".agpr_count"])
        alloc = max(8, (v + 7) // 8 * 8)
        print(f"| `{n}` | {r['obj']} | {r['.vgpr_count']} | {r['.agpr_count']} | {r['.sgpr_count']} | {r['.vgpr_spill_count']} "
              f"| {r['.private_segment_fixed_size']} | {r['.group_segment_fixed_size']} | {r['.max_flat_workgroup_size']} | {min(8, 512 // alloc)} |")
        if "moves" in r:
            notes.append(f"* `{n}`: {r['.vgpr_spill_count']} spilled VGPRs, {r['.private_segment_fixed_size']} B of scratch: "
                         f"{r['moves'][1]} AGPR moves and {r['moves'][2]} scratch accesses in {r['moves'][0]} instructions.")
    if notes:
        print("\nKernels that spill or use scratch memory (AGPR moves are register-to-register; only scratch accesses touch memory):\n")
        print("\n".join(notes))


if __name__ == "__main__":
    main()
