#!/usr/bin/env python3
"""What every kernel of the built library costs a compute unit: registers, LDS, scratch, spills and the waves per SIMD
that leaves -- read from the gfx950 code objects inside nrtsearch_amd/libnrtgpu.so (the AMDGPU metadata note the compiler
writes), no GPU needed.

    python scripts/kernel_resources.py [path/to/lib.so] > profiles/rNN_kernel_resources.txt

`tests/test_kernel_resources.py` holds the hot kernels to these numbers (no scratch, no spills, the occupancy DESIGN.md
§4 quotes), so a compiler or source change that pushes one over a register edge fails on the CPU box already."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    if not names:
        return []
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True, check=True).stdout
        return out.split("\n")[: len(names)]
    except (OSError, subprocess.CalledProcessError):   # no binutils here: short() falls back to by_hand() for mangled names
        return list(names)


def by_hand(mangled):
    """c++filt does not know the _Float16 vector types in some signatures: `_ZN6nrtgpu17knn_sketch_kernelILi4ELi8EEEv...` ->
    `knn_sketch_kernel<4, 8>` (nested name, integer / bool template arguments -- all this library uses)."""
    m = re.match(r"_ZN(\d+)", mangled)
    if not m:
        return mangled
    pos = 3 + len(m.group(1)) + int(m.group(1))       # past the namespace
    m = re.match(r"(\d+)", mangled[pos:])
    if not m:
        return mangled
    n = int(m.group(1))
    pos += len(m.group(1))
    name, pos = mangled[pos: pos + n], pos + n
    args = []
    if mangled[pos: pos + 1] == "I":
        pos += 1
        while True:
            m = re.match(r"L([ibjlmxy])(n?\d+)E", mangled[pos:])
            if not m:
                break
            v = m.group(2).replace("n", "-")
            args.append(("true" if v != "0" else "false") if m.group(1) == "b" else v)
            pos += len(m.group(0))
    return name + ("<" + ", ".join(args) + ">" if args else "")


def short(name):
    """`void ns::kernel<a, b>(args...)` -> `kernel<a, b>`"""
    if name.startswith("_Z"):
        return by_hand(name)
    name = re.sub(r"^void\s+", "", name)
    depth, cut = 0, len(name)
    for i, c in enumerate(name):   # the argument list opens at the first "(" outside template brackets
        if c == "<":
            depth += 1
        elif c == ">":
            depth -= 1
        elif c == "(" and depth == 0:
            cut = i
            break
    name = name[:cut]
    return re.sub(r"\(anonymous namespace\)::|\bnrtgpu::", "", name)


def waves_per_simd(vgpr, agpr, lds, wg_size):
    """gfx950: 512 VGPRs per SIMD lane split between the waves in blocks of 8 (arch + acc registers are one file), at most 8
    waves per SIMD; 160 KB of LDS per CU shared by the workgroups resident on its 4 SIMDs."""
    regs = -(-(max(1, vgpr + agpr)) // 8) * 8
    by_regs = min(8, 512 // regs)
    if lds > 0 and wg_size > 0:
        waves_per_wg = -(-wg_size // 64)
        groups = (160 * 1024) // lds
        by_lds = max(0, groups * waves_per_wg) / 4.0
        return min(by_regs, by_lds) if by_lds < by_regs else by_regs
    return by_regs


def disassembly_of(lib, only=None):
    """{short kernel name: [instruction lines]} of the gfx950 code objects in `lib` (`only`: a substring the MANGLED name must hold)."""
    out, mangled = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, f)], text=True, capture_output=True, check=True).stdout
            cur = None
            for line in dis.split("\n"):
                mm = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
                if mm:
                    cur = mm.group(1) if (only is None or only in mm.group(1)) else None
                    if cur:
                        mangled[cur] = []
                elif cur and line.startswith("\t"):
                    mangled[cur].append(line.strip().split("//")[0].strip())
    names = list(mangled)
    for m, d in zip(names, demangle(names)):
        out[short(d)] = mangled[m]
    return out


def kernels_of(lib):
    """[{name, vgpr, agpr, sgpr, lds, scratch, vgpr_spills, sgpr_spills, wg_size, waves}] for every kernel in `lib`."""
    import yaml
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], text=True, capture_output=True,
                                   check=True).stdout
            m = re.search(r"\n\s*---\n(.*?)\n\s*\.\.\.\s*\n", notes, flags=re.S)
            if not m:
                continue
            meta = yaml.safe_load(m.group(1))
            # generic-address ("flat") memory instructions per kernel: a flat load counts on vmcnt AND lgkmcnt and cannot become
            # a scalar load -- a pointer that lost its address space on the way (e.g. rebuilt from integers) shows up here
            flat, cur = {}, None
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, f)], text=True, capture_output=True,
                                 check=True).stdout
            for line in dis.split("\n"):
                mm = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
                if mm:
                    cur = mm.group(1)
                elif cur and re.match(r"\s+flat_(load|store|atomic)", line):
                    flat[cur] = flat.get(cur, 0) + 1
            for k in meta.get("amdhsa.kernels", []):
                rows.append({"mangled": k[".name"], "vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0),
                             "sgpr": k.get(".sgpr_count", 0), "lds": k.get(".group_segment_fixed_size", 0),
                             "scratch": k.get(".private_segment_fixed_size", 0), "vgpr_spills": k.get(".vgpr_spill_count", 0),
                             "sgpr_spills": k.get(".sgpr_spill_count", 0), "wg_size": k.get(".max_flat_workgroup_size", 0),
                             "flat_ops": flat.get(k[".name"], 0)})
    for r, d in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = short(d)
        r["waves"] = waves_per_simd(r["vgpr"], r["agpr"], r["lds"], r["wg_size"])
    return rows


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "nrtsearch_amd", "libnrtgpu.so")
    rows = sorted(kernels_of(lib), key=lambda r: r["name"])
    print(f"# {os.path.relpath(lib, root)}: {len(rows)} kernels (gfx950 code objects)")
    print("# LDS = the static part (the sketch kernel's panel + queue are dynamic: up to the CU's 160 KB, one workgroup per CU);")
    print("# waves/SIMD = what the registers and the static LDS allow (an upper bound where the launch adds dynamic LDS);")
    print("# spills v/s = VGPRs spilled to scratch / SGPRs spilled to VGPR lanes; flat = generic-address memory instructions")
    print(f"# {'kernel':<58} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'LDS B':>7} {'scratch B':>9} {'spills v/s':>10} {'wg':>5} {'waves/SIMD':>10} {'flat':>4}")
    for r in rows:
        print(f"{r['name'][:60]:<60} {r['vgpr']:>4} {r['agpr']:>4} {r['sgpr']:>4} {r['lds']:>7} {r['scratch']:>9} "
              f"{str(r['vgpr_spills']) + '/' + str(r['sgpr_spills']):>10} {r['wg_size']:>5} {r['waves']:>10} {r['flat_ops']:>4}")


if __name__ == "__main__":
    main()
