"""What the compiler made of the kernels, checked where they are built (no GPU): registers, scratch and spills of every kernel in
nrtsearch_amd/libnrtgpu.so, read from the gfx950 code objects (scripts/kernel_resources.py; profiles/rNN_kernel_resources.txt is
its table).  The hot kernels sit on register edges by design (168 VGPRs = 3 waves per SIMD of a 768-thread
workgroup; §4.4: 128 = the 16 waves of the sketch kernel's workgroup): a source or compiler change that pushes one over its edge
shows up as scratch traffic and a lost wave long before a benchmark is read."""
import fnmatch
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "nrtsearch_amd", "libnrtgpu.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# scratch bytes a kernel may use: nothing, except where the table of profiles/ already shows it and DESIGN names it
KNOWN_SCRATCH = {
    # The MaxScore kernel's workgroups are persistent since round 4 (one per CU, round after round): a handful of values that live
    # from one round to the next (thread / wave ids, the launch record's pointer halves) are parked in scratch at the round's head
    # and fetched in its epilogue -- NOT inside the walk, which test_the_maxscore_walk_touches_no_scratch pins.  Same box, same
    # run: identical launch times with and without them (profiles/r04_persistent_spare_ab.log: "record" vs "kargs").
    "bm25_maxscore_kernel<false, false, 2>": 112,     # SHAPES == 2 (tie breaker / MUST + SHOULD): a second accumulator per posting slot,
    "bm25_maxscore_kernel<false, true, 2>": 144,      #   sixteen registers the kernel does not have -- spills INSIDE the walk, these shapes only
    "bm25_maxscore_kernel<false, false, *>": 48,      # (round 5: +16 B for the scattered window order's multiplier, at the round's head)
    "bm25_maxscore_kernel<false, true, *>": 96,       # packed postings (round 6: +16 B where the epilogue writes out the keys that still reach theta)
    "bm25_scan_kernel<true, true, 8, false>": 16,     # clause counting on the exhaustive route (COMPLETE mode): 2 VGPRs, outside the loop
}
VGPR_EDGE = {"bm25_maxscore_kernel<*>": 168, "bm25_scan_kernel<*>": 168, "knn_sketch_kernel<*>": 128, "knn_score_kernel": 128,
             "knn_select_kernel<*>": 168, "merge_topk_kernel": 168}


@pytest.fixture(scope="module")
def kernels():
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-readelf")) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("the built library or the LLVM binutils are not here")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "scripts", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = mod.kernels_of(LIB)
    assert len(rows) >= 50, "the code objects of the library were not found"
    return {r["name"]: r for r in rows}


def _allowed(name, table, default):
    for pat, v in table.items():
        if fnmatch.fnmatchcase(name, pat):
            return v
    return default


def test_every_kernel_is_named_and_the_routes_are_all_there(kernels):
    for must in ("bm25_maxscore_kernel<false, false, 0>", "bm25_maxscore_kernel<false, true, 1>", "bm25_maxscore_kernel<false, false, 2>", "bm25_scan_kernel<true, true, 0, false>",
                 "bm25_scan_kernel<true, true, 9, true>", "knn_sketch_kernel<1, 4>", "knn_sketch_kernel<4, 8>", "knn_select_kernel<true>",
                 "knn_score_kernel", "merge_topk_kernel", "hybrid_rescore_kernel", "knn_sketch_build_kernel", "knn_panel_fp16_kernel"):
        assert must in kernels, f"{must} is not in the library (or its name was not understood): {sorted(kernels)[:5]} ..."


def test_no_kernel_uses_scratch_it_is_not_known_to_use(kernels):
    bad = []
    for name, r in kernels.items():
        allowed = _allowed(name, KNOWN_SCRATCH, 0)
        if r["scratch"] > allowed:
            bad.append(f"{name}: {r['scratch']} B of scratch, {r['vgpr_spills']} VGPRs spilled (allowed {allowed})")
        if allowed == 0:
            assert r["vgpr_spills"] == 0, f"{name} spills {r['vgpr_spills']} VGPRs"
    assert not bad, "\n".join(bad)


def test_hot_kernels_keep_their_occupancy(kernels):
    """The workgroup shapes fix the register budget: 12 waves per CU = 3 per SIMD for the BM25 kernels (512 / 3 -> 168 VGPRs),
    16 waves = 4 per SIMD for the sketch kernel (128).  More registers than that and the workgroup does not launch at all
    (__launch_bounds__ would have failed the build); the point here is that nothing drifted BELOW the designed occupancy either."""
    for name, r in kernels.items():
        edge = _allowed(name, VGPR_EDGE, None)
        if edge is None:
            continue
        assert r["vgpr"] + r["agpr"] <= edge, f"{name}: {r['vgpr']} + {r['agpr']} registers, designed for {edge}"
    # the exhaustive BM25 route and the exact-kNN pass at every panel width: not one byte of scratch (the MaxScore kernel: its
    # WALK has none -- test_the_maxscore_walk_touches_no_scratch)
    for name in ("bm25_scan_kernel<true, true, 0, false>", "knn_sketch_kernel<1, 8>", "knn_sketch_kernel<2, 8>", "knn_sketch_kernel<3, 8>",
                 "knn_sketch_kernel<4, 8>", "knn_select_kernel<true>"):
        assert kernels[name]["scratch"] == 0 and kernels[name]["vgpr_spills"] == 0, name


def test_the_maxscore_walk_touches_no_scratch():
    """Where the MaxScore kernel's scratch traffic is: at a round's head and in its epilogue (values that live from one round of
    a persistent workgroup to the next).  The walk -- from the first streamed posting column load (16-byte non-temporal loads)
    to the first workgroup barrier behind it (a candidate-buffer meeting): bounds, window bits, lookups in the later clauses --
    must not hold a single scratch instruction, in any product instantiation."""
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("the built library or the LLVM binutils are not here")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "scripts", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    isa = mod.disassembly_of(LIB, only="bm25_maxscore_kernelILb0E")   # (the mangled name: PROF = false)
    assert len(isa) == 6, sorted(isa)   # (the product library holds no instrumented instantiation: include/nrtgpu_dev.h)
    for name, lines in isa.items():
        if name.endswith(", 2>"):   # (the second-accumulator shapes: sixteen more registers than there are -- DESIGN 4.0)
            continue
        first = next(i for i, l in enumerate(lines) if l.startswith("global_load_dwordx4") and " nt" in l)
        barrier = next(i for i in range(first, len(lines)) if lines[i].startswith("s_barrier"))
        assert barrier - first > 1500, f"{name}: the walk is {barrier - first} instructions?"
        walk = lines[first:barrier]
        assert not [l for l in walk if l.startswith("scratch_")], f"{name}: scratch traffic inside the walk"


def test_the_committed_table_is_the_built_library(kernels):
    """profiles/r06_kernel_resources.txt is this build's table (re-run scripts/kernel_resources.py after a kernel change)."""
    path = os.path.join(ROOT, "profiles", "r06_kernel_resources.txt")
    seen = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        # name (may hold spaces inside <...>), then 9 numeric / a-b columns
        parts = line.rstrip("\n").rsplit(None, 9)
        seen[parts[0].strip()] = (int(parts[1]), int(parts[5]))
    assert set(seen) == set(kernels), sorted(set(seen) ^ set(kernels))
    for name, (vgpr, scratch) in seen.items():
        assert (vgpr, scratch) == (kernels[name]["vgpr"], kernels[name]["scratch"]), f"{name}: table {vgpr, scratch}, library {kernels[name]['vgpr'], kernels[name]['scratch']}"


@pytest.fixture(scope="module")
def sketch_isa():
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("the built library or the LLVM binutils are not here")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "scripts", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    isa = mod.disassembly_of(LIB, only="knn_sketch_kernel")
    assert len(isa) == 8, sorted(isa)
    return isa


def test_the_sketch_kernels_ring_of_row_requests_is_what_the_source_drives_by_hand(sketch_isa):
    """D requests of 16 bytes per lane stay in flight; a slot is refilled the moment its matrix instructions have read
    it, and the only wait inside a group of D pieces is `s_waitcnt vmcnt(D - 1)` -- never a drain.  Left to the compiler this loop
    drained once per group in three different ways; this is the check that it has not found a fourth."""
    for name, lines in sketch_isa.items():
        P, D = (int(x) for x in name[name.index("<") + 1: -1].split(","))
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma_f32_16x16x32_f16")]
        assert len(mf) == P * D, f"{name}: {len(mf)} matrix instructions, one group of {D} pieces x {P} panels expected"
        body = lines[mf[0]: mf[-1] + 1]
        loads = [l for l in body if l.startswith("global_load_dwordx4")]
        waits = [l for l in body if l.startswith("s_waitcnt") and "vmcnt" in l]
        assert len(loads) == D - 1, f"{name}: {len(loads)} refills between the first and the last matrix instruction"
        assert waits and all(w == f"s_waitcnt vmcnt({D - 1})" for w in waits), f"{name}: {sorted(set(waits))}"
        assert not [l for l in body if l.startswith(("scratch_", "flat_", "buffer_", "s_cbranch"))], f"{name}: the group is not straight-line any more"


def test_the_tile_norms_come_through_the_scalar_cache(sketch_isa):
    """(round 3 shipped flat vector loads here, an xfail; the constant-address-space cast was validated on the GPU in round 4:
    profiles/r04_knn_scalar_norms_ab.log)"""
    for name, lines in sketch_isa.items():
        assert not [l for l in lines if l.startswith("flat_load_dwordx4")], f"{name}: 16-byte flat loads (the tile's norms)"
        assert [l for l in lines if l.startswith(("s_load_dwordx16", "s_load_dwordx8"))], name
