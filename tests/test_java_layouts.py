"""The FFM shim's struct accesses against include/nrtgpu.h ("one field added to nrtgpu_topdocs
silently corrupts every Java caller").  No JDK here, so the Java side is checked as text -- but against what the C compiler
says (scripts/gen_java_layouts.py: gcc's offsetof / sizeof), not against numbers typed from memory:
  * java/.../NrtGpuLayouts.java is what the generator emits from today's header;
  * no Java source carries a literal struct offset (pointer cells read at offset 0 aside);
  * every accessor names an existing NrtGpuLayouts constant and its ValueLayout is as wide as the C field;
  * the StructLayouts of NrtGpu.java (sequential, explicit padding: FFM does not pad) put every member where C does."""
import importlib.util
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "java", "src", "main", "java", "com", "yelp", "nrtsearch", "gpu")
WIDTH = {"JAVA_INT": 4, "JAVA_FLOAT": 4, "JAVA_LONG": 8, "JAVA_DOUBLE": 8, "ADDRESS": 8, "JAVA_BYTE": 1}


@pytest.fixture(scope="module")
def gen():
    if not shutil.which("gcc"):
        pytest.skip("gcc is not here")
    spec = importlib.util.spec_from_file_location("gen_java_layouts", os.path.join(ROOT, "scripts", "gen_java_layouts.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def java_sources():
    return {f: open(os.path.join(JAVA, f)).read() for f in sorted(os.listdir(JAVA)) if f.endswith(".java") and f != "NrtGpuLayouts.java"}


def test_generated_file_is_current(gen):
    assert open(gen.OUT).read() == gen.java_source(gen.layouts()), "run `python scripts/gen_java_layouts.py`: include/nrtgpu.h changed"


def test_no_literal_struct_offsets_in_the_shim():
    bad = []
    for f, text in java_sources().items():
        for m in re.finditer(r"\.(?:get|set)\(\s*(JAVA_\w+|ADDRESS)\s*,\s*(\d+)\b", text):
            if not (m.group(1) == "ADDRESS" and m.group(2) == "0"):    # `out.get(ADDRESS, 0)`: a pointer cell, not a struct
                bad.append((f, m.group(0)))
    assert not bad, f"literal struct offsets (use NrtGpuLayouts.*): {bad}"


def test_accessors_name_existing_fields_of_the_right_width(gen):
    lay = gen.layouts()
    const = {}
    for st, short in gen.STRUCTS.items():
        for field, off, size in lay[st][1]:
            const[f"{short}_{field.upper()}"] = (off, size)
    used = 0
    for f, text in java_sources().items():
        for m in re.finditer(r"\.(?:get|set)\(\s*(JAVA_\w+|ADDRESS)\s*,\s*NrtGpuLayouts\.(\w+)", text):
            assert m.group(2) in const, (f, m.group(0))
            assert WIDTH[m.group(1)] == const[m.group(2)][1], (f, m.group(0), "ValueLayout width != C field size")
            used += 1
        for m in re.finditer(r"NrtGpuLayouts\.(\w+)", text):
            assert m.group(1) in const or m.group(1).endswith("_SIZE"), (f, m.group(0))
    assert used >= 30   # query, term, topdocs, config, diagnostics: the shim does go through the names


def test_struct_layouts_of_the_binding_match_c(gen):
    lay = gen.layouts()
    text = open(os.path.join(JAVA, "NrtGpu.java")).read()
    seen = 0
    for st, short in gen.STRUCTS.items():
        m = re.search(r"static final StructLayout %s\s*=\s*MemoryLayout\.structLayout\((.*?)\);" % short, text, re.S)
        if not m:
            continue
        seen += 1
        off, got = 0, {}
        for mem in re.finditer(r"(JAVA_\w+|ADDRESS)\.withName\(\"(\w+)\"\)|MemoryLayout\.paddingLayout\((\d+)\)", m.group(1)):
            if mem.group(3):
                off += int(mem.group(3))
                continue
            w = WIDTH[mem.group(1)]
            assert off % w == 0, (st, mem.group(2), "member not aligned: FFM rejects the layout")
            got[mem.group(2)] = (off, w)
            off += w
        size, fields = lay[st]
        assert off == size, (st, off, size)
        assert got == {f: (o, s) for f, o, s in fields}, (st, got, fields)
    assert seen >= 5
