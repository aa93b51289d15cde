"""The Java side (java/src/main/java/com/yelp/nrtsearch/gpu) cannot be compiled in this image (no JDK).  What CAN be checked
without one: every `com.yelp.nrtsearch.server.*` type it imports exists in the reference, and every constructor / method it
calls on such a type exists there WITH THAT ARITY -- in the reference sources as they are, or in the lines added by the
committed server patch (java/patches/nrtsearch-gpu-hook.diff), which must itself apply cleanly to the reference.  Regexes over
the sources are enough for that; the reference is only present in the build container (the GPU boxes skip)."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/main/java"
SHIM = os.path.join(ROOT, "java", "src", "main", "java", "com", "yelp", "nrtsearch", "gpu")
PATCH = os.path.join(ROOT, "java", "patches", "nrtsearch-gpu-hook.diff")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are only present in the build container")


def split_top(s):
    """Top-level comma split of an argument / parameter list (ignores commas inside <>, (), [], {} and string literals)."""
    out, depth, cur, quote = [], 0, "", None
    for i, ch in enumerate(s):
        if quote:
            cur += ch
            if ch == quote and s[i - 1] != "\\":
                quote = None
            continue
        if ch in "\"'":
            quote = ch
        elif ch in "<([{":
            depth += 1
        elif ch in ">)]}":
            if ch == ">" and i > 0 and s[i - 1] == "-":      # the arrow of a lambda
                pass
            else:
                depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def call_args(src, open_paren):
    """The text between the parenthesis at src[open_paren] and its match."""
    depth, i, quote = 0, open_paren, None
    while i < len(src):
        ch = src[i]
        if quote:
            if ch == quote and src[i - 1] != "\\":
                quote = None
        elif ch in "\"'":
            quote = ch
        elif ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                return src[open_paren + 1: i]
        i += 1
    raise AssertionError("unbalanced parentheses")


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", lambda m: " " * len(m.group(0)), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


class RefClass:
    """A reference class as text: the file, plus the lines the server patch adds to it."""

    def __init__(self, fqcn, added):
        self.fqcn = fqcn
        outer = fqcn
        path = os.path.join(REF, *outer.split(".")) + ".java"
        while not os.path.exists(path) and "." in outer:      # nested type: the file of the outer class
            outer = outer.rsplit(".", 1)[0]
            path = os.path.join(REF, *outer.split(".")) + ".java"
        assert os.path.exists(path), f"{fqcn}: no such class in the reference"
        self.path = path
        self.text = strip_comments(open(path).read()) + "\n" + "\n".join(added.get(os.path.relpath(path, REF), []))
        self.simple = fqcn.rsplit(".", 1)[1]

    def arities(self, name):
        """Arities of the methods (or record components, arity 0; or constructors for name == the class) called `name`."""
        found = set()
        for m in re.finditer(r"[\w>\]]\s+%s\s*\(" % re.escape(name), self.text):
            head = self.text[max(0, m.start() - 120): m.start() + 1]
            if re.search(r"\b(new|return|throw|else)\s*$", head.rstrip()[: -1].rstrip()[-8:] if False else ""):
                continue
            args = call_args(self.text, m.end() - 1)
            after = self.text[m.end() + len(args): m.end() + len(args) + 40]
            if not re.match(r"\)\s*(throws [\w., ]+)?\s*[{;]", after):   # a declaration ends in `{` or `;` (interface / abstract)
                continue
            if re.search(r"\b(return|new|throw)\s+%s\s*\($" % re.escape(name), self.text[max(0, m.start() - 20): m.end()]):
                continue
            found.add(len(split_top(args)))
        for m in re.finditer(r"\brecord\s+(\w+)\s*\(", self.text):      # record components are accessors
            comps = split_top(call_args(self.text, m.end() - 1))
            if name in [c.split()[-1] for c in comps]:
                found.add(0)
            if m.group(1) == name:
                found.add(len(comps))
        for m in re.finditer(r"(?:public|protected|private)?\s*%s\s*\(" % re.escape(name), self.text):   # constructors
            if name != self.simple:
                break
            args = call_args(self.text, m.end() - 1)
            after = self.text[m.end() + len(args): m.end() + len(args) + 40]
            if re.match(r"\)\s*(throws [\w., ]+)?\s*\{", after):
                found.add(len(split_top(args)))
        return found

    def superclass(self, known):
        m = re.search(r"\bclass\s+%s(?:<[^{]*?>)?\s+extends\s+(\w+)" % re.escape(self.simple), self.text)
        if not m:
            return None
        sup = m.group(1)
        imp = re.search(r"import\s+(com\.yelp\.nrtsearch\.server\.[\w.]*\.%s)\s*;" % sup, self.text)
        fq = imp.group(1) if imp else self.fqcn.rsplit(".", 1)[0] + "." + sup
        try:
            return known(fq)
        except AssertionError:
            return None      # a Lucene / JDK superclass: outside the reference tree


def patch_additions():
    added = {}
    cur = None
    for line in open(PATCH):
        if line.startswith("+++ "):
            cur = line.split()[1].split("src/main/java/", 1)[1]
            added.setdefault(cur, [])
        elif line.startswith("+") and not line.startswith("+++") and cur:
            added[cur].append(line[1:].rstrip("\n"))
    return added


def test_server_patch_applies_to_the_reference():
    tmp = tempfile.mkdtemp(prefix="nrt_patch_")
    try:
        for line in open(PATCH):
            if line.startswith("--- "):
                rel = line.split()[1].split("/", 1)[1]
                os.makedirs(os.path.dirname(os.path.join(tmp, rel)), exist_ok=True)
                shutil.copy(os.path.join("/root/reference", rel), os.path.join(tmp, rel))
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", PATCH], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_server_patch_adds_no_method_the_reference_already_has():
    """A patch that applies cleanly can still break the build: round 3's added getWrapped() to SearchCutoffWrapper and
    SearchStatsWrapper, which both declare it already (duplicate method = compile error).  Every method a hunk ADDS must be new to
    its file by (name, arity)."""
    decl = re.compile(r"^\s*(?:public|protected|private)\s+(?:static\s+)?(?:<[^>]+>\s+)?[\w<>\[\], ?.]+?\s+(\w+)\s*\(")
    for rel, lines in patch_additions().items():
        ref = strip_comments(open(os.path.join(REF, rel)).read())
        text = "\n".join(lines)
        for i, line in enumerate(lines):
            m = decl.match(line)
            if not m or m.group(1) in ("if", "for", "while", "switch", "return", "new"):
                continue
            name = m.group(1)
            joined = "\n".join(lines[i:])
            args = call_args(joined, joined.index("(", joined.index(name)))
            arity = len(split_top(args))
            for m2 in re.finditer(r"[\w>\]]\s+%s\s*\(" % re.escape(name), ref):
                a2 = call_args(ref, m2.end() - 1)
                after = ref[m2.end() + len(a2): m2.end() + len(a2) + 40]
                if re.match(r"\)\s*(throws [\w., ]+)?\s*\{", after):
                    assert len(split_top(a2)) != arity, f"{rel}: the patch adds {name}/{arity}, which the reference already declares"


def test_every_nrtsearch_type_constructor_and_method_the_shim_uses_exists():
    added = patch_additions()
    cache = {}

    def known(fqcn):
        if fqcn not in cache:
            cache[fqcn] = RefClass(fqcn, added)
        return cache[fqcn]

    def has(cls, name, arity):
        c = cls
        while c is not None:
            if arity in c.arities(name):
                return True
            c = c.superclass(known)
        return False

    checked = []
    for path in sorted(glob.glob(os.path.join(SHIM, "*.java"))):
        src = strip_comments(open(path).read())
        types = {}      # simple name -> RefClass
        for m in re.finditer(r"import\s+(com\.yelp\.nrtsearch\.server\.[\w.]+)\s*;", src):
            cls = known(m.group(1))
            types[cls.simple] = cls
        # nested reference types named through their outer class: Outer.Inner
        for outer in list(types):
            for m in re.finditer(r"\b%s\.([A-Z]\w+)\b" % outer, src):
                inner = m.group(1)
                if re.search(r"\b(class|interface|record|enum)\s+%s\b" % inner, types[outer].text):
                    types[outer + "." + inner] = types[outer]
        # variables of those types: declarations, parameters, pattern matches
        var_type = {}
        for name, cls in types.items():
            if "." in name:
                continue
            for m in re.finditer(r"\b%s(?:<[^>;(){}]*>)?\s+(\w+)\s*[=;,)]" % name, src):
                var_type[m.group(1)] = cls
        file_checks = 0
        # instance calls
        for var, cls in var_type.items():
            for m in re.finditer(r"\b%s\.(\w+)\s*\(" % re.escape(var), src):
                arity = len(split_top(call_args(src, m.end() - 1)))
                assert has(cls, m.group(1), arity), f"{os.path.basename(path)}: {cls.simple}.{m.group(1)}/{arity} is not in the reference (+ patch)"
                file_checks += 1
        # static calls and constructors
        for name, cls in types.items():
            if "." in name:
                continue
            for m in re.finditer(r"(?<![\w.])%s\.(\w+)\s*\(" % name, src):
                args = split_top(call_args(src, m.end() - 1))
                assert has(cls, m.group(1), len(args)), f"{os.path.basename(path)}: static {name}.{m.group(1)}/{len(args)} is not in the reference (+ patch)"
                file_checks += 1
                # a lambda handed to a hook: its arity is the hook method's
                for a in args:
                    lam = re.match(r"\(([^()]*)\)\s*->", a)
                    if lam and m.group(1) == "setSearcherHook":
                        assert has(cls, "newSearcher", len(split_top(lam.group(1)))), "SearcherHook.newSearcher arity"
            for m in re.finditer(r"\bnew\s+%s(?:<[^>]*>)?\s*\(" % name, src):
                arity = len(split_top(call_args(src, m.end() - 1)))
                assert has(cls, name, arity), f"{os.path.basename(path)}: new {name}/{arity} is not in the reference (+ patch)"
                file_checks += 1
        # super(...) of a class that extends a reference class
        ext = re.search(r"\bclass\s+\w+\s+extends\s+(\w+)", src)
        if ext and ext.group(1) in types:
            for m in re.finditer(r"\bsuper\s*\(", src):
                arity = len(split_top(call_args(src, m.end() - 1)))
                assert has(types[ext.group(1)], ext.group(1), arity), f"{os.path.basename(path)}: super/{arity}: no such {ext.group(1)} constructor"
                file_checks += 1
            for m in re.finditer(r"\bsuper\.(\w+)\s*\(", src):   # (IndexSearcher.search: Lucene's, outside the tree -> not checked)
                pass
        checked.append((os.path.basename(path), file_checks))
    total = sum(n for _, n in checked)
    assert total >= 15, checked      # the test looks at something: GpuIndexSearcher / GpuEligibility / GpuPlugin use the reference's API
    by_file = dict(checked)
    assert by_file["GpuIndexSearcher.java"] >= 4 and by_file["GpuEligibility.java"] >= 4 and by_file["GpuPlugin.java"] >= 2


def test_the_round_two_mistakes_are_what_this_test_catches():
    """MyIndexSearcher has no (reader, executor, slicing) constructor and RelevanceCollector had no getTotalHitsThreshold /
    getSearchAfter: the first exists nowhere, the accessors only through the patch."""
    plain = RefClass("com.yelp.nrtsearch.server.search.MyIndexSearcher", {})
    assert 3 not in plain.arities("MyIndexSearcher") and 2 in plain.arities("MyIndexSearcher")
    assert 3 in plain.arities("create") and 4 not in plain.arities("create")
    patched = RefClass("com.yelp.nrtsearch.server.search.MyIndexSearcher", patch_additions())
    assert 4 in patched.arities("create") and 1 in patched.arities("setSearcherHook") and 4 in patched.arities("newSearcher")
    rc = RefClass("com.yelp.nrtsearch.server.search.collectors.RelevanceCollector", {})
    assert not rc.arities("getTotalHitsThreshold") and not rc.arities("getSearchAfter")
    rcp = RefClass("com.yelp.nrtsearch.server.search.collectors.RelevanceCollector", patch_additions())
    assert 0 in rcp.arities("getTotalHitsThreshold") and 0 in rcp.arities("getSearchAfter")
    assert 0 in RefClass("com.yelp.nrtsearch.server.search.collectors.DocCollector", {}).arities("getNumHitsToCollect")


def test_exact_vector_query_accessors():
    """GpuEligibility.vectorShape reads the field and the query vector of an ExactFloatVectorQuery: getField() is the reference's,
    getQueryVector() comes with the patch (the vector is a private field of the nested class)."""
    plain = RefClass("com.yelp.nrtsearch.server.query.vector.ExactVectorQuery", {})
    assert 0 in plain.arities("getField") and not plain.arities("getQueryVector")
    assert re.search(r"class\s+ExactFloatVectorQuery\s+extends\s+ExactVectorQuery", plain.text)
    patched = RefClass("com.yelp.nrtsearch.server.query.vector.ExactVectorQuery", patch_additions())
    assert 0 in patched.arities("getQueryVector")
    shim = open(os.path.join(SHIM, "GpuEligibility.java")).read()
    assert "evq.getField()" in shim and "evq.getQueryVector()" in shim


def test_java_sources_are_lexically_whole():
    """No JDK here: the least a file can promise is that its braces, parentheses and brackets pair up outside comments and
    literals, that it declares the package of its directory and a top-level type named like the file."""
    files = sorted(glob.glob(os.path.join(ROOT, "java", "src", "main", "java", "**", "*.java"), recursive=True)
                   + glob.glob(os.path.join(ROOT, "bench", "lucene", "*.java")))
    assert len(files) >= 9
    for f in files:
        src = open(f).read()
        s = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        s = re.sub(r"//[^\n]*", "", s)
        s = re.sub(r'"(\\.|[^"\\])*"', '""', s)
        s = re.sub(r"'(\\.|[^'\\])'", "''", s)
        for a, b in ("{}", "()", "[]"):
            assert s.count(a) == s.count(b), f"{os.path.relpath(f, ROOT)}: {s.count(a)} '{a}' against {s.count(b)} '{b}'"
        name = os.path.splitext(os.path.basename(f))[0]
        assert re.search(r"\b(class|interface|record|enum)\s+" + name + r"\b", s), f"{os.path.relpath(f, ROOT)}: no top-level type {name}"
        if os.sep + "java" + os.sep + "src" + os.sep in f:
            pkg = os.path.relpath(os.path.dirname(f), os.path.join(ROOT, "java", "src", "main", "java")).replace(os.sep, ".")
            assert re.search(r"^\s*package\s+" + re.escape(pkg) + r"\s*;", s, flags=re.M), f"{os.path.relpath(f, ROOT)}: package {pkg} expected"
