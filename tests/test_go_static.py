"""Static guard for the Go host shim (integration/go/**), which has never met a compiler in this image (no Go toolchain on either box).

What CAN be checked without one:
  1. every package-qualified identifier the files use (`framework.NewPodInfo`, `utils.MakeValidPodsByDaemonset`, `algo.NewAffinityQueue`,
     `corev1.PodSpec`, ...) is DECLARED in the package directory its import path names under the reference tree (vendor included) --
     a renamed or misremembered helper is the most likely way for never-compiled Go to be wrong.  Needs /root/reference: skipped on
     the GPU box, runs in the build container;
  2. every C symbol / constant / struct type the cgo files name (`C.simon_run_batch`, `C.SIMON_MAX_GPU_DEV`, `C.simon_nodes_soa`) exists
     in include/simon_hip.h, and every struct FIELD assigned through a C struct variable exists in that struct;
  3. every Go constant whose comment says it restates a header constant (`// SIMON_SPREAD_DUP_KEY`) has the header's value -- the
     round-4 advisor finding (spreadDupKey = 1 << 14 against 0x40000000) is exactly what this catches.
Test infrastructure only; nothing here is imported by the product."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO_DIR = os.path.join(ROOT, "integration", "go")
HEADER = os.path.join(ROOT, "include", "simon_hip.h")
REF = "/root/reference"
MODULE = "github.com/alibaba/open-simulator"


def go_files():
    out = []
    for d, _, fs in os.walk(GO_DIR):
        out += [os.path.join(d, f) for f in sorted(fs) if f.endswith(".go")]
    return sorted(out)


def strip_go(src):
    """Comments and string / rune literals blanked out (lengths kept), so that identifiers inside them are not mistaken for code."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j])); i = j
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('"' + " " * (j - i - 1) + '"'); i = j + 1
        elif c == "`":
            j = src.find("`", i + 1)
            j = n if j < 0 else j
            out.append("`" + "".join(ch if ch == "\n" else " " for ch in src[i + 1:j]) + "`"); i = j + 1
        elif c == "'":
            j = i + 1
            while j < n and src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            out.append("'" + " " * (j - i - 1) + "'"); i = j + 1
        else:
            out.append(c); i += 1
    return "".join(out)


IMPORT_LINE = re.compile(r'^\s*(?:([A-Za-z_][A-Za-z0-9_]*)\s+)?"([^"]+)"\s*$')


def imports_of(src):
    """alias -> import path, for the non-stdlib, non-cgo imports of one file (raw source: the paths are string literals)."""
    out = {}
    lines = src.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i].strip()
        if ln.startswith("import ("):
            i += 1
            while i < len(lines) and lines[i].strip() != ")":
                m = IMPORT_LINE.match(lines[i].split("//")[0])
                if m:
                    alias, path = m.group(1), m.group(2)
                    out[alias or path.rsplit("/", 1)[-1]] = path
                i += 1
        elif ln.startswith("import "):
            m = IMPORT_LINE.match(ln[len("import "):])
            if m and m.group(2) != "C":
                out[m.group(1) or m.group(2).rsplit("/", 1)[-1]] = m.group(2)
        i += 1
    return out


def package_dir(path):
    if "." not in path.split("/")[0]:
        return None                                            # standard library: not part of the reference tree
    if path == MODULE or path.startswith(MODULE + "/"):
        return os.path.join(REF, path[len(MODULE):].lstrip("/"))
    return os.path.join(REF, "vendor", path)


_DECL_CACHE = {}


def declared_in(pkg_dir):
    """Exported top-level names a package directory declares: func / type / var / const, single or in ( ) blocks, methods excluded."""
    if pkg_dir in _DECL_CACHE:
        return _DECL_CACHE[pkg_dir]
    names = set()
    for f in sorted(os.listdir(pkg_dir)):
        if not f.endswith(".go") or f.endswith("_test.go"):
            continue
        src = strip_go(open(os.path.join(pkg_dir, f), encoding="utf-8", errors="replace").read())
        depth, block = 0, None
        for ln in src.split("\n"):
            s = ln.strip()
            if depth == 0 and block is None:
                m = re.match(r"func\s+([A-Za-z_][A-Za-z0-9_]*)\s*[\(\[]", s)
                if m:
                    names.add(m.group(1))
                m = re.match(r"type\s+([A-Za-z_][A-Za-z0-9_]*)\b", s)
                if m:
                    names.add(m.group(1))
                m = re.match(r"(var|const)\s+([A-Za-z_][A-Za-z0-9_]*(?:\s*,\s*[A-Za-z_][A-Za-z0-9_]*)*)", s)
                if m:
                    names.update(x.strip() for x in m.group(2).split(","))
                m = re.match(r"(var|const|type)\s*\($", s)
                if m:
                    block = m.group(1)
                    continue
            elif block is not None and depth == 0:
                if s == ")":
                    block = None
                    continue
                m = re.match(r"([A-Za-z_][A-Za-z0-9_]*(?:\s*,\s*[A-Za-z_][A-Za-z0-9_]*)*)", s)
                if m:
                    names.update(x.strip() for x in m.group(1).split(","))
            depth += ln.count("{") - ln.count("}")
            depth = max(depth, 0)
    _DECL_CACHE[pkg_dir] = names
    return names


QUALIFIED = re.compile(r"(?<![A-Za-z0-9_\.\)\]])([A-Za-z_][A-Za-z0-9_]*)\.([A-Z][A-Za-z0-9_]*)")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_every_package_qualified_identifier_exists_in_the_reference_tree():
    missing, checked, pkgs = [], 0, set()
    for path in go_files():
        raw = open(path).read()
        imps = imports_of(raw)
        code = strip_go(raw)
        for m in QUALIFIED.finditer(code):
            alias, name = m.group(1), m.group(2)
            if alias not in imps:
                continue
            d = package_dir(imps[alias])
            if d is None:
                continue
            # a local variable that shadows the import alias (`utils := ...`) would make this a field access: none of the files does that
            assert os.path.isdir(d), f"{os.path.relpath(path, ROOT)}: import {imps[alias]!r} has no directory {d}"
            checked += 1
            pkgs.add(imps[alias])
            if name not in declared_in(d):
                line = code.count("\n", 0, m.start()) + 1
                missing.append(f"{os.path.relpath(path, ROOT)}:{line}: {alias}.{name} is not declared in {os.path.relpath(d, REF)}")
    assert not missing, "\n".join(sorted(set(missing)))
    assert checked > 120 and len(pkgs) >= 12, (checked, sorted(pkgs))


def header_text():
    return open(HEADER).read()


def header_structs():
    """struct name -> set of field names, for the typedef'd structs of include/simon_hip.h."""
    txt = re.sub(r"/\*.*?\*/", " ", header_text(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)?\s*\{(.*?)\}\s*(\w+)\s*;", txt, flags=re.S):
        fields = set()
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                fm = re.search(r"([A-Za-z_]\w*)\s*(?:\[[^\]]*\]\s*)*$", part.strip())
                if fm:
                    fields.add(fm.group(1))
        out[m.group(3)] = fields
    return out


def header_defines():
    out = {}
    for m in re.finditer(r"^#define\s+(SIMON_\w+)\s+(.+?)\s*(?:/\*.*)?$", header_text(), flags=re.M):
        val = m.group(2).strip().strip("()")
        try:
            out[m.group(1)] = int(val.rstrip("uUlL"), 0)
        except ValueError:
            pass
    return out


def test_every_c_symbol_the_cgo_files_name_exists_in_the_header():
    hdr = re.sub(r"/\*.*?\*/", " ", header_text(), flags=re.S)
    funcs = set(re.findall(r"\b(simon_\w+)\s*\(", hdr))
    types = set(header_structs()) | set(re.findall(r"typedef\s+struct\s+\w+\s+(\w+)\s*;", hdr))
    defines = header_defines()
    builtin = {"int", "uint", "char", "long", "size_t", "int32_t", "int64_t", "uint8_t", "uint16_t", "uint32_t", "uint64_t", "double", "float",
               "malloc", "free", "calloc", "memcpy", "memset", "GoString", "CString", "GoBytes", "uchar", "ulong", "longlong", "ulonglong", "schar"}
    missing, n = [], 0
    for path in go_files():
        code = strip_go(open(path).read())
        for m in re.finditer(r"\bC\.([A-Za-z_]\w*)", code):
            name = m.group(1)
            n += 1
            if name in builtin or name in funcs or name in types or name in defines:
                continue
            if name.startswith("struct_") and name[len("struct_"):] in types:
                continue
            line = code.count("\n", 0, m.start()) + 1
            missing.append(f"{os.path.relpath(path, ROOT)}:{line}: C.{name}")
    assert not missing, "not in include/simon_hip.h:\n" + "\n".join(sorted(set(missing)))
    assert n > 100


def go_functions(code):
    """(start offset, text) of every top-level func of a stripped Go source: from `func` in column 0 to the closing brace in column 0."""
    out = []
    for m in re.finditer(r"^func\b.*?^\}", code, flags=re.M | re.S):
        out.append((m.start(), m.group(0)))
    return out


def test_every_field_set_on_a_c_struct_exists_in_that_struct():
    structs = header_structs()
    missing, n = [], 0
    for path in go_files():
        code = strip_go(open(path).read())
        for base, fn in go_functions(code):
            # variables of a C struct type in this function: `var x C.T`, `x := C.T{`, `x = C.T{`, parameters / named results `x C.T`, `x *C.T`
            var_type = {}
            for m in re.finditer(r"\b([A-Za-z_]\w*)\s*(?::=|=)\s*&?\s*C\.(simon_\w+)\s*\{|\bvar\s+([A-Za-z_]\w*)\s+\*?C\.(simon_\w+)\b|[\(,]\s*([A-Za-z_]\w*)\s+\*?C\.(simon_\w+)\b", fn):
                var, ty = (m.group(1), m.group(2)) if m.group(1) else (m.group(3), m.group(4)) if m.group(3) else (m.group(5), m.group(6))
                if ty in structs:
                    var_type.setdefault(var, set()).add(ty)
            for var, tys in var_type.items():
                for m in re.finditer(r"(?<![\w\.])" + re.escape(var) + r"\.([A-Za-z_]\w*)\b", fn):
                    n += 1
                    if not any(m.group(1) in structs[t] for t in tys):
                        line = code.count("\n", 0, base + m.start()) + 1
                        missing.append(f"{os.path.relpath(path, ROOT)}:{line}: {var}.{m.group(1)} is no field of {sorted(tys)}")
            # keys of composite literals: C.T{ key: value, ... }
            for m in re.finditer(r"C\.(simon_\w+)\{", fn):            # (gofmt writes no blank between type and brace; `) C.T {` is a result type + body)
                if m.group(1) not in structs:
                    continue
                depth, k = 1, m.end()
                while k < len(fn) and depth:
                    depth += (fn[k] in "{(") - (fn[k] in "})")
                    k += 1
                body = fn[m.end():k - 1]
                flat, d = [], 0
                for ch in body:                                   # keys sit at nesting depth 0 of the literal
                    d += (ch in "{([") - (ch in "})]")
                    flat.append(ch if d == 0 else " ")
                for km in re.finditer(r"(?:^|,)\s*([A-Za-z_]\w*)\s*:", "".join(flat)):
                    n += 1
                    if km.group(1) not in structs[m.group(1)]:
                        line = code.count("\n", 0, base + m.start()) + 1
                        missing.append(f"{os.path.relpath(path, ROOT)}:{line}: {km.group(1)} is no field of {m.group(1)}")
    assert not missing, "\n".join(sorted(set(missing)))
    assert n > 80, n


def test_go_constants_that_restate_header_constants_hold_the_header_values():
    defines = header_defines()
    seen = 0
    bad = []
    for path in go_files():
        for ln_no, ln in enumerate(open(path).read().split("\n"), 1):
            if "//" not in ln:
                continue
            code, comment = ln.split("//", 1)
            cites = re.findall(r"\bSIMON_[A-Z0-9_]+\b", comment)
            m = re.match(r"\s*([A-Za-z_][\w, ]*?)\s*=\s*(.+?)\s*$", code)
            if not cites or not m or "(" in m.group(2) or '"' in m.group(2):
                continue
            names = [x.strip() for x in m.group(1).split(",")]
            vals = [x.strip() for x in m.group(2).split(",")]
            if len(names) != len(vals) or len(cites) < len(names):
                continue
            for name, val, cite in zip(names, vals, cites):
                if cite not in defines:
                    continue
                try:
                    v = eval(val.replace("<<", "<<"), {"__builtins__": {}})       # integer literals and shifts only
                except Exception:
                    continue
                seen += 1
                if v != defines[cite]:
                    bad.append(f"{os.path.relpath(path, ROOT)}:{ln_no}: {name} = {val} but {cite} = {defines[cite]:#x}")
    assert not bad, "\n".join(bad)
    assert seen >= 5, seen


GO_BUILTINS = {"len", "cap", "make", "new", "append", "copy", "delete", "panic", "recover", "print", "println", "close", "complex", "real", "imag",
               "int", "int8", "int16", "int32", "int64", "uint", "uint8", "uint16", "uint32", "uint64", "uintptr", "float32", "float64", "string", "byte",
               "rune", "bool", "error", "func", "if", "for", "switch", "select", "return", "go", "defer", "range", "case", "else", "map", "chan", "struct", "interface", "var", "const", "type"}


def test_go_files_are_balanced_and_use_what_they_import():
    """Two things a Go compiler refuses outright and a reader easily misses in files that were never compiled: an imported package that the
    file does not use, and unbalanced brackets."""
    problems = []
    for path in go_files():
        raw = open(path).read()
        code = strip_go(raw)
        for o, c in ("()", "[]", "{}"):
            if code.count(o) != code.count(c):
                problems.append(f"{os.path.relpath(path, ROOT)}: {code.count(o)} '{o}' against {code.count(c)} '{c}'")
        # imports incl. the standard library
        aliases = {}
        lines = raw.split("\n")
        i = 0
        while i < len(lines):
            ln = lines[i].strip()
            if ln.startswith("import ("):
                i += 1
                while i < len(lines) and lines[i].strip() != ")":
                    m = IMPORT_LINE.match(lines[i].split("//")[0])
                    if m:
                        aliases[m.group(1) or m.group(2).rsplit("/", 1)[-1]] = m.group(2)
                    i += 1
            elif ln.startswith("import "):
                m = IMPORT_LINE.match(ln[len("import "):])
                if m:
                    aliases[m.group(1) or m.group(2).rsplit("/", 1)[-1]] = m.group(2)
            i += 1
        body = code[code.index("\n", code.rindex("import")):] if "import" in code else code
        for alias, imp in aliases.items():
            if alias in ("_", "."):
                continue
            if not re.search(r"(?<![\w\.])" + re.escape(alias) + r"\.", body):
                problems.append(f"{os.path.relpath(path, ROOT)}: imports {imp!r} as {alias} and never uses it (a compile error in Go)")
    assert not problems, "\n".join(problems)


def test_every_function_the_go_package_calls_is_declared_somewhere_in_it():
    """A bare call `name(...)` inside package hipengine / parity must resolve to a function, type or closure declared in one of the package's
    files (or to a Go builtin): a helper renamed in one file and still called in another is a compile error nobody has seen yet."""
    by_pkg = {}
    for path in go_files():
        by_pkg.setdefault(os.path.dirname(path), []).append(path)
    problems = []
    checked = 0
    for pkg, files in by_pkg.items():
        declared = set()
        codes = {}
        for path in files:
            code = strip_go(open(path).read())
            codes[path] = code
            declared |= set(re.findall(r"^func\s+(?:\([^)]*\)\s*)?([A-Za-z_]\w*)\s*[\(\[]", code, flags=re.M))      # functions and methods
            declared |= set(re.findall(r"^type\s+([A-Za-z_]\w*)\b", code, flags=re.M))
            declared |= set(re.findall(r"^\s*type\s+([A-Za-z_]\w*)\b", code, flags=re.M))                              # types declared inside functions
            declared |= set(re.findall(r"\b([A-Za-z_]\w*)\s*:=\s*func\b", code))                                        # closures
            declared |= set(re.findall(r"\bvar\s+([A-Za-z_]\w*)\s+func\b", code))
            declared |= set(re.findall(r"\b([A-Za-z_]\w*)\s+func\(", code))                                             # parameters of function type
        # a file that declares itself part of a REFERENCE package (parity_test.go is dropped into pkg/simulator by oracle/run_ref.sh) also sees
        # that package's own declarations
        ref_pkgs = {"simulator": os.path.join(REF, "pkg", "simulator")}
        for path in files:
            pm = re.search(r"^package\s+(\w+)", codes[path], flags=re.M)
            if pm and pm.group(1) in ref_pkgs:
                if not os.path.isdir(ref_pkgs[pm.group(1)]):
                    codes[path] = ""                                      # no reference tree here (the GPU box): nothing to resolve against
                else:
                    declared |= declared_in(ref_pkgs[pm.group(1)])
        for path, code in codes.items():
            for m in re.finditer(r"(?<![\w\.\)\]])([A-Za-z_]\w*)\(", code):
                name = m.group(1)
                if name in GO_BUILTINS or name in declared:
                    continue
                prev = code[max(0, m.start() - 6):m.start()]
                if prev.rstrip().endswith("func"):
                    continue
                checked += 1
                line = code.count("\n", 0, m.start()) + 1
                problems.append(f"{os.path.relpath(path, ROOT)}:{line}: {name}( is declared nowhere in {os.path.relpath(pkg, ROOT)}")
    assert not problems, "\n".join(sorted(set(problems))[:40])
