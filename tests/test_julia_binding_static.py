"""Static check of the two host-side bindings of the C-ABI against include/ilqr_hip.h (CPU only, no library call).

The Julia wrapper iterativelqr.jl_amd/julia/IterativeLQRAMD.jl has never run (no Julia in the image); its `ccall` tuples and its
C-layout structs are hand-written. This test parses them — and the ctypes mirror iterativelqr.jl_amd/_ffi.py — and compares, for
every entry point they bind: the symbol exists in the header, arity, the C type of every argument and of the result; for every
struct that crosses the ABI: field order, field types, offsets and total size under the C layout rules (which are Julia's for an
isbits struct and ctypes' for a Structure). It also pins the literals the Julia file carries (stage numbers, trace width, scalar
slot names). Reference interfaces the bound structs restate: src/options.jl:1-15 (Options), src/solver.jl:28-66 (Solver)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ilqr_hip.h")
JULIA = os.path.join(ROOT, "iterativelqr.jl_amd", "julia", "IterativeLQRAMD.jl")

# ------------------------------------------------------------------------------------------------ the header
SCALARS = {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "size_t": "usize", "double": "f64",
           "char": "char", "void": "void"}
SIZES = {"i32": 4, "i64": 8, "u64": 8, "usize": 8, "f64": 8, "char": 1, "ptr": 8}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_type(decl, structs, fnptrs):
    """canonical type of a C declarator without its name: 'const double*' -> 'ptr:f64'"""
    t = decl.replace("const", " ").replace("struct", " ").strip()
    stars = t.count("*")
    base = t.replace("*", " ").split()
    assert len(base) == 1, decl
    base = base[0]
    if base in fnptrs:
        assert stars == 0
        return "ptr:void"
    if base == "ilqr_handle":
        canon = "void"            # opaque
    elif base in SCALARS:
        canon = SCALARS[base]
    elif base in structs or base == "ilqr_model_vtable":
        canon = "struct:" + base
    else:
        raise AssertionError("unknown C type %r" % decl)
    for _ in range(stars):
        canon = "ptr:" + canon
    return canon


def parse_header():
    text = strip_comments(open(HEADER).read())
    defines = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(\w+)\s+\(?(-?\d+)\)?", text)}
    fnptrs = set(re.findall(r"typedef\s+\w+\s*\(\s*\*\s*(\w+)\s*\)", text))
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        structs[m.group(2)] = None
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(1).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            # 'int32_t nx, nu, nw' / 'const int32_t* dynamics_nx' / 'uint64_t ineq_term[4]'
            first, *rest = [p.strip() for p in stmt.split(",")]
            mm = re.match(r"(.*?)(\w+)\s*(\[\s*(\w+)\s*\])?$", first)
            base_decl = mm.group(1)
            names = [(mm.group(2), mm.group(4))] + [(re.match(r"\**\s*(\w+)", r_).group(1), None) for r_ in rest]
            for nm, arr in names:
                n = None if arr is None else (int(arr) if arr.isdigit() else defines[arr])
                fields.append((nm, c_type(base_decl, structs, fnptrs), n))
        structs[m.group(2)] = fields
    enums = {}
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", text, flags=re.S):
        for item in m.group(1).split(","):
            mm = re.match(r"\s*(\w+)\s*=\s*(\d+)", item)
            if mm:
                enums[mm.group(1)] = int(mm.group(2))
    protos = {}
    body = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    body = re.sub(r"typedef[^;]*;", " ", body)
    body = re.sub(r"^\s*#[^\n]*", " ", body, flags=re.M)            # preprocessor lines
    body = re.sub(r"enum\s*\{.*?\}\s*;", " ", body, flags=re.S)
    body = body.replace('extern "C" {', " ")
    for m in re.finditer(r"([\w\s\*]+?)\b(ilqr_\w+)\s*\(([^)]*)\)\s*;", body):
        ret = c_type(m.group(1), structs, fnptrs)
        args = []
        a = " ".join(m.group(3).split())
        if a and a != "void":
            for part in a.split(","):
                mm = re.match(r"(.*?)(\w+)$", part.strip())
                args.append(c_type(mm.group(1), structs, fnptrs))
        protos[m.group(2)] = (ret, args)
    return structs, protos, enums, defines


def c_layout(fields):
    """offsets and size under the natural-alignment rules (x86-64 SysV = Julia isbits = ctypes)"""
    off, out, amax = 0, [], 1
    for nm, t, n in fields:
        sz = SIZES["ptr" if t.startswith("ptr:") else t]
        off = (off + sz - 1) // sz * sz
        out.append((nm, off))
        off += sz * (n or 1)
        amax = max(amax, sz)
    return out, (off + amax - 1) // amax * amax


# ------------------------------------------------------------------------------------------------ the Julia file
JL_SCALARS = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "UInt64": "u64", "Csize_t": "usize", "Float64": "f64", "UInt8": "char",
              "Cvoid": "void", "Cstring": "ptr:char"}
JL_STRUCTS = {"Options": "ilqr_options", "ProblemDesc": "ilqr_problem_desc", "Stats": "ilqr_stats", "StageKinds": "ilqr_stage_kinds",
              "StagePlan": "ilqr_stage_plan"}


def jl_type(t):
    t = t.strip()
    m = re.match(r"(Ptr|Ref)\{(.*)\}$", t)
    if m:
        return "ptr:" + jl_type(m.group(2))
    if t in JL_SCALARS:
        return JL_SCALARS[t]
    if t in JL_STRUCTS:
        return "struct:" + JL_STRUCTS[t]
    raise AssertionError("unknown Julia type %r" % t)


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [o.strip() for o in out]


def parse_julia():
    text = re.sub(r"#[^\n]*", "", open(JULIA).read())
    text = re.sub(r'""".*?"""', '""', text, flags=re.S)
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*LIB\[\]\),", text):
        # scan the argument list of this ccall with bracket balance
        i, depth, start = m.end(), 1, m.end()
        while depth > 0:
            ch = text[i]
            depth += ch in "({["
            depth -= ch in ")}]"
            i += 1
        parts = split_top(text[start:i - 1])
        ret, tup, actual = parts[0], parts[1], parts[2:]
        assert tup.startswith("(") and tup.endswith(")"), tup
        argt = [a for a in split_top(tup[1:-1]) if a]
        calls.append((m.group(1), ret, argt, actual))
    structs = {}
    for m in re.finditer(r"(?:Base\.@kwdef\s+)?(?:mutable\s+)?struct\s+(\w+)\s*(.*?)\bend\b", text, flags=re.S):
        if m.group(1) not in JL_STRUCTS:
            continue
        fields = []
        for stmt in re.split(r"[;\n]", m.group(2)):
            stmt = stmt.strip()
            if not stmt:
                continue
            mm = re.match(r"(\w+)::([^=]+?)(\s*=.*)?$", stmt)
            assert mm, stmt
            t = mm.group(2).strip()
            nt = re.match(r"NTuple\{(\d+),\s*(\w+)\}$", t)
            if nt:
                fields.append((mm.group(1), jl_type(nt.group(2)), int(nt.group(1))))
            else:
                fields.append((mm.group(1), jl_type(t), None))
        structs[JL_STRUCTS[m.group(1)]] = fields
    return calls, structs, text


# ------------------------------------------------------------------------------------------------ ctypes
def ct_type(t):
    if t is None:
        return "void"
    if t in (C.c_int, C.c_int32):
        return "i32"
    if t is C.c_int64:
        return "i64"
    if t is C.c_uint64:
        return "u64"
    if t is C.c_size_t:
        return "usize"
    if t is C.c_double:
        return "f64"
    if t is C.c_char_p:
        return "ptr:char"
    if t is C.c_void_p:
        return "ptr:*"            # any pointer
    if isinstance(t, type) and issubclass(t, C._Pointer):
        inner = t._type_
        if isinstance(inner, type) and issubclass(inner, C.Structure):
            return "ptr:struct:" + inner.__name__
        return "ptr:" + ct_type(inner)
    raise AssertionError("unknown ctypes type %r" % (t,))


def same(c, other, struct_names=None):
    """C canonical type against a binding's: char buffers are Cstring / Ptr{UInt8} / c_char_p alike; '...ptr:*' is any pointer at
    that depth; size_t is a 64-bit unsigned on this ABI (ctypes.c_size_t IS c_uint64 here)"""
    c, other = c.replace("usize", "u64"), other.replace("usize", "u64")
    if other.endswith("ptr:*"):
        return c.startswith(other[:-1])
    if struct_names and other.startswith("ptr:struct:"):
        other = "ptr:struct:" + struct_names.get(other[len("ptr:struct:"):], other[len("ptr:struct:"):])
    return c == other


H_STRUCTS, H_PROTOS, H_ENUMS, H_DEFINES = parse_header()


def test_header_parses_completely():
    text = strip_comments(open(HEADER).read())
    declared = set(re.findall(r"\b(ilqr_\w+)\s*\(", text)) - {"ilqr_allreduce_sum_fn"}
    assert declared == set(H_PROTOS), declared ^ set(H_PROTOS)
    assert len(H_PROTOS) >= 50
    for name in ("ilqr_options", "ilqr_problem_desc", "ilqr_stats", "ilqr_model_source", "ilqr_stage_kinds", "ilqr_stage_plan"):
        assert H_STRUCTS[name], name


def test_julia_ccalls_match_the_header():
    calls, _, _ = parse_julia()
    assert len(calls) >= 30
    bad = []
    for name, ret, argt, actual in calls:
        if name not in H_PROTOS:
            bad.append("%s: not declared in include/ilqr_hip.h" % name)
            continue
        cret, cargs = H_PROTOS[name]
        if not same(cret, jl_type(ret)):
            bad.append("%s: returns %s in C, %s in Julia" % (name, cret, ret))
        if len(argt) != len(cargs):
            bad.append("%s: %d arguments in C, %d in the ccall tuple" % (name, len(cargs), len(argt)))
            continue
        if len(actual) != len(argt) and not any(a.endswith("...") for a in actual):
            bad.append("%s: %d argument types but %d arguments passed" % (name, len(argt), len(actual)))
        for i, (ca, ja) in enumerate(zip(cargs, argt)):
            if not same(ca, jl_type(ja)):
                bad.append("%s: argument %d is %s in C, %s in Julia" % (name, i + 1, ca, ja))
    assert not bad, "\n".join(bad)


def test_julia_structs_match_the_header_layout():
    _, jstructs, _ = parse_julia()
    assert set(jstructs) == set(JL_STRUCTS.values())
    for name, jf in jstructs.items():
        cf = H_STRUCTS[name]
        assert [f[0] for f in jf] == [f[0] for f in cf], (name, "field order / names")
        for (nm, ct, cn), (_, jt, jn) in zip(cf, jf):
            assert same(ct, jt) and cn == jn, (name, nm, ct, cn, jt, jn)
        assert c_layout(cf) == c_layout(jf), name


def test_ctypes_mirror_matches_the_header():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ilqr_ffi_static", os.path.join(ROOT, "iterativelqr.jl_amd", "_ffi.py"))
    ffi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ffi)
    names = {"Options": "ilqr_options", "ProblemDesc": "ilqr_problem_desc", "Stats": "ilqr_stats", "ModelSource": "ilqr_model_source",
             "StageKinds": "ilqr_stage_kinds", "StagePlan": "ilqr_stage_plan"}
    assert set(ffi.SYMBOLS) == set(H_PROTOS), set(ffi.SYMBOLS) ^ set(H_PROTOS)
    bad = []
    for name, (res, args) in ffi.SYMBOLS.items():
        cret, cargs = H_PROTOS[name]
        if not same(cret, ct_type(res), names):
            bad.append("%s: returns %s in C, %s in ctypes" % (name, cret, ct_type(res)))
        if len(args) != len(cargs):
            bad.append("%s: %d arguments in C, %d in ctypes" % (name, len(cargs), len(args)))
            continue
        for i, (ca, pa) in enumerate(zip(cargs, args)):
            if not same(ca, ct_type(pa), names):
                bad.append("%s: argument %d is %s in C, %s in ctypes" % (name, i + 1, ca, ct_type(pa)))
    assert not bad, "\n".join(bad)
    for pyname, cname in names.items():
        S = getattr(ffi, pyname)
        offs, size = c_layout(H_STRUCTS[cname])
        assert [f[0] for f in S._fields_] == [o[0] for o in offs], cname
        assert [getattr(S, nm).offset for nm, _ in offs] == [o for _, o in offs], cname
        assert C.sizeof(S) == size, cname
    cb = ffi.ALLREDUCE_SUM_FN            # ilqr_allreduce_sum_fn: int (*)(double*, int32_t, void*)
    assert cb._restype_ is C.c_int and [ct_type(a) for a in cb._argtypes_] == ["ptr:f64", "i32", "ptr:*"]
    assert ffi.STAGES == {k[len("ILQR_STAGE_"):].lower(): v for k, v in H_ENUMS.items() if k.startswith("ILQR_STAGE_")}
    assert ffi.MAX_STAGE_KINDS == H_DEFINES["ILQR_MAX_STAGE_KINDS"] and ffi.MODEL_DENSE_TABLES == H_DEFINES["ILQR_MODEL_DENSE_TABLES"]


def test_julia_literals_match_the_library_sources():
    _, _, text = parse_julia()
    # the host-stepped AL loop names its stages by number
    assert re.search(r":ilqr_run_stage[^\n]*s\.handle,\s*7\)", text) and H_ENUMS["ILQR_STAGE_AL_BEGIN"] == 7
    assert re.search(r":ilqr_run_stage[^\n]*s\.handle,\s*8\)", text) and H_ENUMS["ILQR_STAGE_AL_OUTER"] == 8
    api = open(os.path.join(ROOT, "iterativelqr.jl_amd", "csrc", "ilqr_api.hip")).read()
    dev = open(os.path.join(ROOT, "iterativelqr.jl_amd", "csrc", "ilqr_device.hpp")).read()
    for slot in re.findall(r':ilqr_scalar_slot, LIB\[\]\), Cint, \(Cstring,\), "(\w+)"\)', text):
        assert '{"%s", ilqr::S_' % slot in api, slot
    assert '"_scalars"' in text and '{"_scalars", L.scal, ilqr::S_COUNT}' in api
    m = re.search(r"Array\{Float64,3\}\(undef, (\d+), capacity, s\.B\)", text)
    assert m and int(m.group(1)) == int(re.search(r"enum \{ TRACE_W = (\d+) \}", dev).group(1))
    # the callback the shared-step loop hands over: int (*)(double*, int32_t, void*)
    assert "@cfunction($cb, Cint, (Ptr{Float64}, Int32, Ptr{Cvoid}))" in text
    # the prefixes Solver(...) gives the per-kind C sources are the ones the library looks for
    for prefix in ("dynamics_", "cost_stage_", "constraint_stage_", "cost_terminal", "constraint_terminal"):
        assert prefix in text and prefix in api, prefix


@pytest.mark.parametrize("case", ["swapped", "narrowed"])
def test_the_checker_notices_a_mismatch(case):
    """the comparison has teeth: a swapped argument pair and a narrowed integer are both reported"""
    cret, cargs = H_PROTOS["ilqr_run_stage_param"]
    wrong = ["Ptr{Cvoid}", "Float64", "Int32", "Int32"] if case == "swapped" else ["Ptr{Cvoid}", "Int32", "Float64", "Int64"]
    assert not all(same(ca, jl_type(ja)) for ca, ja in zip(cargs, wrong))
