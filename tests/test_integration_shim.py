"""INTEGRATION.md §2 shows the C# (P/Invoke) binding a maintainer would add; no .NET SDK exists in this image, so it cannot be compiled.
What CAN be held: that the text is a binding of THIS header -- every [DllImport] names a function include/ocean_waves.h declares, with the
same number of arguments and a return type of the same width, and every [StructLayout] struct lists the fields of its C struct in the same
order with types of the same size (so that LayoutKind.Sequential reproduces the C layout) -- and that §7's index names every export."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "ocean_waves.h")).read()
DOC = open(os.path.join(ROOT, "INTEGRATION.md")).read()
SHIM = DOC.split("## 2. C# (P/Invoke)")[1].split("## 3. Plain C")[0]

C_SIZES = {"float": 4, "double": 8, "int32_t": 4, "uint32_t": 4, "void *": 8, "size_t": 8, "uint64_t": 8}
CS_SIZES = {"float": 4, "double": 8, "int": 4, "uint": 4, "IntPtr": 8, "nuint": 8, "ulong": 8}
STRUCTS = {"OwCascadeParams": "ow_cascade_params", "OwConfig": "ow_config", "OwSurfaceSample": "ow_surface_sample", "OwGroupLink": "ow_group_link",
           "OwGroupConfig": "ow_group_config"}


def strip_comments(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def c_functions():
    """name -> (return type, number of parameters) for every function the header declares"""
    out = {}
    for m in re.finditer(r"^([A-Za-z_][A-Za-z0-9_ \*]*?)\b(ow_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", strip_comments(HEADER), flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        out[name] = (ret, 0 if args in ("", "void") else args.count(",") + 1)
    return out


def c_struct_fields(name):
    body = re.search(r"typedef struct %s \{(.*?)\}\s*%s\s*;" % (name, name), strip_comments(HEADER), flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(void \*|[A-Za-z_][A-Za-z0-9_]*)\s*(.*)$", decl)
        ctype, names = m.group(1), m.group(2)
        for n in names.split(","):
            d = re.match(r"\s*([A-Za-z_][A-Za-z0-9_]*)(\[(\w+)\])?\s*$", n)
            count = d.group(3)
            if count is not None and not count.isdigit():
                count = re.search(r"#define %s (\d+)" % count, HEADER).group(1)
            fields.append((d.group(1), C_SIZES[ctype] * int(count or 1)))
    return fields


def cs_struct_fields(name):
    body = re.search(r"struct %s \{(.*?)\n\}" % name, strip_comments(SHIM), flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.replace("public", "").split())
        if not decl:
            continue
        fixed = decl.startswith("fixed ")
        decl = decl[len("fixed "):] if fixed else decl
        ctype, names = decl.split(" ", 1)
        for n in names.split(","):
            m = re.match(r"\s*([A-Za-z_][A-Za-z0-9_]*)(\[(\d+)\])?\s*$", n)
            fields.append((m.group(1), CS_SIZES[ctype] * int(m.group(3) or 1)))
    return fields


def test_every_dllimport_is_a_function_of_the_header_with_the_same_shape():
    cfun = c_functions()
    imports = re.findall(r"\[DllImport\(Lib\)\]\s*public static extern (\w+) (ow_[a-z0-9_]+)\((.*?)\);", strip_comments(SHIM))
    assert len(imports) >= 30
    for ret, name, args in imports:
        assert name in cfun, f"{name}: not declared in include/ocean_waves.h"
        c_ret, c_n = cfun[name]
        n = 0 if not args.strip() else args.count(",") + 1
        assert n == c_n, f"{name}: {n} arguments in the shim, {c_n} in the header"
        width = {"void": 0, "int": 4, "IntPtr": 8, "double": 8}[ret]
        c_width = 0 if c_ret == "void" else (8 if c_ret.endswith("*") or c_ret == "double" else 4)
        assert width == c_width, f"{name}: returns {ret} in the shim, {c_ret} in the header"
    # the calls the reference's own scene code maps to are all bound
    for must in ("ow_create", "ow_destroy", "ow_update", "ow_process", "ow_cascades_remaining", "ow_set_cascade_params", "ow_get_device_ptrs",
                 "ow_readback_begin", "ow_readback_wait", "ow_last_error"):
        assert must in [i[1] for i in imports], must


def test_every_shim_struct_lists_the_fields_of_its_c_struct():
    for cs, c in STRUCTS.items():
        want, got = c_struct_fields(c), cs_struct_fields(cs)
        assert [f for f, _ in got] == [f for f, _ in want], f"{cs}: fields {got} against {c}: {want}"
        assert [s for _, s in got] == [s for _, s in want], f"{cs}: field sizes {got} against {c}: {want}"
    assert sum(s for _, s in c_struct_fields("ow_cascade_params")) == 128   # no padding: every double sits on a multiple of 8
    assert sum(s for _, s in c_struct_fields("ow_surface_sample")) == 64
    assert sum(s for _, s in c_struct_fields("ow_group_link")) == 32


def test_the_index_names_every_export():
    index = DOC.split("## 7. Index")[1]
    missing = [name for name in c_functions() if "`%s`" % name not in index]
    assert not missing, missing
    assert "(%d;" % len(c_functions()) in index.splitlines()[0]
