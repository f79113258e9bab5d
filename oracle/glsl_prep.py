#!/usr/bin/env python3
"""glsl_prep.py SRC.glsl DST.inc -- mechanical GLSL -> C++ token rewrite for oracle/glsl_ref.cpp.

TEST INFRASTRUCTURE.  Reads a compute shader of the reference where it lies (under /root/reference)
and writes a C++-compilable fragment into oracle/_ref/ (git-ignored; reference source is never
committed).  Only syntax that C++ cannot parse is touched; every expression, operand order and
constant of the shader is kept verbatim:
  * `#[compute]`, `#version`                      -> dropped
  * floating literals `2.0`, `1e-6`               -> `2.0f`, `1e-6f` (GLSL literals are 32-bit floats)
  * `layout(local_size_x = A, ...) in;`           -> `GLSL_LOCAL_SIZE(A, B, C)`
  * `layout(...) <quals> uniform image2DArray n;` -> `image2DArray n;`
  * `layout(...) <quals> buffer B { T n[]; };`    -> `T *n;`
  * `layout(push_constant) ... { members };`      -> the members as plain globals
  * `void main()`                                 -> `void shader_main()`
  * `case N:` ... `break;`                        -> `case N: {` ... `break; }` (C++ forbids jumping over the
                                                     initialised declarations GLSL allows inside a switch)
`in` parameter qualifiers, `shared`, `restrict`, ... are handled by macros in glsl_shim.h.

glsl_prep.py --extract SRC.gdshader DST.inc [--define NAME]... [--function NAME]... [--block NAME START END EPILOGUE]...
Godot spatial / particle shaders (.gdshader) are not GLSL translation units (shader_type, render_mode, hinted uniforms,
built-ins such as VERTEX / UV / ACTIVE), so they are not converted whole: the named `#define`s, the named function
definitions (verbatim, brace-matched) and the named statement ranges of vertex() / fragment() / process() are copied out.
A --block is the lines from the first one matching regex START to the first later one matching regex END (inclusive),
wrapped as `static void NAME() { ...lines...; EPILOGUE }` -- the epilogue (C++ written by the caller) only copies the
block's local results into globals of the including namespace; the statements themselves are the reference's text.
"""
import re
import sys

FLOAT_LIT = re.compile(r"(?<![\w.])(\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])")


def split_comment(line):
    i = line.find("//")
    return (line, "") if i < 0 else (line[:i], line[i:])


def convert(text):
    # strip block comments (documentation only) so that their words are never rewritten
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    lines = text.split("\n")
    out = []
    i = 0
    while i < len(lines):
        code, comment = split_comment(lines[i])
        s = code.strip()
        if s.startswith("#[compute]") or s.startswith("#version"):
            out.append("")
            i += 1
            continue
        if s.startswith("layout"):
            m = re.match(r"layout\s*\(\s*local_size_x\s*=\s*(.+?),\s*local_size_y\s*=\s*(.+?),\s*local_size_z\s*=\s*(.+?)\)\s*in\s*;", s)
            if m:
                out.append("GLSL_LOCAL_SIZE(%s, %s, %s)" % m.groups())
                i += 1
                continue
            if "{" in s:  # interface block: gather until the closing "};"
                block = [s]
                while "}" not in block[-1]:
                    i += 1
                    block.append(split_comment(lines[i])[0].strip())
                body = " ".join(block)
                inner = body[body.index("{") + 1:body.rindex("}")]
                members = [x.strip() for x in inner.split(";") if x.strip()]
                if "push_constant" in body:
                    for mem in members:
                        out.append(mem + ";")
                elif re.search(r"\bbuffer\b", body):
                    for mem in members:
                        mm = re.match(r"(\w+)\s+(\w+)\s*\[\s*\]", mem)
                        if not mm:
                            raise SystemExit("unsupported buffer member: " + mem)
                        out.append("%s *%s;" % mm.groups())
                else:
                    raise SystemExit("unsupported interface block: " + body)
                i += 1
                continue
            m = re.match(r"layout\s*\((.*?)\)\s*(.*?)\buniform\s+(\w+)\s+(\w+)\s*;", s)
            if m:
                out.append("%s %s;" % (m.group(3), m.group(4)))
                i += 1
                continue
            raise SystemExit("unsupported layout line: " + s)
        code = FLOAT_LIT.sub(lambda m: m.group(1) + "f", code)
        code = re.sub(r"\bvoid\s+main\s*\(\s*\)", "void shader_main()", code)
        code = re.sub(r"\b(case\s+\w+\s*:)", r"\1 {", code)
        code = re.sub(r"\bbreak\s*;", "break; }", code)
        out.append(code)
        i += 1
    return "\n".join(out) + "\n"


def literals(code):
    return FLOAT_LIT.sub(lambda m: m.group(1) + "f", code)


def extract(text, defines, functions, blocks):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    lines = text.split("\n")
    out = []
    for name in defines:
        hit = [ln for ln in lines if re.match(r"\s*#define\s+%s\b" % re.escape(name), ln)]
        if len(hit) != 1:
            raise SystemExit("#define %s: %d matches" % (name, len(hit)))
        out.append(literals(split_comment(hit[0])[0]))
    for name in functions:
        start = [i for i, ln in enumerate(lines) if re.match(r"\w+\s+%s\s*\(" % re.escape(name), ln)]
        if len(start) != 1:
            raise SystemExit("function %s: %d definitions" % (name, len(start)))
        depth, i, seen = 0, start[0], False
        while True:
            code = split_comment(lines[i])[0]
            out.append(literals(code))
            depth += code.count("{") - code.count("}")
            seen = seen or "{" in code
            if seen and depth == 0:
                break
            i += 1
    for name, start_re, end_re, epilogue in blocks:
        first = [i for i, ln in enumerate(lines) if re.search(start_re, ln)]
        if not first:
            raise SystemExit("block %s: start %r not found" % (name, start_re))
        # several places may open with the same statement (e.g. `vec3 displacement = vec3(0);`): the caller picks by ordinal
        m = re.match(r"(.*)#(\d+)$", name)
        ordinal = int(m.group(2)) if m else 0
        name = m.group(1) if m else name
        i = first[ordinal]
        out.append("static void %s() {" % name)
        while True:
            code = split_comment(lines[i])[0]
            out.append(literals(code))
            if i > first[ordinal] and re.search(end_re, lines[i]) or (i == first[ordinal] and start_re == end_re):
                break
            i += 1
            if i >= len(lines):
                raise SystemExit("block %s: end %r not found" % (name, end_re))
        out.append(epilogue)
        out.append("}")
    return "\n".join(out) + "\n"


def main_extract(argv):
    src, dst, rest = argv[0], argv[1], argv[2:]
    defines, functions, blocks = [], [], []
    while rest:
        if rest[0] == "--define":
            defines.append(rest[1]); rest = rest[2:]
        elif rest[0] == "--function":
            functions.append(rest[1]); rest = rest[2:]
        elif rest[0] == "--block":
            blocks.append(tuple(rest[1:5])); rest = rest[5:]
        else:
            raise SystemExit("unknown argument " + rest[0])
    with open(src) as f:
        text = f.read()
    with open(dst, "w") as f:
        f.write("// GENERATED by oracle/glsl_prep.py --extract from %s -- do not commit\n" % src)
        f.write(extract(text, defines, functions, blocks))


if __name__ == "__main__":
    if sys.argv[1] == "--extract":
        main_extract(sys.argv[2:])
        sys.exit(0)
    src, dst = sys.argv[1], sys.argv[2]
    with open(src) as f:
        text = f.read()
    with open(dst, "w") as f:
        f.write("// GENERATED by oracle/glsl_prep.py from %s -- do not commit\n" % src)
        f.write(convert(text))
