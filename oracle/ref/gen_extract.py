"""oracle/_ref build step (fuse): pull the DEFINITIONS of named functions out of a reference .cpp, unchanged, into a build intermediate.

The host-side translation units of the reference (fuseCut/Fuser.cpp, mvsUtils/MultiViewParams.cpp, mvsUtils/common.cpp) cannot be
compiled whole here: they include OpenImageIO, Boost, Eigen and the SfMData model for their file I/O and scene loading.  The functions
the depth-map filtering step consists of use none of that.  This script reads a reference source where it lies under /root/reference,
finds every definition whose qualified name is in the list (all overloads, unless one is excluded by a substring of its parameter
list), and writes them — text untouched, `#line` directives pointing back at the reference — into oracle/_ref/gen/ (git-ignored).
They are then compiled against stand-in declarations (oracle/ref/fuse_standin.hpp, oracle/ref/shim_host/aliceVision/...).  Reference sources are never copied into the
repository.

    python gen_extract.py <in.cpp> <out.cpp> <header to include> <namespace path, e.g. aliceVision::mvsUtils> name[!excluded-substring] ... [old=new ...]
"""
import re
import sys


def _blank_comments_and_strings(text):
    """same length as text, with comments, string and character literals replaced by spaces (their braces must not be counted)"""
    out = list(text)
    i, n = 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            j = text.find("\n", i)
            j = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            j = n if j < 0 else j + 2
        elif c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            j += 1
        else:
            i += 1
            continue
        for k in range(i, min(j, n)):
            if out[k] != "\n":
                out[k] = " "
        i = j
    return "".join(out)


def find_definitions(text, qualname):
    """yield (start, end, parameter list) of every definition `... qualname(...) [const] { ... }` that starts at the beginning of a line"""
    code = _blank_comments_and_strings(text)
    pat = re.compile(r"^[A-Za-z_][^\n;{}()]*?\b%s\s*\(" % re.escape(qualname), re.M)
    for m in pat.finditer(code):
        i = m.end() - 1
        depth = 0
        while True:  # the parameter list
            c = code[i]
            depth += c == "("
            depth -= c == ")"
            i += 1
            if depth == 0:
                break
        j = i
        while code[j] in " \t\r\nconst":  # trailing `const`, white space
            j += 1
        if code[j] != "{":
            continue  # a declaration or a call, not a definition
        depth = 0
        k = j
        while True:
            c = code[k]
            depth += c == "{"
            depth -= c == "}"
            k += 1
            if depth == 0:
                break
        yield m.start(), k, code[m.end():i]


def main(src, dst, header, namespace, names):
    # `old=new` among the names: a `#define old new` after the include (the stand-in declares the function under the other name)
    defines = [n.split("=") for n in names if "=" in n]
    names = [n for n in names if "=" not in n]
    text = open(src, encoding="utf-8-sig").read()
    chunks = []
    for spec in names:
        name, _, excluded = spec.partition("!")
        found = 0
        for a, b, params in find_definitions(text, name):
            if excluded and excluded in params:
                continue
            prev = text.rfind("\n", 0, a - 1) + 1  # a definition that is a template keeps its `template<...>` line
            if text[prev:a].lstrip().startswith("template"):
                a = prev
            line = text.count("\n", 0, a) + 1
            chunks.append((a, '#line %d "%s"\n%s\n' % (line, src, text[a:b])))
            found += 1
        if not found:
            raise SystemExit("%s: no definition of %s" % (src, name))
    chunks.sort()
    with open(dst, "w") as f:
        f.write('#include %s\n' % header)
        for old, new in defines:
            f.write("#define %s %s\n" % (old, new))
        for ns in namespace.split("::"):
            f.write("namespace %s {\n" % ns)
        for _, c in chunks:
            f.write(c)
        f.write("}" * len(namespace.split("::")) + "\n")
    print("%s: %d definitions -> %s" % (src, len(chunks), dst))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
