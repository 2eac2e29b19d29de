"""Textual consistency check of the Ada sources under ada/ (no Ada toolchain exists in this image, so they cannot be
compiled here): every name they use must be DECLARED somewhere a GNAT build would find it --

  * in ada/ itself,
  * in the reference's own specs (madarch/*.ads, madarch/support/*.ads), after ada/apply_patches.sh has made its
    three edits, or
  * in the small list below of what comes from outside both trees (the Ada standard library, Interfaces.C, and
    OpenGLAda's GL / GL.Types, which the reference depends on but does not vendor).

and every SELECTED name `P.N` whose prefix is a package known here must name something declared in THAT package (this
is what catches a call of Scenes.Describe when no such subprogram exists).  It is a text check: it knows nothing of
types or overloading.  Usage: python scripts/check_ada_sources.py [reference root]   (exit status 1 on a finding)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RESERVED = set("""abort abs abstract accept access aliased all and array at begin body case constant declare delay delta
digits do else elsif end entry exception exit for function generic goto if in interface is limited loop mod new not null
of or others out overriding package pragma private procedure protected raise range record rem renames requeue return
reverse select separate some subtype synchronized tagged task terminate then type until use when while with xor""".split())

# names from outside ada/ and the reference tree
EXTERNAL = set(n.lower() for n in """
Ada Containers Vectors Vector Append Length Hashed_Maps Count_Type Hash_Type Unchecked_Conversion Unchecked_Deallocation
Strings Unbounded Unbounded_String To_Unbounded_String To_String Null_Unbounded_String
Interfaces C int C_float size_t long unsigned_char chars_ptr New_String Value Free Unsigned_8 Unsigned_32 Integer_32 Shift_Left
System Address Null_Address
Standard Natural Positive Integer Boolean True False String Character Float
Constraint_Error Program_Error
Import Convention External_Name Inline Warnings Off
GL Types Single Singles Ints Int Size UInt Vector3 Matrix3 Vector2 Index_3D X Y Z
""".split())


def strip(text):
    """comments and string / character literals out"""
    out = []
    for line in text.splitlines():
        line = re.sub(r'"(?:[^"]|"")*"', '""', line)
        line = re.sub(r"'.'", "' '", line)
        i = line.find("--")
        out.append(line if i < 0 else line[:i])
    return "\n".join(out)


DECL = [
    r"\b(?:function|procedure)\s+(\"[^\"]+\"|\w+)",
    r"\b(?:type|subtype|package(?:\s+body)?|exception)\s+([\w.]+)",
    r"^\s*([\w, ]+?)\s*:\s*(?:constant\b|aliased\b|in\b|out\b|access\b|not\b|[\w.]+)",   # objects, components, parameters
    r"\(\s*([\w, ]+?)\s*:\s*(?:in\b|out\b|access\b|aliased\b|not\b|[\w.]+)",              # first parameter of a profile
    r";\s*([\w, ]+?)\s*:\s*(?:in\b|out\b|access\b|aliased\b|not\b|[\w.]+)",               # later parameters
    r"\bfor\s+(\w+)\s+(?:in|of)\b",
    r"\bwhen\s+(\w+)\s*:\s*others\b",
]


def declared_names(text):
    names = set()
    t = strip(text)
    for pat in DECL:
        for m in re.finditer(pat, t, re.M):
            for n in re.split(r"[,\s]+", m.group(1)):
                n = n.strip('"').split(".")[-1]
                if re.fullmatch(r"\w+", n) and n.lower() not in RESERVED:
                    names.add(n.lower())
    # enumeration literals and discriminants: type T is (A, B, C) / type T (D : ...) is
    for m in re.finditer(r"\btype\s+\w+\s+is\s*\(([^)]*)\)", t, re.S):
        for n in re.split(r"[,\s]+", m.group(1)):
            if re.fullmatch(r"\w+", n):
                names.add(n.lower())
    # named numbers / several names before a colon anywhere in a line (X_LIT : constant := 0;   X_MOV : ...)
    for m in re.finditer(r"(\w+)\s*:\s*constant\b", t):
        names.add(m.group(1).lower())
    return names


def unit_name(text):
    m = re.search(r"^\s*(?:private\s+)?package\s+(?:body\s+)?([\w.]+)\s+is", strip(text), re.M)
    return m.group(1).lower() if m else None


def load(paths):
    """package (full lower-case name) -> names declared in its spec and body"""
    units = {}
    for p in paths:
        text = open(p, encoding="utf-8", errors="replace").read()
        u = unit_name(text)
        if u:
            units.setdefault(u, set()).update(declared_names(text))
            # nested packages and instantiations: their names are reachable through the unit too
            for m in re.finditer(r"^\s*package\s+(\w+)\s+is\s+new\b", strip(text), re.M):
                units.setdefault(u + "." + m.group(1).lower(), set()).add("*")  # an instance: its names are the generic's
            for m in re.finditer(r"^\s+package\s+(\w+)\s+is\s*$(.*?)^\s+end\s+\1\s*;", strip(text), re.M | re.S):
                units.setdefault(u + "." + m.group(1).lower(), set()).update(declared_names(m.group(2)))
    return units


def check(ref_root, ada_dir=None):
    findings = []
    ada_dir = ada_dir or os.path.join(ROOT, "ada")
    ours = sorted(os.path.join(ada_dir, f) for f in os.listdir(ada_dir) if f.endswith((".ads", ".adb")))
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["bash", os.path.join(ROOT, "ada", "apply_patches.sh"), ref_root, tmp], stdout=subprocess.DEVNULL)
        patched = {f: os.path.join(tmp, f) for f in os.listdir(tmp)}
        ref = []
        for d in (os.path.join(ref_root, "madarch"), os.path.join(ref_root, "madarch", "support")):
            for f in sorted(os.listdir(d)):
                if f.endswith(".ads"):
                    ref.append(patched.get(f, os.path.join(d, f)))
        units = load(ref + ours)
    everything = set().union(*units.values()) | EXTERNAL | set(u.split(".")[-1] for u in units)
    by_last = {}
    for u in units:
        by_last.setdefault(u.split(".")[-1], []).append(u)
    for path in ours:
        text = strip(open(path).read())
        text = re.sub(r"'\s*\w+", "", text)  # attributes
        local = declared_names(open(path).read())
        renames = dict((m.group(1).lower(), m.group(2).lower()) for m in re.finditer(r"package\s+(\w+)\s+renames\s+([\w.]+)", text))
        for m in re.finditer(r"(?<![\w.])([A-Za-z]\w*(?:\s*\.\s*(?:[A-Za-z]\w*|\"[^\"]*\"))*)", text):
            parts = [p.strip().strip('"') for p in m.group(1).split(".")]
            low = [p.lower() for p in parts]
            if low[0] in RESERVED or re.fullmatch(r"\d\w*", parts[0]):
                continue
            if low[0] in renames:
                low = renames[low[0]].split(".") + low[1:]
            line = text.count("\n", 0, m.start()) + 1
            for i, n in enumerate(low):
                if not re.fullmatch(r"\w+", n) or n in RESERVED:
                    continue
                if n not in everything and n not in local:
                    findings.append("%s:%d: `%s` is declared nowhere (in `%s`)" % (os.path.relpath(path, ROOT), line, parts[min(i, len(parts) - 1)], m.group(1)))
                    continue
                if i == 0:
                    continue
                # P.N with P a package known here: N must be one of P's names (or a child / nested package of P)
                cands = [u for u in by_last.get(low[i - 1], []) if i == 1 or u.endswith(".".join(low[:i])) or len(by_last[low[i - 1]]) == 1]
                if cands and low[i - 1] not in local - set(by_last):
                    ok = any(n in units[u] or "*" in units[u] or (u + "." + n) in units for u in cands)
                    if not ok and low[i - 1] not in EXTERNAL:
                        findings.append("%s:%d: package `%s` declares no `%s` (in `%s`)" % (os.path.relpath(path, ROOT), line, parts[i - 1] if i - 1 < len(parts) else low[i - 1], n, m.group(1)))
    return findings


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    if not os.path.isdir(os.path.join(ref, "madarch")):
        print("no reference tree at %s: nothing checked" % ref)
        sys.exit(0)
    bad = check(ref)
    for b in bad:
        print(b)
    print("%d finding(s) in ada/" % len(bad))
    sys.exit(1 if bad else 0)
