"""Resolves preprocessor conditionals on a given set of macros with FIXED values and drops the dead branches (what `unifdef` does):
    python scripts/tools/strip_switches.py file.hip NAME=0 OTHER=1 -UNDEFINED_NAME ...
Only `#if / #elif / #ifdef / #ifndef` lines whose condition mentions nothing but the given names, integer literals and the operators
! & | == != < > ( ) && || are touched; everything else -- including code inside kept branches -- is copied verbatim.  `#ifndef NAME` /
`#define NAME v` / `#endif` default-definition blocks of the given names are removed too.  Used in round 4 to take the timing-experiment
switches out of the production kernels once their numbers were on record (profiles/r0N_experiments.md)."""
import re
import sys


def main():
    path = sys.argv[1]
    vals, undef = {}, set()
    for a in sys.argv[2:]:
        if a.startswith("-U"):
            undef.add(a[2:])
        else:
            k, v = a.split("=")
            vals[k] = int(v)
    names = set(vals) | undef
    tok = re.compile(r"[A-Za-z_][A-Za-z0-9_]*")

    def known(cond):
        cond = re.sub(r"//.*$", "", cond).strip()
        cond = re.sub(r"/\*.*?\*/", "", cond)
        ids = set(tok.findall(cond)) - {"defined"}
        return bool(ids) and ids <= names and re.fullmatch(r"[A-Za-z0-9_ \t!&|=<>()]+", cond) is not None

    def evaluate(cond):
        cond = re.sub(r"//.*$", "", cond).strip()
        cond = re.sub(r"/\*.*?\*/", "", cond)
        cond = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if m.group(1) in vals else "0", cond)
        cond = tok.sub(lambda m: str(vals.get(m.group(0), 0)), cond)
        cond = cond.replace("&&", " and ").replace("||", " or ")
        cond = re.sub(r"!(?!=)", " not ", cond)
        return bool(eval(cond))       # noqa: S307 (integers and operators only, checked by known())

    lines = open(path).read().split("\n")
    out = []
    # stack entries: dict(ours=bool, emitting=bool, taken=bool, parent_emitting=bool)
    stack = []

    def emitting():
        return all(f["emitting"] for f in stack)

    i = 0
    while i < len(lines):
        ln = lines[i]
        st = ln.strip()
        m_if = re.match(r"#\s*(if|ifdef|ifndef)\b(.*)", st)
        if m_if:
            kind, rest = m_if.group(1), m_if.group(2)
            if kind == "ifndef" and rest.split()[0] in names and i + 2 < len(lines) and re.match(r"#\s*define\s+" + re.escape(rest.split()[0]) + r"\b", lines[i + 1].strip()):
                # default-definition block of one of our names: drop through its #endif (the #define may be followed by comment lines)
                j = i + 2
                while not re.match(r"#\s*endif", lines[j].strip()):
                    j += 1
                i = j + 1
                continue
            if kind == "if" and known(rest):
                v = evaluate(rest)
                stack.append({"ours": True, "emitting": v, "taken": v})
            elif kind in ("ifdef", "ifndef") and rest.split()[0] in names:
                d = rest.split()[0] in vals
                v = d if kind == "ifdef" else not d
                stack.append({"ours": True, "emitting": v, "taken": v})
            else:
                if emitting():
                    out.append(ln)
                stack.append({"ours": False, "emitting": True, "taken": True})
            i += 1
            continue
        m_elif = re.match(r"#\s*elif\b(.*)", st)
        if m_elif and stack:
            f = stack[-1]
            if f["ours"]:
                if not known(m_elif.group(1)):
                    raise SystemExit("%s:%d: #elif of a resolved #if mentions other names" % (path, i + 1))
                v = (not f["taken"]) and evaluate(m_elif.group(1))
                f["emitting"] = v
                f["taken"] = f["taken"] or v
            elif emitting():
                out.append(ln)
            i += 1
            continue
        if re.match(r"#\s*else\b", st) and stack:
            f = stack[-1]
            if f["ours"]:
                f["emitting"] = not f["taken"]
                f["taken"] = True
            elif emitting():
                out.append(ln)
            i += 1
            continue
        if re.match(r"#\s*endif\b", st) and stack:
            f = stack.pop()
            if not f["ours"] and emitting():
                out.append(ln)
            i += 1
            continue
        if emitting():
            out.append(ln)
        i += 1
    if stack:
        raise SystemExit("%s: unbalanced conditionals" % path)
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
