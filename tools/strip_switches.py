#!/usr/bin/env python3
"""A small unifdef: resolves the preprocessor conditionals of a source file whose outcome is known for a given set of macros and leaves
the rest alone.  Used once to turn the instrumented kernel source into the shipped one (and its reverse diff into
tools/experiments/frames_instrumentation.patch).
    tools/strip_switches.py in.h out.h -DNAME=VALUE ... -UNAME ... [-Uprefix*]
Handles #if / #ifdef / #ifndef / #elif / #else / #endif whose condition mentions only given macros (simple expressions: NAME, !NAME,
defined(NAME), NAME == n, combined with && and ||, or a bare integer); `#ifndef X / #define X v / #endif` default blocks of a given macro
are dropped; a given macro that appears in ordinary code is replaced by its value."""
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    defs, undefs, undef_prefixes = {}, set(), []
    for a in sys.argv[3:]:
        if a.startswith("-D"):
            k, _, v = a[2:].partition("=")
            defs[k] = v if v != "" else "1"
        elif a.startswith("-U"):
            (undef_prefixes.append(a[2:-1]) if a.endswith("*") else undefs.add(a[2:]))

    def known(name):
        return name in defs or name in undefs or any(name.startswith(p) for p in undef_prefixes)

    def is_def(name):
        return name in defs

    def evaluate(expr):
        """True / False, or None when the expression mentions a macro that was not given."""
        e = expr.split("//")[0].strip()
        names = set(re.findall(r"[A-Za-z_]\w*", e)) - {"defined"}
        if not names and not re.fullmatch(r"[\d\s()!&|=<>]+", e):
            return None
        if any(not known(n) for n in names):
            return None
        e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if is_def(m.group(1)) else "0", e)
        e = re.sub(r"defined\s+(\w+)", lambda m: "1" if is_def(m.group(1)) else "0", e)
        e = re.sub(r"[A-Za-z_]\w*", lambda m: defs.get(m.group(0), "0"), e)
        e = e.replace("&&", " and ").replace("||", " or ").replace("!=", "__NE__").replace("!", " not ").replace("__NE__", "!=")
        return bool(eval(e, {"__builtins__": {}}, {}))

    lines = open(src).read().split("\n")
    out = []
    # stack entries: [mode, taken]  mode: 'keep' (unknown condition: lines pass, directives kept), 'on' / 'off' (resolved)
    stack = []
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", ln)
        active = all(s[0] != "off" for s in stack)
        if m:
            kind, rest = m.group(1), m.group(2).strip()
            if kind in ("if", "ifdef", "ifndef"):
                if kind == "ifdef":
                    name = rest.split()[0]
                    val = is_def(name) if known(name) else None
                elif kind == "ifndef":
                    name = rest.split()[0]
                    val = (not is_def(name)) if known(name) else None
                    # default block: #ifndef X / #define X ... / #endif
                    if known(name) and i + 2 < len(lines) and re.match(r"\s*#\s*define\s+%s\b" % re.escape(name), lines[i + 1]) \
                            and re.match(r"\s*#\s*endif", lines[i + 2]):
                        i += 3
                        continue
                else:
                    val = evaluate(rest)
                if val is None:
                    stack.append(["keep", False])
                    if active:
                        out.append(ln)
                else:
                    stack.append(["on" if val else "off", val])
            elif kind == "elif":
                top = stack[-1]
                if top[0] == "keep":
                    if all(s[0] != "off" for s in stack[:-1]):
                        out.append(ln)
                else:
                    if top[1]:
                        top[0] = "off"
                    else:
                        val = evaluate(rest)
                        if val is None:
                            raise SystemExit("%s:%d: #elif with unknown condition after a resolved #if" % (src, i + 1))
                        top[0] = "on" if val else "off"
                        top[1] = val
            elif kind == "else":
                top = stack[-1]
                if top[0] == "keep":
                    if all(s[0] != "off" for s in stack[:-1]):
                        out.append(ln)
                else:
                    top[0] = "off" if top[1] else "on"
                    top[1] = True
            else:
                top = stack.pop()
                if top[0] == "keep" and all(s[0] != "off" for s in stack):
                    out.append(ln)
            i += 1
            continue
        if active:
            if not re.match(r"\s*#\s*define\b", ln):
                for k, v in defs.items():
                    ln = re.sub(r"\b%s\b" % re.escape(k), v, ln)
            out.append(ln)
        i += 1
    if stack:
        raise SystemExit("unbalanced conditionals")
    open(dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
