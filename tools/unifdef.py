"""Partial preprocessor evaluation (a small `unifdef`): resolve the #if / #ifdef / #ifndef / #elif / #else / #endif blocks whose
conditions mention only macros given on the command line, keep everything else as it is.  Used once in round 6 to delete the
measured-negative experiment switches from csrc/*.h; kept because the next experiment will want it.

  python tools/unifdef.py -DNAME=VALUE ... -UNAME ... file.h [...]      (rewrites the files in place)

A `#ifndef NAME / #define NAME default / #endif` block of a -D macro is removed too (the default lived there), and remaining uses
of a -D macro in ordinary code are NOT substituted (the caller greps for leftovers).
"""
from __future__ import annotations

import re
import sys


def evaluate(expr: str, defs: dict[str, str | None]):
  """-> True / False, or None when the expression mentions a macro we know nothing about."""
  e = re.sub(r"//.*$", "", expr).strip()
  e = re.sub(r"/\*.*?\*/", "", e)

  def rep_defined(m):
    n = m.group(1) or m.group(2)
    if n not in defs:
      return f"__UNKNOWN_{n}__"
    return "1" if defs[n] is not None else "0"

  e = re.sub(r"defined\s*(?:\(\s*(\w+)\s*\)|(\w+))", rep_defined, e)

  def rep_name(m):
    n = m.group(0)
    if n.startswith("__UNKNOWN_") or n in ("and", "or", "not"):
      return n
    if n not in defs:
      return f"__UNKNOWN_{n}__"
    return defs[n] if defs[n] not in (None, "") else ("0" if defs[n] is None else "1")

  e = re.sub(r"\b[A-Za-z_]\w*\b", rep_name, e)
  if "__UNKNOWN_" in e:
    return None
  e = e.replace("&&", " and ").replace("||", " or ")
  e = re.sub(r"!(?!=)", " not ", e)
  return bool(eval(e, {"__builtins__": {}}, {}))  # noqa: S307 (our own headers)


def process(text: str, defs: dict[str, str | None]) -> str:
  out: list[str] = []
  # stack entries: dict(kind="known"|"unknown", taken=bool (a branch already emitted), emitting=bool, parent_emitting=bool)
  stack: list[dict] = []
  lines = text.split("\n")
  i = 0

  def emitting():
    return all(s["emitting"] for s in stack)

  while i < len(lines):
    line = lines[i]
    m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", line)
    if not m:
      if emitting():
        out.append(line)
      i += 1
      continue
    kw, rest = m.group(1), m.group(2)
    if kw in ("if", "ifdef", "ifndef"):
      cond = rest if kw == "if" else (f"defined({rest.split('//')[0].strip()})" if kw == "ifdef" else f"!defined({rest.split('//')[0].strip()})")
      # the default-definition idiom of a macro we pin: #ifndef X / #define X v / #endif
      name = rest.split("//")[0].strip()
      if kw == "ifndef" and name in defs and i + 2 < len(lines) and re.match(rf"\s*#\s*define\s+{name}\b", lines[i + 1]) and re.match(r"\s*#\s*endif", lines[i + 2]):
        i += 3
        continue
      v = evaluate(cond, defs) if emitting() else False
      if not emitting():
        stack.append({"kind": "known", "taken": True, "emitting": False})
      elif v is None:
        out.append(line)
        stack.append({"kind": "unknown", "taken": False, "emitting": True})
      else:
        stack.append({"kind": "known", "taken": v, "emitting": v})
    elif kw == "elif":
      s = stack[-1]
      if s["kind"] == "unknown":
        out.append(line)
      else:
        outer = all(t["emitting"] for t in stack[:-1])
        if s["taken"] or not outer:
          s["emitting"] = False
        else:
          v = evaluate(rest, defs)
          if v is None:
            raise SystemExit(f"#elif with an unknown condition after a resolved #if is not supported: {line}")
          s["emitting"], s["taken"] = v, v
    elif kw == "else":
      s = stack[-1]
      if s["kind"] == "unknown":
        out.append(line)
      else:
        outer = all(t["emitting"] for t in stack[:-1])
        s["emitting"] = outer and not s["taken"]
        s["taken"] = True
    else:  # endif
      s = stack.pop()
      if s["kind"] == "unknown":
        out.append(line)
    i += 1
  assert not stack, "unbalanced conditionals"
  return "\n".join(out)


def main(argv):
  defs: dict[str, str | None] = {}
  files = []
  for a in argv:
    if a.startswith("-D"):
      n, _, v = a[2:].partition("=")
      defs[n] = v if v else "1"
    elif a.startswith("-U"):
      defs[a[2:]] = None
    else:
      files.append(a)
  for f in files:
    src = open(f).read()
    new = process(src, defs)
    if new != src:
      open(f, "w").write(new)
      print(f"{f}: {src.count(chr(10)) + 1} -> {new.count(chr(10)) + 1} lines")


if __name__ == "__main__":
  main(sys.argv[1:])
