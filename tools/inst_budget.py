"""Count instructions per class in the kernels of tools/inst_budget.hip (static, per trip); see that file."""
import collections
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
nvp = sys.argv[1] if len(sys.argv) > 1 else "36"
out = Path("/tmp/inst_budget.s")
subprocess.run(["hipcc", "-O3", "-std=c++17", "-ffp-contract=on", "--offload-arch=gfx950", "-S", "--cuda-device-only", f"-DMJLAB_NVP={nvp}", *sys.argv[2:],
                str(ROOT / "tools" / "inst_budget.hip"), "-o", str(out)], check=True)
cur, cnt = None, collections.defaultdict(collections.Counter)
for line in out.read_text().splitlines():
  m = re.match(r"^(kb_\w+):", line)
  if m:
    cur = m.group(1)
    continue
  if cur is None:
    continue
  t = line.strip()
  if t.startswith("s_endpgm"):
    cur = None
    continue
  if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
    continue
  op = t.split()[0]
  cls = ("mfma" if "mfma" in op else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_barrier")) else
         "wait" if op.startswith(("s_waitcnt", "s_nop")) else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
  cnt[cur][cls] += 1
  if op.startswith(("v_readlane", "v_readfirstlane")):
    cnt[cur]["readlane"] += 1
  if "dpp" in t:
    cnt[cur]["dpp"] += 1
  if op.startswith("v_pk_"):
    cnt[cur]["pk"] += 1
base = cnt["kb_empty"]
print(f"{'block':14s} {'valu':>6s} {'salu':>6s} {'lds':>5s} {'vmem':>5s} {'mfma':>5s} {'wait':>5s} | readlane dpp pk   (minus kb_empty: {dict(base)})")
for k, c in cnt.items():
  d = {x: c[x] - base.get(x, 0) for x in ("valu", "salu", "lds", "vmem", "mfma", "wait")}
  print(f"{k:14s} {d['valu']:6d} {d['salu']:6d} {d['lds']:5d} {d['vmem']:5d} {d['mfma']:5d} {d['wait']:5d} | {c['readlane']:6d} {c['dpp']:4d} {c['pk']:4d}")
