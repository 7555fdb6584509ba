"""Rewrite the generated parts of DESIGN.md / README.md from profiles/r05_* (the files scripts/summarise_profiles.py writes): the three
summary tables of DESIGN section 5, the round-5 row of its img/s table and README's headline numbers -- so the documents quote the
committed JSONs by construction.   python scripts/refresh_docs.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, *a)
b2, b4, b5 = (json.load(open(P("profiles", f"{t}_bench.json"))) for t in ("r05", "r05_hrnet_x4", "r05_blurskip_x8"))
pair = lambda b: f"{b['value']:.2f} / {b['other_precision']['value']:.2f}"
s = open(P("DESIGN.md")).read()
tables = ("**Config 2** (`profiles/r05_summary.md`):\n\n" + open(P("profiles", "r05_summary.md")).read() + "\n"
          "**Config 4** (`profiles/r05_hrnet_x4_summary.md`, first rows):\n\n" + "\n".join(open(P("profiles", "r05_hrnet_x4_summary.md")).read().splitlines()[:12]) + "\n\n"
          "**Config 5** (`profiles/r05_blurskip_x8_summary.md`, first rows):\n\n" + "\n".join(open(P("profiles", "r05_blurskip_x8_summary.md")).read().splitlines()[:12]) + "\n")
s = re.sub(r"(<!-- r05:tables[^\n]*-->\n).*?(<!-- /r05:tables -->)", lambda m: m.group(1) + tables + m.group(2), s, flags=re.S)
s = re.sub(r"\| round 5 \| \*\*[^|]*\*\* \| \*\*[^|]*\*\* \| \*\*[^|]*\*\* \|",
           f"| round 5 | **{pair(b2)} ({b2['ms_per_step']} ms)** | **{pair(b4)} ({b4['ms_per_step']} ms)** | **{pair(b5)} ({b5['ms_per_step']} ms)** |", s)
open(P("DESIGN.md"), "w").write(s)
r = open(P("README.md")).read()
r = re.sub(r"\*\*[0-9.]+ img/s in the detector precision mode that passes parity\*\*", f"**{b2['value']:.2f} img/s in the detector precision mode that passes parity**", r)
r = re.sub(r"and [0-9.]+ img/s in plain fp16; config 4 \(HRNet-OCR, B=4\) [0-9.]+ / [0-9.]+ img/s, config 5 \(×8, BlurSkip,\nB=4\) \*\*[0-9.]+\*\* / [0-9.]+ img/s",
           f"and {b2['other_precision']['value']:.2f} img/s in plain fp16; config 4 (HRNet-OCR, B=4) {pair(b4)} img/s, config 5 (×8, BlurSkip,\nB=4) **{b5['value']:.2f}** / {b5['other_precision']['value']:.2f} img/s", r)
open(P("README.md"), "w").write(r)
print("config 2", pair(b2), b2["ms_per_step"], "| config 4", pair(b4), "| config 5", pair(b5))
