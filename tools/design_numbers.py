#!/usr/bin/env python3
"""The numbers of DESIGN.md section 3 / README from a committed profile set: python tools/design_numbers.py [tag = r06zz]
(reads profiles/<tag>_{bench,sustained}_<config>.json, <tag>_<config>_{kernel_stats,pmc}.txt, <tag>_shapes.txt)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06zz"
P = lambda name: os.path.join(ROOT, "profiles", f"{TAG}_{name}")
PEAK = 157.3e12


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def pmc(path):
    d = {}
    for line in open(path):
        m = re.match(r"\s+(\w+)\s+([\d.]+)\s+\(", line)
        if m:
            d[m.group(1)] = float(m.group(2))
    return d


def kernel_avg_us(path):
    m = re.search(r"\n\s+\d+\s+[\d.]+\s+([\d.]+)\s+[\d.]+\s+void mcd::score", open(path).read())
    return float(m.group(1))


print(open(P("MANIFEST.txt")).read().strip())
for cfg in ("avenue", "stc", "ubnormal_concat", "seq24", "concat24", "concat32"):
    b = last_json(P(f"bench_{cfg}.json")) if os.path.exists(P(f"bench_{cfg}.json")) else None
    avg = kernel_avg_us(P(f"{cfg}_kernel_stats.txt"))
    c = pmc(P(f"{cfg}_pmc.txt"))
    line = f"{cfg:16s} rocprofv3 avg {avg:10.1f} us"
    if b:
        r = b["roofline"]
        # the profiled launch may hold fewer windows than the benched step (seq24: 1024 of 4096)
        win_prof = 1024 if cfg == "seq24" else b["config"]["windows_per_step_per_gpu"]
        frac_prof = win_prof * r["flop_per_window"] / (avg * 1e-6) / PEAK
        line += f"  frac profiler {frac_prof:.4f} / bench {r['frac']:.4f}  value {b['value']:.1f}"
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * avg * 1e-6 * 2.4e9)
    other = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
    traffic = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    line += f"  pipe {busy:.3f}  valu/mfma {other:.2f}  traffic {traffic / 1e6:.1f} MB  L2 hit {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.4f}"
    line += f"  wait_inst {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}  bank {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}"
    print(line)
    if cfg == "avenue":
        print("   per launch: MFMA %.1f M, other VALU %.1f M (%.0f per wave-pass), SALU %.1f M, LDS %.1f M, VMEM_RD %.1f M" % (
            c["SQ_INSTS_MFMA"] / 1e6, (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / 1e6, (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / 184320,
            c["SQ_INSTS_SALU"] / 1e6, c["SQ_INSTS_LDS"] / 1e6, c["SQ_INSTS_VMEM_RD"] / 1e6))
for cfg in ("avenue", "stc", "ubnormal_concat", "seq24"):
    if os.path.exists(P(f"sustained_{cfg}.json")):
        s = last_json(P(f"sustained_{cfg}.json"))
        print(f"sustained {cfg:16s} value {s['value']:.1f}  frac {s['roofline']['frac']:.4f}  steps {s['steps']}")
b = last_json(P("bench_avenue.json"))
for k in ("sustained", "e2e", "cpu_baseline", "auc"):
    print(k, json.dumps(b.get(k))[:400])
print(open(P("shapes.txt")).read())
