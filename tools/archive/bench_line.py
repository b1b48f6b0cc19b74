"""Condense bench.py's JSON line (stdin or a file) to one human line: ms/step and the three buckets."""
import json, sys
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
d = json.loads([l for l in src if l.startswith("{")][-1])
r = d["roofline"]
print(f"{d['ms_per_step']:.1f} ms/step ({d['value']:.3f} it/s)  split-GEMM {r['ms_per_step']:.1f}  fp32-GEMM {r['other_gemm_family']['ms_per_step']:.1f}  rest {r['hbm_regime']['ms_per_step']:.1f}  "
      f"frac {r['frac']:.3f} pipe {r['mfma_pipe_util']:.3f}  traffic {r['traffic']}")
