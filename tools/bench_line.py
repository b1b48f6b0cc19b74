import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: {d['value']:.3f} it/s  {d['ms_per_step']:.1f} ms/step  alg {d['algorithmic_tflops']:.1f} TF  gemm {r['achieved']:.0f} TF(mfma) / {r.get('achieved_algorithmic', r['achieved']):.0f} TF(alg) frac {r['frac']:.3f} share {r['share_of_step']:.2f}")
    elif line and "amdgpu.ids" not in line:
        print("   |", line[:200])
