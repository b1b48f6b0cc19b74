"""tests/golden/mfma_probe_hw.npz: a slice of the raw MI355X matrix-core results (csrc/mfma_probe.hip via tools/gpu_mfma_probe.sh) with
their operand tiles, so that the CPU suite can hold tools/mfma_emul.c against HARDWARE data without a GPU.

    python tools/make_mfma_fixture.py gpurun_out/mfma_probe"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mfma_probe_cases as G  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma_probe"
TILES = 6
out = {}
for kind in ("bf16_32", "f16_32", "bf16_16"):
    sets = G.make_sets(kind, 100 + G.KINDS.index(kind))
    for name, (A, B, C0) in sets.items():
        T, steps, R, K = A.shape
        hw = np.fromfile(os.path.join(src, f"{name}.{kind}.out.bin"), np.float32).reshape(T, R, R)
        if kind.startswith("f16"):
            Av, Bv = A.astype(np.float16), B.astype(np.float16)
        else:
            Av, Bv = G.to_bf16_bits(A), G.to_bf16_bits(B)            # uint16 bit patterns
        sel = np.linspace(0, T - 1, TILES).astype(int)              # spread over the set's sub-cases (t % 3, t % 4 patterns)
        out[f"{kind}.{name}.A"], out[f"{kind}.{name}.B"] = Av[sel], Bv[sel]
        out[f"{kind}.{name}.C0"], out[f"{kind}.{name}.hw"] = C0[sel].astype(np.float32), hw[sel]
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mfma_probe_hw.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path))
