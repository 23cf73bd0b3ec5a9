"""Dev tool: time the bare split-precision weight-gradient GEMM (es_gemm_atb) on a problem of the training step's size: 256 tasks of
~6 400 rows.  usage: python tools/dev/wgrad_x3_probe.py [rows]   (prints ms, TB/s of operand bytes, bf16 TFLOP/s incl. the six products)"""
import sys
import torch
from endosurf_amd import _lib
from endosurf_amd.engine import Engine

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 6464
FILL = sys.argv[2] if len(sys.argv) > 2 else "randn"      # randn | zeros | relu (half of the entries exactly zero)
eng = Engine("cuda")
X = torch.randn(M, 256, device="cuda")
dA = torch.randn(M, 256, device="cuda")
if FILL == "zeros":
    X.zero_(); dA.zero_()
elif FILL == "relu":
    X.relu_(); dA.mul_((torch.rand_like(dA) < 0.5).float())
out = torch.zeros(256, 256, device="cuda")
def run(x3):
    _lib.check(eng.lib.es_gemm_atb(_lib.ptr(X), _lib.ptr(dA), M, _lib.ptr(out), x3, 0, eng.st()), "es_gemm_atb")
for x3 in (1, 0):
    for _ in range(3): run(x3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run(x3)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{FILL} x3={x3} rows={M} {ms:.4f} ms  {M * 2048 / ms / 1e9:.2f} TB/s  {M * 65536 * 2 * (6 if x3 else 1) / ms / 1e9:.0f} TFLOP/s")
