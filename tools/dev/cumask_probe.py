"""Dev probe: does hipExtStreamCreateWithCUMask partition the CUs?  Times the 32768-point SDF query on streams with different masks."""
import ctypes as C
import sys

import torch

sys.path[:0] = [".", "tests"]
import weightgen  # noqa: E402
from endosurf_amd import _lib, params  # noqa: E402
from endosurf_amd._lib import es_points  # noqa: E402

hip = C.CDLL("libamdhip64.so")
lib = _lib.load()
_lib.check(lib.es_init(), "es_init")
state = weightgen.make_state(3, "trained", True)
flat = torch.from_numpy(params.flatten_state(state)).cuda()
weff = torch.zeros(lib.es_weff_floats(), device="cuda")
packed = torch.zeros(lib.es_packed_floats(), device="cuda")
_lib.check(lib.es_weightnorm_pack(_lib.ptr(flat), _lib.ptr(weff), _lib.ptr(packed), 1, _lib.stream_ptr()))
torch.cuda.synchronize()


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def query(M, stream, n=20):
    x = (torch.rand(M, 3, device="cuda") * 1.6 - 0.8).contiguous(); t = torch.rand(M, device="cuda"); out = torch.empty(M, device="cuda")
    pts = es_points(); pts.x, pts.t = _lib.ptr(x), _lib.ptr(t); pts.mode, pts.t_scalar, pts.n_per_ray, pts.ldz, pts.M = 0, 0, 1, 1, M
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(n + 2):
            if i == 2:
                e0.record()
            lib.es_query_sdf(C.byref(pts), _lib.ptr(packed), _lib.ptr(weff), _lib.ptr(out), 1, C.c_void_p(stream.cuda_stream))
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


full = (1 << 256) - 1
for name, bits in [("all 256 bits", full), ("bits 0..63", (1 << 64) - 1), ("bits 64..255", full ^ ((1 << 64) - 1)), ("bits 0..127", (1 << 128) - 1),
                   ("every 4th bit", sum(1 << i for i in range(0, 256, 4))), ("bits with (i%32)<8", sum(1 << i for i in range(256) if i % 32 < 8))]:
    s = masked_stream(bits)
    print(f"{name:22s}  32768 pts: {query(32768, s):7.1f} us   1024 pts: {query(1024, s):6.1f} us")
# concurrency: big query on bits 64..255 while small queries run on bits 0..63
sa, sb = masked_stream(full ^ ((1 << 64) - 1)), masked_stream((1 << 64) - 1)
import threading
res = {}
def run(k, M, s, n): res[k] = query(M, s, n)
ta = threading.Thread(target=run, args=("big", 32768, sa, 40)); tb = threading.Thread(target=run, args=("small", 1024, sb, 160))
ta.start(); tb.start(); ta.join(); tb.join()
print("concurrent on disjoint masks: big", round(res["big"], 1), "us, small", round(res["small"], 1), "us")
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
ta = threading.Thread(target=run, args=("big", 32768, s0, 40)); tb = threading.Thread(target=run, args=("small", 1024, s1, 160))
ta.start(); tb.start(); ta.join(); tb.join()
print("concurrent on plain streams:  big", round(res["big"], 1), "us, small", round(res["small"], 1), "us")
