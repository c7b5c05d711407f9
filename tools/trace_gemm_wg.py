"""Phase timeline of gemm_wg_kernel (trace build, -DMFP_GEMM_TRACE): s_memrealtime stamps (10 ns).
Stamps of math wave 0: start, prologue barrier, every 4th k-tile, tile in LDS, barrier, stores retired."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch
from mfp import hip
from mfp.hip import ops

lib = ctypes.CDLL(os.path.join(ROOT, "tools", os.environ.get("TRACE_LIB", "libmfp_trace.so")))
lib.mfp_gemm.restype = ctypes.c_int32
lib.mfp_last_error.restype = ctypes.c_char_p
K = 32768
M, N = int(os.environ.get("M", 256)), int(os.environ.get("N", 512))
A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.float32)
sk = ops.wgrad_splitk(K, M, N)
ws = torch.empty(sk * M * N + sk * M, device="cuda", dtype=torch.float32)
nwg = ((M + 127) // 128) * ((N + 127) // 128) * sk
trace = torch.zeros(2 * nwg, 24, dtype=torch.int64, device="cuda")
a = hip.GemmArgs()
a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
a.M, a.N, a.K, a.lda, a.ldb, a.ldc = M, N, K, M, N, N
a.a_kmajor, a.b_kmajor, a.in_dtype, a.out_dtype, a.flags, a.splitk = 0, 0, 1, 0, 0, sk
a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
lib.mfp_trace_buffer(ctypes.c_void_p(trace.data_ptr()))
for _ in range(3):
    trace.zero_()
    rc = lib.mfp_gemm(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.mfp_last_error()
torch.cuda.synchronize()
full = trace.cpu().double()
mt = full[nwg:]
t = full[:nwg]
mt = mt[t[:, 0] > 0]
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
n = int((t[0] > 0).sum().item())
print("wgrad M=%d N=%d K=%d splitk=%d: %d workgroups, %d stamps" % (M, N, K, sk, t.shape[0], n))
for i in range(n):
    col = (t[:, i] - t0) / 100
    d = ((t[:, i] - t[:, i - 1]) / 100) if i else col
    print("stamp %2d  abs median %6.2f  p10 %6.2f p90 %6.2f | delta median %5.2f p90 %5.2f" % (
        i, col.median().item(), col.quantile(0.1).item(), col.quantile(0.9).item(), d.median().item(), d.quantile(0.9).item()))

names = ["lds write tile t+1 (waits its loads)", "issue loads tile t+5", "barrier", "loop overhead -> next step"]
d = mt[:, 1:16] - mt[:, 0:15]
print("memory wave 0, k-tiles 4..7, phases in core clocks (median / p90):")
for i in range(15):
    print("   %-40s %7.0f %7.0f" % (names[i % 4], d[:, i].median().item(), d[:, i].quantile(0.9).item()))
