"""Phase timeline of gemm_ws_kernel (trace build, -DMFP_GEMM_TRACE): s_memrealtime stamps (10 ns)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch
from mfp import hip

lib = ctypes.CDLL(os.path.join(ROOT, "tools", os.environ.get("TRACE_LIB", "libmfp_trace.so")))
lib.mfp_gemm.restype = ctypes.c_int32
lib.mfp_last_error.restype = ctypes.c_char_p
M, N, K = 32768, int(os.environ.get("N", 768)), int(os.environ.get("K", 256))
obf = int(os.environ.get("OUT_BF16", 1))
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
nwg = 256
trace = torch.zeros(2 * nwg, 24, dtype=torch.int64, device="cuda")
a = hip.GemmArgs()
a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
a.M, a.N, a.K, a.lda, a.ldb, a.ldc = M, N, K, K, K, N
a.a_kmajor, a.b_kmajor, a.in_dtype, a.out_dtype, a.flags, a.splitk = 1, 1, 1, obf, 0, 1
lib.mfp_trace_buffer(ctypes.c_void_p(trace.data_ptr()))
for _ in range(3):
    trace.zero_()
    rc = lib.mfp_gemm(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.mfp_last_error()
torch.cuda.synchronize()
tt = trace.cpu().double()[nwg:]
t = trace.cpu().double()[:nwg]
live = t[:, 0] > 0
t = t[live]
t0 = t[:, 0].min()
n = int((t[0] > 0).sum().item())
print("N=%d K=%d out_bf16=%d: %d workgroups, %d stamps each; times in us relative to the first workgroup start" % (N, K, obf, t.shape[0], n))
for i in range(n):
    col = (t[:, i] - t0) / 100
    d = ((t[:, i] - t[:, i - 1]) / 100) if i else col
    print("stamp %2d  abs median %6.2f  p10 %6.2f p90 %6.2f | delta median %5.2f p90 %5.2f" % (
        i, col.median().item(), col.quantile(0.1).item(), col.quantile(0.9).item(), d.median().item(), d.quantile(0.9).item()))

tt = tt[live]
names = ["top->mfma issued", "lds write X(t+1) (waits loads)", "issue loads X(t+3)", "epilogue+stores issued", "barrier"]
for k in range(2):
    d = tt[:, k * 8 + 1:k * 8 + 6] - tt[:, k * 8:k * 8 + 5]
    print("tile %d phases (s_memtime ticks), median / p90:" % (4 + k))
    for i, nm in enumerate(names):
        print("   %-32s %7.0f %7.0f" % (nm, d[:, i].median().item(), d[:, i].quantile(0.9).item()))
    print("   total %7.0f" % (tt[:, k * 8 + 5] - tt[:, k * 8]).median().item())
