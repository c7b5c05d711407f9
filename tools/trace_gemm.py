"""Phase timeline of gemm_kernel (trace build: hipcc -DMFP_GEMM_TRACE): s_memtime stamps of wave 0."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch
from mfp import hip

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libmfp_trace.so"))
lib.mfp_gemm.restype = ctypes.c_int32
lib.mfp_last_error.restype = ctypes.c_char_p
M, N, K = 32768, int(os.environ.get("N", 768)), int(os.environ.get("K", 256))
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
nwg = (M // 128) * ((N + 127) // 128)
trace = torch.zeros(nwg, 24, dtype=torch.int64, device="cuda")
a = hip.GemmArgs()
a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
a.M, a.N, a.K, a.lda, a.ldb, a.ldc = M, N, K, K, K, N
a.a_kmajor, a.b_kmajor, a.in_dtype, a.out_dtype, a.flags, a.splitk = 1, 1, 1, 1, 0, 1
lib.mfp_trace_buffer(ctypes.c_void_p(trace.data_ptr()))
os.environ["MFP_GEMM_TILE"] = "b2"
for _ in range(3):
    rc = lib.mfp_gemm(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.mfp_last_error()
torch.cuda.synchronize()
t = trace.cpu().double()
d = t[:, 1:] - t[:, :-1]
names = ["issue loads0", "land+lds0", "barrier0"] + sum([["mfma t%d" % i, "land+lds t%d" % (i + 1), "barrier t%d" % i] for i in range(4)], []) + ["(to epilogue)", "epilogue issue", "stores retire"]
print("s_memtime ticks (100 MHz constant clock => 10 ns/tick); median over %d workgroups" % nwg)
for i, n in enumerate(names):
    print("%-14s median %8.0f  p90 %8.0f" % (n, d[:, i].median().item(), d[:, i].quantile(0.9).item()))
print("total start->stores retired: median %.0f ticks" % (t[:, 18] - t[:, 0]).median().item())
span = (t[:, 18].max() - t[:, 0].min()).item()
q = torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64)
print("kernel span %.0f ticks; workgroup start offsets p10/p50/p90: %s" % (span, (t[:, 0] - t[:, 0].min()).quantile(q).tolist()))
print("workgroup end offsets p10/p50/p90: %s" % (t[:, 18] - t[:, 0].min()).quantile(q).tolist())

# global view on the 100 MHz s_memrealtime clock (10 ns ticks)
st, en = t[:, 20], t[:, 21]
span = (en.max() - st.min()).item()
print("realtime: kernel span %.2f us; mean workgroup life %.2f us; avg resident workgroups per CU %.2f" % (
    span / 100, (en - st).mean().item() / 100, (en - st).sum().item() / span / 256))
rel = torch.sort(st - st.min()).values / 100
print("workgroup start times (us) at ranks 0,255,256,511,512,767,1023,1535: %s" % [round(rel[i].item(), 2) for i in (0, 255, 256, 511, 512, 767, 1023, 1535) if i < nwg])
rele = torch.sort(en - st.min()).values / 100
print("workgroup end times (us) at ranks 0,255,511,767,1023,1279,1535: %s" % [round(rele[i].item(), 2) for i in (0, 255, 511, 767, 1023, 1279, 1535) if i < nwg])
