"""Phase timeline of mlp_fused_kernel (trace build, -DMFP_GEMM_TRACE: tools/trace_gemm.sh with
TRACE_TOOL=trace_mlp.py): s_memrealtime stamps (10 ns) of thread 0 of every workgroup.  COLD=1 (default) gives
every launch its own input and output buffers, as the train step does."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
os.environ["MFP_HIP_LIB"] = os.path.join(ROOT, "tools", "libmfp_trace.so")
import torch
from mfp import hip
from mfp.hip import ops

T, D = int(os.environ.get("T", 32768)), 256
F = 2 * D
dev = "cuda"
cold = os.environ.get("COLD", "1") == "1"
X1 = [torch.randn(T, D, device=dev) for _ in range(12 if cold else 1)]
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
W1 = (torch.randn(F, D, device=dev) * 0.06).to(torch.bfloat16)
W2 = (torch.randn(D, F, device=dev) * 0.05).to(torch.bfloat16)
b1, b2 = torch.zeros(F, device=dev), torch.zeros(D, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev)
nb = (T + 127) // 128
trace = torch.zeros(nb, 24, dtype=torch.int64, device=dev)
lib = hip.load()
lib.mfp_mlp_trace_buffer.restype = None
keep = []
bwd = os.environ.get("BWD", "0") == "1"
W2t, W1t = W2.t().contiguous(), W1.t().contiguous()
DO = [(x * 0.3).to(torch.bfloat16) for x in X1]
HH = [torch.randn(T, F, device=dev).clamp(min=0).to(torch.bfloat16) for _ in X1]


def run(i):
    if bwd:
        return ops.mlp_fused_bwd(DO[i], HH[i], W2t, W1t)
    return ops.mlp_fused_fwd(X1[i], gamma, beta, W1, b1, W2, b2, (0.1, 7, 2), step)


for i in range(11):
    out = run(i % len(X1))
    if cold:
        keep.append(out)
torch.cuda.synchronize()
lib.mfp_mlp_trace_buffer(ctypes.c_void_p(trace.data_ptr()))
out = run(len(X1) - 1)
torch.cuda.synchronize()
lib.mfp_mlp_trace_buffer(None)
t = trace[:, :20].cpu().double()
rel = (t - t[:, 0].min()) / 100.0
names = ["start", "LN done", "barrier"] + ["chunk %d" % c for c in range(16)] + ["end"]
med, p90 = rel.median(0).values, rel.quantile(0.9, 0)
prev = 0.0
print("cold buffers: %s, backward: %s" % (cold, bwd))
for i, n in enumerate(names):
    print("%-9s median %6.2f us (+%5.2f)   p90 %6.2f   max %6.2f" % (n, med[i], med[i] - prev, p90[i], rel[:, i].max()))
    prev = med[i]
