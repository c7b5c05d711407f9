"""Phase timeline of gemm_wgg_kernel (trace build, -DMFP_GEMM_TRACE): s_memrealtime stamps (10 ns) of
thread 0 of every workgroup: start, first tile staged, every 16th k-tile, tile in LDS, slab drained,
ticket drawn, (last arrivers) reduced."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
os.environ["MFP_HIP_LIB"] = os.path.join(ROOT, "tools", "libmfp_trace.so")
import torch
from mfp import hip
from mfp.hip import ops
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_wgrad

lib = hip.load()
lib.mfp_trace_buffer.restype = None
jobs = bench_wgrad.block_jobs()
T = bench_wgrad.T
sk = int(os.environ.get("SK", 8))
nwg = 32 * sk
trace = torch.zeros(nwg, 24, dtype=torch.int64, device="cuda")
lib.mfp_trace_buffer(ctypes.c_void_p(trace.data_ptr()))
for _ in range(3):
    trace.zero_()
    ops.wgrad_group(jobs, T, sk)
torch.cuda.synchronize()
t = trace.cpu().double()
t0 = t[:, 0].min()
nst = (t > 0).sum(1)
print("block group, splitk=%d: %d workgroups; stamps per workgroup: %s" % (sk, nwg, sorted(set(nst.tolist()))))
names = ["start", "first tile staged"] + ["k-tile %d" % (16 * (i + 1)) for i in range(64 // 16)] + ["tile in LDS", "slab drained", "ticket drawn", "reduced"]
for i in range(int(nst.max())):
    sel = nst > i
    col = (t[sel, i] - t0) / 100
    print("%-18s n=%4d  abs median %6.2f us  p10 %6.2f  p90 %6.2f  max %6.2f" % (names[i] if i < len(names) else i, int(sel.sum()), col.median(), col.quantile(0.1), col.quantile(0.9), col.max()))
