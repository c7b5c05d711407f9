"""Phase timing of the d_model-512 Dense kernels from the D5_TRACE build (tools/abl/build_abl.sh block_d512 D5_TRACE 1):
MFP_HIP_LIB=tools/abl/libmfp_block_d512_1.so python tools/trace_d512.py   (shader clock = CLK_MHZ, default 100 MHz s_memtime?)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp import hip
from mfp.hip import ops
lib = hip.load()
T, D = int(os.environ.get("T", 16384)), 512
dev = "cuda"
bf = torch.bfloat16
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(bf)
trace = torch.zeros(1024 * 8 * 64, dtype=torch.int64, device=dev)
lib.mfp_debug_d512_trace.argtypes = [ctypes.c_void_p]
lib.mfp_debug_d512_trace(trace.data_ptr())
clk = float(os.environ.get("CLK_MHZ", 2080.0))      # s_memtime counts shader clocks


def report(name, fn, nwg, idxs):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    trace.zero_()
    fn()
    torch.cuda.synchronize()
    tr = trace[:nwg * 512].view(nwg, 8, 64).cpu().double()
    t0 = tr[:, :, 0].min()
    print("== %s: %d workgroups; workgroup start spread %.2f us, kernel span %.2f us" % (
        name, nwg, (tr[:, :, 0].max() - t0) / clk, (tr[:, :, max(idxs)].max() - t0) / clk))
    rel = tr - tr[:, :, :1]
    prev = 0.0
    for i in idxs:
        col = rel[:, :, i]
        v = col.mean().item() / clk
        print("  %2d  mean %7.2f us  +%6.2f   (min %.2f max %.2f over waves)" % (i, v, v - prev, col.min().item() / clk, col.max().item() / clk))
        prev = v


x = torch.randn(T, D, device=dev)
gam, bet = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
W, b = rnd(1536, D), torch.randn(1536, device=dev)
report("as512 LN1 + QKV (stamps: 0 start, 1 image read, 2+2cg kh0 done, 3+2cg kh1 done, 40 end)",
       lambda: ops.ln_dense_d512(x, gam, bet, W, b, 1536), 256, [42, 43, 44, 45, 1] + list(range(2, 26)) + [40])
A5, W2t, h = rnd(T, D), rnd(1024, D), torch.relu(torch.randn(T, 1024, device=dev)).to(bf)
report("as512 dh (mask)", lambda: ops.dense_relumask_d512(A5, W2t, h), 256, [1] + list(range(2, 18)) + [40])
res, bo = torch.randn(T, D, device=dev), torch.randn(D, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev)
A, Wf = rnd(T, 1024), rnd(D, 1024)
report("os512 FFN2 (stamps: 2+s stage s done, 1 loop done, 40 image written, 41 end)",
       lambda: ops.dense_n512_res(A, Wf, bo, res, (0.1, 5, 3), step), 256, list(range(2, 18)) + [1, 40, 41])
A, Wf = rnd(T, 1536), rnd(D, 1536)
report("os512 dy1 bf16", lambda: ops.dense_n512(A, Wf), 256, list(range(2, 26)) + [1, 40, 41])
