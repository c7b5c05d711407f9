"""Layout probe of v_mfma_scale_f32_16x16x128_f8f6f4 (through mfp_debug_mx_probe): which scale lane applies to which
(lane group, byte) of the A and B operands."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp import hip
lib = hip.load()
dev = "cuda"
ONE = 0x38  # e4m3 1.0
def run(a, b, sa, sb):
    out = torch.zeros(64, 4, device=dev)
    rc = lib.mfp_debug_mx_probe(a.data_ptr(), b.data_ptr(), sa.data_ptr(), sb.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    return out.cpu()
s127 = torch.full((64,), 127, dtype=torch.int32, device=dev)
allone = torch.full((64, 32), ONE, dtype=torch.uint8, device=dev)
for side in "AB":
    print("operand", side, ": scale lane group that applies to (data lane group g, byte t), row/col 5")
    for g in range(4):
        row = []
        for t in range(32):
            x = torch.zeros(64, 32, dtype=torch.uint8, device=dev); x[16 * g + 5, t] = ONE
            base = run(x, allone, s127, s127) if side == "A" else run(allone, x, s127, s127)
            found = []
            for s in range(4):
                sc = s127.clone(); sc[16 * s + 5] = 130
                o = run(x, allone, sc, s127) if side == "A" else run(allone, x, s127, sc)
                if not torch.equal(o, base): found.append(s)
            row.append("".join(map(str, found)) or "-")
        print(" g=%d:" % g, " ".join(row))
