"""Timeline of the LAST mlp_fused_kernel launch of a captured train step (trace build; c2 shape)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
os.environ["MFP_HIP_LIB"] = os.path.join(ROOT, "tools", "libmfp_trace.so")
import torch
from mfp import hip
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.models.mfp import MFP

dev = "cuda:0"
ic = make_input_columns("crello")
batch = synthetic_batch(ic, 256, 128, seed=0, ragged=False, device=dev)
model = MFP(ic, num_blocks=4, latent_dim=256, dropout=0.1, l2=1e-2, masking_method="random", dtype="bf16", device=dev, seed=0)
model.compile(learning_rate=1e-4, clipnorm=1.0)
model.model.store.refresh_shadow()
lib = hip.load()
lib.mfp_mlp_trace_buffer.restype = None
trace = torch.zeros(256, 24, dtype=torch.int64, device=dev)
dummy = torch.zeros(1 << 23, dtype=torch.int64, device=dev)       # the GEMM kernels of the trace build stamp unconditionally
lib.mfp_trace_buffer.restype = None
lib.mfp_trace_buffer(ctypes.c_void_p(dummy.data_ptr()))
lib.mfp_mlp_trace_buffer(ctypes.c_void_p(trace.data_ptr()))      # before capture: the pointer is baked into the graph
model.capture_train_step(batch, warmup=2)
batch = model.static_batch
for _ in range(5):
    model.train_step(batch)
torch.cuda.synchronize()
t = trace[:, :20].cpu().double()
rel = (t - t[:, 0].min()) / 100.0
names = ["start", "LN done", "barrier"] + ["chunk %d" % c for c in range(16)] + ["end"]
med, p90 = rel.median(0).values, rel.quantile(0.9, 0)
prev = 0.0
for i, n in enumerate(names):
    print("%-9s median %6.2f us (+%5.2f)   p90 %6.2f   max %6.2f" % (n, med[i], med[i] - prev, p90[i], rel[:, i].max()))
    prev = med[i]
