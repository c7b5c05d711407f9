"""Would two half-batch chains on two streams beat one full-batch chain?  A chain = N block-forward launches back to back.
(a) one stream, 256 documents per launch; (b) two streams, 128 documents per launch each, the second chain started half
a launch late.  Every launch is one workgroup per document; in (a) all 256 CUs run the same phase at the same time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.hip import ops
D, dev, N = 256, "cuda", 8
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
gam, bet = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
Wq, bq = rnd(768, 256), torch.randn(768, device=dev)
Wo, bo = rnd(256, 256), torch.randn(256, device=dev)
W1, b1, W2, b2 = rnd(512, 256), torch.randn(512, device=dev), rnd(256, 512), torch.randn(256, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev)


def chain(B, x):
    nvalid = torch.full((B,), 128, dtype=torch.int32, device=dev)
    for _ in range(N):
        x = ops.block_fwd(x, gam, bet, Wq, bq, Wo, bo, nvalid, gam, bet, W1, b1, W2, b2, B, 128, 8, 0.1, 5, 3, 4, step)[0]
    return x


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


x256 = torch.randn(256 * 128, D, device=dev)
xa, xb = x256[:128 * 128].clone(), x256[128 * 128:].clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print("one stream, 256 documents:  %.1f us per launch" % (timed(lambda: chain(256, x256)) / N))
print("one stream, 128 documents:  %.1f us per launch" % (timed(lambda: chain(128, xa)) / N))


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        chain(128, xa)
    with torch.cuda.stream(s2):
        torch.cuda._sleep(int(float(os.environ.get("LAG_US", 35)) * 2100))      # (cycles at ~2.1 GHz)
        chain(128, xb)
    cur.wait_stream(s1); cur.wait_stream(s2)


t = timed(two)
print("two streams, 128 + 128 documents, second chain %s us late: %.1f us per PAIR of launches (= per 256 documents)" % (os.environ.get("LAG_US", 35), t / N))
