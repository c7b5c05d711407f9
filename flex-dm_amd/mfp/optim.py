"""Keras-semantics Adam over the flat parameter buffer (one fused HIP pass).

Replaces ``tf.keras.optimizers.Adam(learning_rate, clipnorm=1.0)`` + the per-variable L2
regularisers as configured at reference train.py:71-77 / architecture/utils.py:8-22.
"""
import torch

from mfp.hip import ops


class AdamKeras:
    def __init__(self, store, learning_rate: float = 1e-4, beta_1: float = 0.9, beta_2: float = 0.999,
                 epsilon: float = 1e-7, clipnorm: float = 1.0):
        self.store = store
        self.lr, self.b1, self.b2, self.eps, self.clipnorm = learning_rate, beta_1, beta_2, epsilon, clipnorm
        dev = store.device
        self.m = torch.zeros_like(store.w)
        self.v = torch.zeros_like(store.w)
        self.chunks = ops.AdamChunks(store.layout.seg_offsets(), dev)
        self.stats = torch.zeros((self.chunks.nseg, 2), dtype=torch.float32, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)

    def step(self, grad_scale: float = 1.0):
        st = self.store
        ops.adam_keras(st.w, st.g, self.m, self.v, st.shadow, self.chunks, st.seg_l2, self.stats,
                       self.step_t, self.lr, self.b1, self.b2, self.eps, self.clipnorm, grad_scale)
        st.refresh_transposed()   # the Adam kernel refreshed the plain bf16 shadow itself

    def reg_loss(self) -> torch.Tensor:
        """sum_v l2_v * sum(w_v^2) with the weights as they were BEFORE the last step()."""
        return (self.stats[:, 1] * self.store.seg_l2).sum()

    def state_dict(self):
        return {"m": self.m.cpu(), "v": self.v.cpu(), "t": int(self.step_t.item())}

    def load_state_dict(self, state):
        self.m.copy_(state["m"])
        self.v.copy_(state["v"])
        self.step_t.fill_(int(state["t"]))
