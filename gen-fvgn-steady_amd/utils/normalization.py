"""Online feature normaliser with the reference's buffers and quirks (utils/normalization.py:4-85):
`acc_count` / `num_accumulations` start at 1.0, accumulation stops once num_accumulations reaches
max_accumulations, std < eps is replaced by 1.  The statistics are updated and applied by HIP kernels
(gfv_normalizer_update / gfv_node_prep); a host-side mirror of `num_accumulations` avoids the reference's
tensor->bool device synchronisation (normalization.py:39)."""
import torch
from torch import nn


class Normalizer(nn.Module):
    def __init__(self, size, max_accumulations=10 ** 7, epsilon=1e-8, device=None):
        super().__init__()
        self.max_accumulations = max_accumulations
        self.epsilon = epsilon
        self.register_buffer("acc_count", torch.tensor(1.0, dtype=torch.float32, device=device))
        self.register_buffer("num_accumulations", torch.tensor(1.0, dtype=torch.float32, device=device))
        self.register_buffer("acc_sum", torch.zeros(size, dtype=torch.float32, device=device))
        self.register_buffer("acc_sum_squared", torch.zeros(size, dtype=torch.float32, device=device))
        self._host_num_acc = None  # lazily synchronised mirror of num_accumulations

    def buffers_dict(self):
        return dict(acc_count=self.acc_count, num_accumulations=self.num_accumulations, acc_sum=self.acc_sum,
                    acc_sum_squared=self.acc_sum_squared)

    def should_accumulate(self):
        if self._host_num_acc is None:
            self._host_num_acc = float(self.num_accumulations)  # one sync, at first use / after load_state_dict
        return self._host_num_acc < self.max_accumulations

    def note_accumulated(self):
        self._host_num_acc += 1.0

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._host_num_acc = None

    def forward(self, batched_data, accumulate=True):
        """Standalone use (x [M, size] on the GPU): same result as the fused path inside NNmodel."""
        from gfv import lib as L
        lib = L.load()
        M, size = batched_data.shape
        assert size == 9, "the HIP normaliser is specialised for the 9 conditioning features (importer.py:32)"
        x = torch.zeros((M, 12), dtype=torch.float32, device=batched_data.device)
        x[:, 3:] = batched_data
        acc = accumulate and self.should_accumulate()
        nb = lib.gfv_normalizer_blocks(M)
        ws = torch.empty((nb, 18), dtype=torch.float32, device=x.device)
        mean_std = torch.empty(18, dtype=torch.float32, device=x.device)
        L.check(lib.gfv_normalizer_update(x.data_ptr(), 12, M, 1 if acc else 0, self.acc_count.data_ptr(),
                                          self.num_accumulations.data_ptr(), self.acc_sum.data_ptr(),
                                          self.acc_sum_squared.data_ptr(), ws.data_ptr(), mean_std.data_ptr(),
                                          L.stream_ptr()), "normalizer_update")
        if acc:
            self.note_accumulated()
        return (batched_data - mean_std[:9]) / mean_std[9:]
