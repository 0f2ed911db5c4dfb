"""SumTree — Python face of the GPU-resident sum tree.

Mirrors the pybind class the reference binds as `sum_tree.SumTreef`
(sum_tree/sum_tree/src/sum_tree_py.cc:9-22): same method names and argument meaning
(`update_value(s)`, `get_index/indices`, `get_value(s)`, `get_capacity`, `get_total_val`), accepting
Python lists / numpy arrays like the original, plus device-tensor methods (`*_dev`) that the
GPU replay uses so nothing crosses PCIe.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi as K


class SumTree:
    def __init__(self, capacity, device=None):
        if not torch.cuda.is_available():
            raise K.HbError("SumTree needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
        self.L = K.lib()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            K.check(self.L.hb_tree_create(int(capacity), C.byref(h)))
        self.h = h
        self.capacity = int(self.L.hb_tree_capacity(h))
        self._total = torch.zeros(1, dtype=torch.float32, device=self.device)

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            self.L.hb_tree_destroy(h)
            self.h = None

    # ---- device-tensor API (hot path) -------------------------------------------------------
    def update_dev(self, idx, val):
        assert idx.dtype == torch.int64 and val.dtype == torch.float32 and idx.is_cuda and val.is_cuda
        K.check(self.L.hb_tree_update(self.h, K.dptr(idx), K.dptr(val), idx.numel(), K.current_stream()))

    def fill_range_dev(self, start, n, value_dev):
        K.check(self.L.hb_tree_fill_range(self.h, int(start), int(n), K.dptr(value_dev), K.current_stream()))

    def sample_dev(self, quantiles, idx_out=None, val_out=None):
        q = quantiles
        assert q.dtype == torch.float32 and q.is_cuda
        n = q.numel()
        idx = idx_out if idx_out is not None else torch.empty(n, dtype=torch.int64, device=self.device)
        val = val_out if val_out is not None else torch.empty(n, dtype=torch.float32, device=self.device)
        K.check(self.L.hb_tree_sample(self.h, K.dptr(q), K.dptr(idx), K.dptr(val), n, K.current_stream()))
        return idx, val

    def get_dev(self, idx):
        val = torch.empty(idx.numel(), dtype=torch.float32, device=self.device)
        K.check(self.L.hb_tree_get(self.h, K.dptr(idx), K.dptr(val), idx.numel(), K.current_stream()))
        return val

    def total_dev(self):
        K.check(self.L.hb_tree_total(self.h, K.dptr(self._total), K.current_stream()))
        return self._total

    def per_sample_dev(self, u, idx_out=None, prob_out=None, unit=False):
        """Stratified PER sampling on given uniforms u[i] in [0, 1/B) — or in [0, 1) with unit=True — (float64)
        -> (indices, probabilities)."""
        assert u.dtype == torch.float64 and u.is_cuda
        b = u.numel()
        idx = idx_out if idx_out is not None else torch.empty(b, dtype=torch.int64, device=self.device)
        prob = prob_out if prob_out is not None else torch.empty(b, dtype=torch.float64, device=self.device)
        K.check(self.L.hb_per_sample(self.h, K.dptr(u), b, 1 if unit else 0, K.dptr(idx), K.dptr(prob), K.current_stream()))
        return idx, prob

    def per_sample_philox_dev(self, seed, counter_dev, batch, idx_out=None, prob_out=None):
        """per_sample_dev with the stratified uniforms drawn inside the kernel from Philox(seed; i, counter)."""
        assert counter_dev.dtype == torch.float32 and counter_dev.is_cuda
        idx = idx_out if idx_out is not None else torch.empty(batch, dtype=torch.int64, device=self.device)
        prob = prob_out if prob_out is not None else torch.empty(batch, dtype=torch.float64, device=self.device)
        K.check(self.L.hb_per_sample_philox(self.h, int(seed), K.dptr(counter_dev), int(batch), K.dptr(idx), K.dptr(prob),
                                            K.current_stream()))
        return idx, prob

    def per_update_dev(self, idx, td, alpha, max_prio_dev, min_prio_dev):
        assert idx.dtype == torch.int64 and td.dtype == torch.float32
        K.check(self.L.hb_per_update(self.h, K.dptr(idx), K.dptr(td), idx.numel(), float(alpha), K.dptr(max_prio_dev),
                                     K.dptr(min_prio_dev), K.current_stream()))

    def set_lazy_top(self, on=True):
        """Writers stop re-summing the levels above the 1024-leaf subtrees; readers do it (hb_tree_set_lazy_top)."""
        K.check(self.L.hb_tree_set_lazy_top(self.h, 1 if on else 0))

    def nodes(self):
        """Copy of the 2*capacity heap (root at index 1) as a CUDA tensor (tests)."""
        out = torch.empty(2 * self.capacity, dtype=torch.float32, device=self.device)
        K.check(self.L.hb_tree_export_nodes(self.h, K.dptr(out), K.current_stream()))
        return out

    def import_nodes(self, nodes):
        """Overwrite the heap with a tensor produced by nodes() (checkpoint resume)."""
        nodes = nodes.to(device=self.device, dtype=torch.float32).contiguous()
        assert nodes.numel() == 2 * self.capacity
        K.check(self.L.hb_tree_import_nodes(self.h, K.dptr(nodes), K.current_stream()))
        torch.cuda.current_stream().synchronize()  # `nodes` may be a temporary

    def error_count(self):
        v = C.c_int64()
        K.check(self.L.hb_tree_error_count(self.h, C.byref(v)))
        return v.value

    # ---- reference-shaped API (sum_tree_py.cc:11-22) -------------------------------------------
    def _i64(self, a):
        return torch.as_tensor(np.asarray(a, dtype=np.int64)).to(self.device)

    def _f32(self, a):
        return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device)

    def update_value(self, index, value):
        self.update_values([index], [value])

    def update_values(self, indices, values):
        self.update_dev(self._i64(indices), self._f32(values))

    def get_index(self, quantile):
        return self.get_indices([quantile])[0]

    def get_indices(self, quantiles):
        idx, _ = self.sample_dev(self._f32(quantiles))
        return [int(i) for i in idx.cpu().numpy()]

    def get_value(self, index):
        if not 0 <= int(index) < self.capacity:
            raise IndexError(index)  # the reference throws from unordered_map::at (sum_tree.h:62)
        return self.get_values([index])[0]

    def get_values(self, indices):
        return [float(v) for v in self.get_dev(self._i64(indices)).cpu().numpy()]

    def get_capacity(self):
        return self.capacity

    def get_total_val(self):
        return float(self.total_dev().cpu()[0])

    def __repr__(self):
        return f"<SumTree(capacity={self.capacity}, maxval={self.get_total_val():f})>"
