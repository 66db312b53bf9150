"""A launch set resident in HBM, traced through srt_trace_batch_device (the entry point bench.py times).

torch is used for device memory and streams only.  One DeviceBatch = the rays of one shard on one GPU:
inputs as SoA [3][n] (coalesced refills), `nbuf` sets of output buffers (rows, nrows, stop, counters) so that
launches on different streams may be in flight together.
"""
import ctypes as C

import numpy as np

from . import api


class DeviceBatch:
    def __init__(self, model, params, pos0, dir0, w0, device, nbuf=1):
        import torch

        self.torch = torch
        self.model, self.p, self.dev = model, params, device
        self.n = int(len(w0))
        self.slots = api.lib().srt_rows_per_ray(C.byref(params))
        pos0 = np.asarray(pos0, dtype=np.float64).reshape(-1, 3)
        dir0 = np.asarray(dir0, dtype=np.float64).reshape(-1, 3)
        self.d_pos = torch.from_numpy(np.ascontiguousarray(pos0.T)).to(device)
        self.d_dir = torch.from_numpy(np.ascontiguousarray(dir0.T)).to(device)
        self.d_w = torch.from_numpy(np.ascontiguousarray(w0, dtype=np.float64)).to(device)
        self.out = [{"rows": torch.zeros((self.n, self.slots, api.ROW), dtype=torch.float64, device=device),
                     "nrows": torch.zeros(self.n, dtype=torch.int32, device=device),
                     "stop": torch.zeros(self.n, dtype=torch.int32, device=device),
                     "cnt": torch.zeros(4, dtype=torch.int64, device=device)} for _ in range(max(1, nbuf))]

    def launch(self, buf=0, stream=None, counters=None):
        """Enqueue one trace of the whole batch; returns the output dict (asynchronous)."""
        torch = self.torch
        o = self.out[buf]
        st = stream if stream is not None else torch.cuda.current_stream(self.dev)
        cnt = counters if counters is not None else o["cnt"]
        if self.n == 0:
            cnt.zero_()
            return o
        rc = api.lib().srt_trace_batch_device(self.model.h, C.byref(self.p), self.n, self.d_pos.data_ptr(),
                                              self.d_dir.data_ptr(), self.d_w.data_ptr(), o["rows"].data_ptr(),
                                              o["nrows"].data_ptr(), o["stop"].data_ptr(), cnt.data_ptr(), st.cuda_stream)
        api._check(rc)
        return o

    def trace(self, buf=0):
        """launch + the (rows, nrows, stop) triple parallel.trace_sharded expects."""
        o = self.launch(buf)
        return o["rows"], o["nrows"], o["stop"]
