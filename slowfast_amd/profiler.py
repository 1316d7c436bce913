"""Per-kernel timing of the C-ABI calls with HIP events on the launch stream (used by bench.py and tools/).

``with KernelProfiler() as prof: step()`` brackets every libsfamd call with torch.cuda events recorded on
the stream the kernel is launched on (torch's current stream); ``prof.summary()`` aggregates per entry
point: calls, total/avg ms and the algorithmic bytes/flops the wrappers in ops.py attach to each call."""
import collections

import torch

from . import lib as _lib

_active = None


class KernelProfiler:
    def __init__(self):
        self.records = []

    def __enter__(self):
        global _active
        _active = self
        _lib.set_call_observer(self._observe)
        return self

    def __exit__(self, *exc):
        global _active
        _lib.set_call_observer(None)
        _active = None
        torch.cuda.synchronize()

    def _observe(self, name, thunk, work):
        if not torch.cuda.is_available():
            return thunk()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        rc = thunk()
        en.record()
        self.records.append((name, st, en, work))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        agg = collections.OrderedDict()
        for name, st, en, work in self.records:
            a = agg.setdefault(name, dict(calls=0, ms=0.0, bytes=0.0, flops=0.0))
            a["calls"] += 1
            a["ms"] += st.elapsed_time(en)
            if work:
                a["bytes"] += work.get("bytes", 0.0)
                a["flops"] += work.get("flops", 0.0)
        for a in agg.values():
            a["avg_ms"] = a["ms"] / max(1, a["calls"])
            a["gbs"] = a["bytes"] / a["ms"] / 1e6 if a["ms"] > 0 else 0.0
            a["tflops"] = a["flops"] / a["ms"] / 1e9 if a["ms"] > 0 else 0.0
        return agg
