"""Inference forward as a HIP graph (`sess.run([f_score, f_geometry], feed_dict=...)` per image:
test.py:131-145, test_pixellink_fast.py:88-92, test_pixellink.py:88-92).

One image through ResNet-v1-50 + heads is ~500 kernels of a few microseconds each: launched one by
one the forward is bound by launch latency, not by the GPU.  The launch sequence of an inference
forward is fixed for a given input shape (static buffers, no host decisions), so it is captured once
per shape on a capture stream and replayed as ONE graph launch; new inputs are copied into the
captured input buffers, outputs are the captured output buffers.

The weight packs (`Graph.packed`) are built before the capture and refreshed IN PLACE when the
variable store changes (checkpoint restore, optimiser step): the graph holds no pack kernels and
reads whatever the packs contain at replay time."""
import torch


class GraphedForward:
    def __init__(self, graph, fn, warmup=1, capture_after=1):
        """fn(graph, *device_tensors) -> outputs (any nesting of tensors / objects with `.data`).
        Must be a pure launch sequence: no `.item()`, no host<->device copies, no data-dependent
        Python control flow.  A shape is captured once it has been seen more than `capture_after`
        times (a capture costs about three forwards: a folder of differently-sized images should
        not pay it per image); until then the forward runs launch by launch."""
        self.g = graph
        self.fn = fn
        self.warmup = max(1, warmup)
        self.capture_after = capture_after
        self.seen = {}
        self.cache = {}

    def _eager(self, static):
        out = self.fn(self.g, *static)
        self.g.reset_tape()
        return out

    def _capture(self, inputs):
        static = [t.clone() for t in inputs]
        side = torch.cuda.Stream(device=self.g.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # variables, packs, workspaces, lazily-set kernel attributes
            for _ in range(self.warmup):
                self._eager(static)
        torch.cuda.current_stream().wait_stream(side)
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            out = self._eager(static)
        return [cg, static, out, self.g.store.version]

    def __call__(self, *inputs):
        key = tuple((tuple(t.shape), t.dtype) for t in inputs)
        ent = self.cache.get(key)
        if ent is None:
            self.seen[key] = self.seen.get(key, 0) + 1
            if self.seen[key] <= self.capture_after:
                return self._eager(inputs)
            ent = self.cache[key] = self._capture(inputs)
        cg, static, out, version = ent
        for dst, src in zip(static, inputs):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        if version != self.g.store.version:    # restored / updated weights: refresh the packs in place
            self._eager(static)
            ent[3] = self.g.store.version
        cg.replay()
        return out
