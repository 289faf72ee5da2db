"""Double-buffered device feeder (the reference's GeneratorEnqueuer + queue,
tool/data_util.py:15-128, datasets/icdar.py:652-668).

A host thread runs the batch generator with ITS OWN HIP stream as torch's current stream, so the
pinned-slab upload and the resize / label kernels of batch k+1 overlap the training step of batch k
on the compute stream.  Each finished batch is handed over with an event; the consumer's stream
waits on it (no host synchronisation) and the tensors are marked as used by the consumer's stream so
the caching allocator does not recycle them early."""
import queue
import threading

import torch


class DeviceFeeder:
    def __init__(self, make_iterator, device, depth=2):
        self.device = torch.device(device)
        self.q = queue.Queue(maxsize=max(1, depth))
        self.stop = threading.Event()
        self.err = None
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.thread = threading.Thread(target=self._run, args=(make_iterator,), daemon=True)
        self.thread.start()

    def _put(self, item):
        while not self.stop.is_set():
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self, make_iterator):
        try:
            if self.stream is not None:
                torch.cuda.set_device(self.device)
                ctx = torch.cuda.stream(self.stream)
            else:
                import contextlib
                ctx = contextlib.nullcontext()
            with ctx:
                it = make_iterator()
                try:
                    for batch in it:
                        ev = None
                        if self.stream is not None:
                            ev = torch.cuda.Event()
                            ev.record(self.stream)
                        if not self._put((batch, ev)):
                            return
                finally:
                    if hasattr(it, "close"):
                        it.close()           # runs the generator's own clean-up (worker pool) in THIS thread, now
        except BaseException as e:          # surfaced to the consumer
            self.err = e
        finally:
            self._put(None)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is None:
            self.q.put(None)
            if self.err is not None:
                raise self.err
            raise StopIteration
        batch, ev = item
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in batch:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)
        return batch

    def close(self):
        self.stop.set()
        while True:                         # unblock the producer
            try:
                self.q.get_nowait()
            except queue.Empty:
                break
        self.thread.join(timeout=10)

    def __del__(self):
        try:
            self.stop.set()
        except Exception:
            pass
