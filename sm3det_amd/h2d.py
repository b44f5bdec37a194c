"""Host-to-device hand-over of a step's per-source batch (second half).

Mirror of ``TriSourceDetector.gather_dict_values`` (``mmrotate/models/detectors/
trisource_H1stage_R2stage_detector.py:190-206``): collect, per training source (``'sar'``, ``'rgb'``, ``'ifr'`` ...), the
entries of a list of per-sample dicts; tensors are stacked (or kept as a list with ``ignore_tensor=True``) and moved to
the GPU.  The reference does ``torch.stack(...).cuda()`` -- a pageable, synchronous copy per source per step.  Here
the stack is written straight into a cached **pinned** staging buffer and uploaded with a non-blocking copy on a
dedicated copy stream; the consumer stream waits on an event, so the 25 MB image batch of a 2x1024^2 step overlaps
with whatever the compute stream is still doing (e.g. the previous step's optimizer graph)."""
import torch


class PinnedUploader:
    def __init__(self, device='cuda'):
        if not torch.cuda.is_available():
            raise RuntimeError('PinnedUploader needs the GPU runtime (pinned host memory); there is no CPU fallback')
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._staging = {}   # (tag, dtype) -> [flat pinned buffer (grow-only), event of its last upload]

    def _stage(self, tag, shape, dtype):
        """A pinned view of `shape` out of ONE grow-only flat buffer per (tag, dtype): variable-length entries
        (``ignore_tensor=True``: per-sample gt boxes / labels of any count) reuse it instead of page-locking a fresh
        buffer for every distinct shape."""
        key = (tag, dtype)
        numel = 1
        for d in shape:
            numel *= int(d)
        slot = self._staging.get(key)
        if slot is not None and slot[1] is not None:
            slot[1].synchronize()  # the previous upload out of this buffer must have left the host memory
        if slot is None or slot[0].numel() < numel:
            slot = [torch.empty(max(numel, 1), dtype=dtype, pin_memory=True), None]
            self._staging[key] = slot
        return [slot[0][:numel].view(tuple(shape)), slot]

    def upload(self, tensors, tag='', stack=True):
        """list of equally-shaped CPU tensors -> one stacked GPU tensor (stack=True) or a list of GPU tensors."""
        if any(t.is_cuda for t in tensors):  # already on the device (the reference's .cuda() accepts both): pass through
            ts = [t.to(self.device) for t in tensors]
            return torch.stack(ts) if stack else ts
        if stack:
            first = tensors[0]
            view, slot = self._stage(tag, (len(tensors),) + tuple(first.shape), first.dtype)
            torch.stack(list(tensors), out=view)
            return self._send(view, slot)
        out = []
        for i, t in enumerate(tensors):
            view, slot = self._stage(f'{tag}/{i}', tuple(t.shape), t.dtype)
            view.copy_(t)
            out.append(self._send(view, slot))
        return out

    def _send(self, view, slot):
        with torch.cuda.stream(self.stream):
            dev = view.to(self.device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(self.stream)
        torch.cuda.current_stream(self.device).wait_event(slot[1])
        dev.record_stream(torch.cuda.current_stream(self.device))
        return dev


def collect_by_source(data, train_datasets):
    """the host half of gather_dict_values (:191-196): {source: [item[source] for item in data if present]}"""
    gathered = {ns: [] for ns in train_datasets}
    for item in data:
        for ns in train_datasets:
            if item.get(ns) is not None:
                gathered[ns].append(item[ns])
    return gathered


def gather_dict_values(data, train_datasets, ignore_tensor=False, uploader=None):
    """Same result as the reference method (:190-206); non-tensor entries (img_metas ...) are passed through."""
    gathered = collect_by_source(data, train_datasets)
    up = uploader
    for ns in train_datasets:
        vals = gathered[ns]
        if vals and isinstance(vals[0], torch.Tensor):
            if up is None:
                up = PinnedUploader()
            gathered[ns] = up.upload(vals, tag=ns, stack=not ignore_tensor)
    return gathered
