"""Pyramid levels on parallel HIP streams.

The reference applies its shared-weight heads level by level (``multi_apply(self.forward_single, feats, ...)``: mmdet
``anchor_head.py`` / ``gfl_head.py:forward``, mmrotate ``rotated_rpn_head.py:forward``).  The five levels of a 1024^2 image
are 128^2 ... 8^2 positions: the launches of the three coarse levels cannot fill 256 CUs (64 ... 1024 rows of a GEMM) and
run at the launch floor one after the other.  The levels are independent until the loss, so each gets its own stream:
the forward of level l is enqueued on stream l, autograd runs its backward on the same stream (and orders the streams
around the shared-weight gradient accumulation itself), and under hipGraph capture the fork / join become graph edges --
the small launches run underneath the 128^2 level's.  Nothing about the arithmetic changes.

MEASURED AND OFF BY DEFAULT (round 6, full detector step, same box, eager / hipGraph replay): levels one after the other
58.0 / 57.3 ms; one stream per level (``SM3_LEVEL_STREAMS=1``) 60.3 / 60.7 ms; the three coarse levels on one side stream
(``=2``) 60.9 / 59.7 ms.  Cross-stream edges cost more on this runtime than the idle CUs under the small launches are
worth; what remains is the grouped form (one launch over the concatenated rows of all levels).
"""
import os

import torch

MODE = int(os.environ.get('SM3_LEVEL_STREAMS', '0'))  # 0 off, 1 one stream per level, 2 the coarse levels (>= 2) on ONE side stream
ENABLED = MODE != 0
_STREAMS = {}  # device index -> [side streams]


def _streams(device, n):
    pool = _STREAMS.setdefault(device.index if device.index is not None else torch.cuda.current_device(), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _record(obj, stream):
    if torch.is_tensor(obj):
        obj.record_stream(stream)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _record(o, stream)


def map_levels(fn, *per_level_args):
    """[fn(*args_l) for l in levels]; level 0 (the finest) on the current stream, every other level on a stream of its own,
    all joined into the current stream before returning"""
    levels = list(zip(*per_level_args))
    first = next((a for a in levels[0] if torch.is_tensor(a)), None) if levels else None
    if not ENABLED or len(levels) < 2 or first is None or not first.is_cuda:
        return [fn(*a) for a in levels]
    main = torch.cuda.current_stream(first.device)
    side = _streams(first.device, len(levels) - 1)
    fork = torch.cuda.Event()
    fork.record(main)
    outs = [None] * len(levels)
    if MODE == 2:
        if len(levels) < 3:
            return [fn(*a) for a in levels]
        side[0].wait_event(fork)
        for l in range(2, len(levels)):
            _record(levels[l], side[0])
            with torch.cuda.stream(side[0]):
                outs[l] = fn(*levels[l])
        outs[0], outs[1] = fn(*levels[0]), fn(*levels[1])
        main.wait_stream(side[0])
        for l in range(2, len(levels)):
            _record(outs[l], main)
        return outs
    for l in range(1, len(levels)):  # the small levels are enqueued first: they are in flight when the big one starts
        side[l - 1].wait_event(fork)
        _record(levels[l], side[l - 1])  # inputs come from the current stream's pool
        with torch.cuda.stream(side[l - 1]):
            outs[l] = fn(*levels[l])
    outs[0] = fn(*levels[0])
    for l in range(1, len(levels)):
        main.wait_stream(side[l - 1])
        _record(outs[l], main)  # allocated on the side stream's pool, consumed on the current stream from here on
    return outs
