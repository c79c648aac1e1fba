"""Sliding context-window schedule (host logic; defines the multi-GPU work decomposition).
Behaviour of src/pipelines/context.py:7-42 (`uniform`, `ordered_halving`), restated from its definition:
F <= context_size -> one window; else closed-loop windows of `context_size` consecutive frames (mod F)
every `context_size - overlap` frames, for each power-of-two stride below `context_stride`."""
import math


def ordered_halving(val: int) -> float:
    return int(f"{val:064b}"[::-1], 2) / (1 << 64)


def uniform(step, num_steps, num_frames, context_size, context_stride=3, context_overlap=4, closed_loop=True):
    if num_frames <= context_size:
        return [list(range(num_frames))]
    windows = []
    n_strides = min(context_stride, int(math.ceil(math.log2(num_frames / context_size))) + 1)
    frac = ordered_halving(step)
    for k in range(n_strides):
        cstep = 1 << k
        pad = int(round(num_frames * frac))
        start = int(frac * cstep) + pad
        stop = num_frames + pad + (0 if closed_loop else -context_overlap)
        for j in range(start, stop, context_size * cstep - context_overlap):
            windows.append([e % num_frames for e in range(j, j + context_size * cstep, cstep)])
    return windows


def get_context_scheduler(name):
    if name == "uniform":
        return uniform
    raise ValueError(f"Unknown context_overlap policy {name}")
