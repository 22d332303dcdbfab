"""hipGraph capture of a training step (torch.cuda.CUDAGraph is a hipGraph on ROCm).

At small batch sizes a step of these networks is hundreds of kernel launches of 2-100 us and the Python / ctypes launch path
(~16 us per launch) bounds it, not the GPU.  `capture_step` runs a step function a few times on a side
stream, captures one more run into a graph and hands back the graph: `graph.replay()` then re-issues every launch of the step from
the driver.  What the caller must guarantee is what `harness.EndToEndTrainer(hip_graphs=True)` arranges for its three problems:

* the step reads its inputs from tensors that stay alive and in place (copy each batch INTO them before a replay);
* parameters' `.grad` are None when the capture starts (`prepare`), so that the captured backward allocates the gradients in the
  graph's memory pool and every replay overwrites them in place;
* nothing the step reads is owned by eager code that might free it: weight packs must be (re)built inside the captured step
  (`engine.invalidate_packs()` in `prepare`; the Pix2Pix engines' pack cache hands a capture only entries that were themselves built
  during a capture), and the split-K workspace of the capture stream is pinned by the returned handle;
* no host synchronisation inside the step (no `.item()`, no host-drawn random numbers: device-side `torch.rand` is fine).
"""
from __future__ import annotations

from typing import Any, Callable, Optional, Tuple

import torch


class CapturedStep:
    """A captured step: `replay()` re-runs it; `result` is whatever the step function returned during the capture (static tensors
    that every replay overwrites, e.g. the loss)."""

    def __init__(self, graph: torch.cuda.CUDAGraph, result: Any, stream: torch.cuda.Stream, keep: tuple):
        self.graph, self.result, self.stream, self._keep = graph, result, stream, keep

    def replay(self) -> Any:
        self.graph.replay()
        return self.result

    def pool(self):
        return self.graph.pool()


def capture_step(step: Callable[[], Any], *, prepare: Optional[Callable[[], None]] = None, warmup: int = 2,
                 stream: Optional[torch.cuda.Stream] = None, pool=None) -> CapturedStep:
    """Run `step` `warmup` times on `stream` (a new side stream by default), call `prepare` (set gradients to None, drop weight
    packs ...), capture one more run of `step` into a hipGraph that shares `pool` (the pool of an earlier capture, or None) and
    return the CapturedStep.  The caller's current stream waits for the side stream on the way out; replays go to whatever stream
    is current (use `with torch.cuda.stream(captured.stream)` to stay on the capture stream)."""
    from . import ops
    side = stream if stream is not None else torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(0, warmup)):
            step()
        if prepare is not None:
            prepare()
        keep = (ops.splitk_workspace_for_capture(),)      # created eagerly on the capture stream and pinned: the graph bakes its pointer in
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, pool=pool, stream=side):
            result = step()
    torch.cuda.current_stream().wait_stream(side)
    return CapturedStep(graph, result, side, keep)
