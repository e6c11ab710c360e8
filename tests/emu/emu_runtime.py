"""TEST INFRASTRUCTURE: run the product's device path on the host emulation of its own kernels (tests/emu/libomni_emu.so).

`emulated_device()` swaps the library bound behind omniparser_amd._lib for the emulation library (so every omni_op_launch / plan call
of the product lands in the host build of csrc/*.hip) and its device gate for one that answers "cpu", makes the torch.cuda entry points the product's host code touches inert (streams are
markers, synchronisation is a no-op: the emulation executes every launch synchronously) and switches plans to eager replay
(hipGraph capture is a property of the HIP runtime, not of the kernels).  Tensors live in host memory; their `data_ptr()` is what
the kernels dereference.  Nothing in the product imports this module."""
import contextlib
import os
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))


class HostStream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def wait_stream(self, other):
        pass

    def wait_event(self, ev):
        pass

    def record_event(self, ev=None):
        return ev or HostEvent()

    def query(self):
        return True


class HostEvent:
    def __init__(self, *a, **k):
        pass

    def record(self, stream=None):
        pass

    def wait(self, stream=None):
        pass

    def synchronize(self):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return 0.0


@contextlib.contextmanager
def emulated_device(env=None):
    import torch
    import build_emu
    from omniparser_amd import _lib as L
    # the product has no emulation switch: the test suite swaps the bound library and the device gate of omniparser_amd._lib
    from omniparser_amd import planner
    prev = (L._lib, L.require_device)
    prev_ws, planner.PlanBuilder.workspace_on_host = planner.PlanBuilder.workspace_on_host, True      # split-K launches as on the GPU
    L._lib = L.bind(build_emu.build())
    L.require_device = lambda device, what: torch.device("cpu")
    saved = {k: getattr(torch.cuda, k) for k in ("is_available", "current_device", "synchronize", "Stream", "Event", "stream", "device",
                                                 "current_stream", "device_count")}
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 1
    torch.cuda.current_device = lambda: 0
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.Stream = HostStream
    torch.cuda.Event = HostEvent
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.device = lambda d: contextlib.nullcontext()
    torch.cuda.current_stream = lambda *a, **k: HostStream()
    env = dict({"OMNI_HIPGRAPH": "0"}, **(env or {}))
    old_env = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    gc = sys.modules.get("gpu_checks")
    saved_g = (gc.DEV, gc._sync) if gc else None
    if gc:
        gc.DEV, gc._sync = "cpu", (lambda: None)
    try:
        yield L
    finally:
        if gc:
            gc.DEV, gc._sync = saved_g
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        for k, v in saved.items():
            setattr(torch.cuda, k, v)
        L._lib, L.require_device = prev
        planner.PlanBuilder.workspace_on_host = prev_ws
