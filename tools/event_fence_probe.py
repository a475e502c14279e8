"""What an event record costs the recording stream on MI355X: a chain of kernels with an event recorded every K kernels (a second
stream waiting on each), for torch's events (hipEventDisableTiming: system-scope fence) and for events created with
hipEventDisableSystemFence.  usage: python tools/event_fence_probe.py"""
import ctypes, time
import torch

hip = ctypes.CDLL('libamdhip64.so')
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
DISABLE_TIMING, NO_SYS_FENCE, REL_DEVICE = 0x2, 0x20000000, 0x40000000

dev = torch.device('cuda:0')
x = torch.randn(16 << 20, device=dev)       # 64 MB: ~25 us per pass, data that lives in L2 / MALL between kernels
y = torch.empty_like(x)
side = torch.cuda.Stream()
small = torch.zeros(1024, device=dev)


def run(kind, n=400, every=10):
    main = torch.cuda.current_stream()
    evs = []
    if kind.startswith('hip'):
        flags = DISABLE_TIMING | (NO_SYS_FENCE if 'nofence' in kind else 0) | (REL_DEVICE if 'reldev' in kind else 0)
        for _ in range(n // every):
            e = ctypes.c_void_p()
            assert hip.hipEventCreateWithFlags(ctypes.byref(e), flags) == 0
            evs.append(e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        torch.mul(x, 1.0001, out=y)
        if kind != 'none' and i % every == every - 1:
            if kind == 'torch':
                e = torch.cuda.Event()
                e.record(main)
                side.wait_event(e)
            else:
                e = evs[i // every]
                assert hip.hipEventRecord(e, ctypes.c_void_p(main.cuda_stream)) == 0
                assert hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), e, 0) == 0
            with torch.cuda.stream(side):
                small.add_(1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / n


for kind in ('none', 'torch', 'hip', 'hip_nofence', 'hip_reldev', 'none', 'torch', 'hip_nofence'):
    run(kind, 50)
    print(f'{kind:12s} {run(kind):7.2f} us per kernel (event every 10 kernels)')
