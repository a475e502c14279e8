"""Can a kernel be timed INSIDE a torch.cuda.graph capture with external event-record nodes on this ROCm?  (tools/graph_event_probe.hip:
yes from plain HIP.)  usage: python tools/graph_event_probe.py torch|raw [blocking]"""
import ctypes, sys, torch
hip = ctypes.CDLL('libamdhip64.so')
hip.hipGetErrorString.restype = ctypes.c_char_p
kind = sys.argv[1]
def rec(ev, st, flags):
    return hip.hipGetErrorString(hip.hipEventRecordWithFlags(ctypes.c_void_p(ev), ctypes.c_void_p(st), ctypes.c_uint(flags))).decode()
x = torch.zeros(1 << 20, device='cuda')
side = torch.cuda.Stream()
def mk():
    if kind == 'torch':
        e = torch.cuda.Event(enable_timing=True); e.record(side); return e, e.cuda_event
    h = ctypes.c_void_p(); hip.hipEventCreate(ctypes.byref(h)); return h, h.value
(k0, e0), (k1, e1) = mk(), mk()
torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[2] == 'rawcapture':
    # capture by hand on a stream created here (blocking flags), torch ops inside
    sh = ctypes.c_void_p(); hip.hipStreamCreate(ctypes.byref(sh))
    ext = torch.cuda.ExternalStream(sh.value)
    with torch.cuda.stream(ext):
        print('begin', hip.hipGetErrorString(hip.hipStreamBeginCapture(sh, 1)).decode())
        a = rec(e0, sh.value, 1)
        print('record in hand-made capture:', a)
        gh = ctypes.c_void_p()
        print('end', hip.hipGetErrorString(hip.hipStreamEndCapture(sh, ctypes.byref(gh))).decode())
    sys.exit(0)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    st = torch.cuda.current_stream().cuda_stream
    y = x * 2
    a = rec(e0, st, 1)
    print('first record:', a, flush=True)
    hip.hipGetLastError()
    y = y * 1.0001
    b = rec(e1, st, 1)
    hip.hipGetLastError()
    z = y + 1
g.replay(); torch.cuda.synchronize()
ms = ctypes.c_float(-1)
r = hip.hipGetErrorString(hip.hipEventElapsedTime(ctypes.byref(ms), ctypes.c_void_p(e0), ctypes.c_void_p(e1))).decode()
print(kind, '| record:', a, '/', b, '| elapsed:', r, ms.value)
